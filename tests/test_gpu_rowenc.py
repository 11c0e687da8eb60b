"""-m gpu: the optional row-BiLSTM encoder (lxo_shape.encoder_rnn; north_star names it, the reference does not have it:
model/encoder.py:4 imports GRUCell / LSTMCell and never uses them).  Off by default and outside the parity contract with the
reference; what is held here is the extension's own specification, oracle/ref_model.py:row_bilstm (a bidirectional TF LSTMCell,
C/2 units per direction, over every feature-map row), through the same C-ABI calls as the default path: loss, every parameter
gradient incl. the four new variables, greedy decode."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_common import *  # noqa

DIMS = dict(row_bilstm=True)
ROW = "Encoder/row_encoder/bidirectional_rnn/"


def _check(dtype, tol_loss, min_cos, hw=(32, 128), n=4):
    V = 50
    img, f, l = batch(n, hw[0], hw[1], V, 5, 12, seed=17)
    eng = Engine(V, dtype=dtype, seed=3, dims=DIMS)
    P = oracle_params(eng)
    assert ROW + "fw/lstm_cell/kernel" in P and P[ROW + "bw/lstm_cell/kernel"].shape == (768, 1024)
    # the TF initialiser leaves the LSTM biases at zero: give them values so that the bias path is exercised
    rng = np.random.Generator(np.random.PCG64(5))
    for d in ("fw", "bw"):
        P[ROW + d + "/lstm_cell/bias"] = torch.from_numpy(rng.uniform(-0.1, 0.1, size=1024).astype(np.float32))
    eng.load_params({k: v.numpy() for k, v in P.items()})
    eng.forward(img, f)
    ntok = int(l.sum())
    stats = eng.loss(l, 1.0 / ntok).cpu().numpy()
    eng.backward()
    torch.cuda.synchronize()
    loss_ref, G, _, _ = R.train_grads(P, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l))
    loss = stats[0] / stats[1]
    assert abs(loss - float(loss_ref)) / float(loss_ref) < tol_loss, (loss, float(loss_ref))
    got = eng.grad_dict()
    worst = (1.0, None)
    for k in G:
        c = cosine(got[k], G[k].numpy())
        if c < worst[0]:
            worst = (c, k)
        assert c > min_cos, (k, c, rel(got[k], G[k].numpy()))
    print("row-BiLSTM %s %dx%d: loss %.6f (oracle %.6f), worst gradient cosine %.7f (%s)" % (dtype, hw[0], hw[1], loss, float(loss_ref), worst[0], worst[1]))
    return eng, P, img


def test_row_bilstm_fwd_bwd_f32():
    eng, P, img = _check("f32", 2e-5, 0.99999)
    enc = R.encoder(P, torch.from_numpy(img)).reshape(img.shape[0], -1, 512).numpy()
    got = eng.region("img", "ct", enc.shape).float().cpu().numpy()
    assert rel(got, enc) < 2e-5
    ids = eng.greedy_decode(img, 49, max_iter=20)
    ref = R.greedy_decode(P, torch.from_numpy(img), 49, max_iter=20).numpy()
    assert ids.shape == ref.shape and np.array_equal(ids, ref)


def test_row_bilstm_fwd_bwd_bf16():
    _check("bf16", 2e-3, 0.97)


def test_row_bilstm_odd_shape_f32():
    _check("f32", 2e-5, 0.9999, hw=(40, 150), n=3)       # H' = 3 rows, W' = 17 positions


def test_row_bilstm_train_steps_and_off_by_default():
    V = 50
    assert Engine(V, dtype="bf16", seed=0).n_params == 8367088          # the default inventory is untouched
    eng = Engine(V, dtype="bf16", seed=0, dims=DIMS)
    assert eng.n_params == 8367088 + 2 * (768 * 1024 + 1024)
    img, f, l = batch(8, 32, 128, V, 5, 12, seed=23)
    losses = [eng.train_step(img, f, l, 1e-3) for _ in range(12)]
    assert losses[-1] < losses[0] and all(np.isfinite(losses)), losses
