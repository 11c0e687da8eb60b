"""-m gpu: the library tolerates HIP-graph capture ("must tolerate ... HIP-graph capture (no sync / malloc inside)").

One training step -- lxo_encoder_fwd, lxo_decoder_train_fwd, lxo_ce_loss_fwd_bwd, lxo_train_bwd (with its weight-gradient side stream forked
and joined by events inside the capture), the optimizer and lxo_pack_weights -- is recorded into ONE graph (torch.cuda.CUDAGraph = hipGraph,
global capture mode: any synchronising or allocating runtime call from the capturing thread would fail the capture) and replayed; the losses of
the replays equal the eager steps' -- bit for bit in the modes without float atomics (f32 parity mode; bf16 deterministic mode, which runs the
persistent decoder chains: 256 spin-waiting workgroups launched from a graph node).

What one sess.run covers in the reference: model/img2seq.py:169."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_common import *  # noqa

V = 50


def _dev_batch(B, H, W, seed):
    img, f, l = batch(B, H, W, V, 5, 20, seed=seed)
    return (torch.from_numpy(img).cuda(), torch.from_numpy(f).cuda(), torch.from_numpy(l.astype(np.int32)).cuda(), int(l.sum()))


def _loss_of(eng):
    s = eng.region("loss")[:2].cpu().numpy()
    return float(s[0]) / float(s[1])


def _body(eng, img, f, l, n, lr, whole_step):
    eng.forward(img, f)
    eng.loss(l, 1.0 / n)
    eng.backward()
    if whole_step:
        eng.optimizer_step(lr)                                   # SGD: the learning rate is a by-value kernel argument, the same in every replay


@pytest.mark.parametrize("dtype,det,B", [("f32", None, 4), ("bf16", True, 16), ("bf16", False, 16)])
def test_a_training_step_captured_into_a_graph_replays_the_eager_losses(dtype, det, B):
    img, f, l, n = _dev_batch(B, 32, 128, seed=11)
    lr = 0.05

    def fresh():
        e = Engine(V, dtype=dtype, seed=3, deterministic=det)
        e.set_optimizer("sgd")
        return e

    # eager reference: 4 SGD steps on the same batch
    ref = fresh()
    want = []
    for _ in range(4):
        _body(ref, img, f, l, n, lr, True)
        want.append(_loss_of(ref))
    assert want[-1] < want[0]
    eng = fresh()
    _body(eng, img, f, l, n, lr, True)                           # eager warm-up = step 1: one-time attributes, workspace, the chains' first-use check
    got = [_loss_of(eng)]
    if dtype == "bf16":
        assert eng.chain_used and eng.chain_used_bwd
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):                                    # capture (nothing executes): steps 2 .. 4 are replays of this one graph
        _body(eng, img, f, l, n, lr, True)
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        got.append(_loss_of(eng))
    print("%s det=%s: eager %s graph %s" % (dtype, det, want, got))
    if dtype == "f32" or det:
        assert got == want, (got, want)                          # no float atomics: bit for bit
        assert torch.equal(eng.params, ref.params)
    else:
        for a, b in zip(got, want):
            assert abs(a - b) <= 2e-4 * abs(b), (got, want)      # atomic-order noise of the default bf16 mode
    if dtype == "bf16":
        used, err = eng.chain_status()
        usedb, errb = eng.chain_status(backward=True)
        assert used and usedb and not err and not errb           # the chains ran (and assembled) inside the graph replays


def test_forward_backward_in_a_graph_adam_outside():
    """The gradient computation as a graph, the Adam update (its bias-corrected step size changes every step: a by-value argument) eagerly
    behind each replay -- how a caller with a host-side learning-rate schedule (model/utils/lr_schedule.py) would use a graph."""
    img, f, l, n = _dev_batch(16, 32, 128, seed=12)
    ref = Engine(V, dtype="bf16", seed=4, deterministic=True)
    want = []
    for _ in range(3):
        _body(ref, img, f, l, n, 1e-3, False)
        ref.optimizer_step(1e-3)
        want.append(_loss_of(ref))
    eng = Engine(V, dtype="bf16", seed=4, deterministic=True)
    _body(eng, img, f, l, n, 1e-3, False)
    eng.optimizer_step(1e-3)
    got = [_loss_of(eng)]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _body(eng, img, f, l, n, 1e-3, False)
    for _ in range(2):
        g.replay()
        eng.optimizer_step(1e-3)
        torch.cuda.synchronize()
        got.append(_loss_of(eng))
    assert got == want, (got, want)
    assert torch.equal(eng.params, ref.params)
