"""-m gpu: the f32 PARITY mode is reproducible bit for bit ("keep deterministic order in parity mode").

Every reduction of the f32 mode runs in a fixed order: the weight-gradient GEMMs take one row range per output tile, bias
gradients / d_beta / the loss statistics / conv1's gradients go through per-workgroup slots that are added in slot order
(ws region "det_part", csrc: DetScratch), the attention backward runs one chunk per sample, the embedding scatter lists its rows
in slot order -- no float atomics anywhere on the path.  So two runs of the same step on the same inputs agree in every bit:
loss statistics, all 28 gradients, and a 30-step Adam trajectory (where any rounding difference would be amplified).
bf16 mode keeps its atomic epilogues by default (speed): its run-to-run differences are held to the size of a reordered f32 sum (test below) --
anything larger would be a race, not an order.  With lxo_shape.deterministic (Engine(deterministic=True) / LXO_DETERMINISTIC=1; round 5) the
bf16 mode takes ordered paths too -- conv weight gradients through per-range slabs + an ordered pass, bias sums / d_beta / loss / conv1 through
the slots, the dense weight gradients with one row range per tile -- and is held to the same bit-for-bit bars, through the persistent
decoder chains (B = 16) and through the launch-per-step kernels (B = 12) alike."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_common import *  # noqa


def _one_pass(V, img, f, l, dims=None):
    eng = Engine(V, dtype="f32", seed=3, dims=dims)
    eng.forward(img, f)
    stats = eng.loss(l, 1.0 / int(l.sum())).cpu().numpy().copy()
    eng.backward()
    torch.cuda.synchronize()
    return stats, eng.grad_dict()


@pytest.mark.parametrize("shape", [(12, 64, 256, 120, 5, 24), (5, 50, 150, 50, 3, 9)])
def test_f32_gradients_bit_identical_run_to_run(shape):
    n, H, W, V, lo, hi = shape
    img, f, l = batch(n, H, W, V, lo, hi, seed=11)
    runs = [_one_pass(V, img, f, l) for _ in range(3)]
    s0, g0 = runs[0]
    assert len(g0) == 28
    for s, g in runs[1:]:
        assert s.tobytes() == s0.tobytes(), (s, s0)
        bad = [k for k in g0 if g[k].tobytes() != g0[k].tobytes()]
        assert not bad, "f32 gradients differ between two runs of the same step: %s" % bad


def test_f32_encoder_variants_bit_identical_run_to_run():
    """the "cnn" encoder (strided conv, col2im) and the optional row-BiLSTM encoder take their own backward kernels"""
    img, f, l = batch(4, 64, 128, 50, 3, 9, seed=12)
    for dims in ({"cnn": True}, {"row_bilstm": True}):
        a = _one_pass(50, img, f, l, dims=dims)
        b = _one_pass(50, img, f, l, dims=dims)
        assert a[0].tobytes() == b[0].tobytes()
        bad = [k for k in a[1] if a[1][k].tobytes() != b[1][k].tobytes()]
        assert not bad, (dims, bad)


def test_f32_adam_trajectory_bit_identical_run_to_run():
    """30 Adam steps with global-norm clipping (its norm is an ordered sum too): identical loss curves and identical final weights"""
    V = 50
    img, f, l = batch(20, 32, 128, V, 5, 12, seed=13)
    out = []
    for _ in range(2):
        eng = Engine(V, dtype="f32", seed=0)
        eng.set_optimizer("adam")
        curve = np.array([eng.train_step(img, f, l, 1e-3, clip=5.0) for _ in range(30)], np.float64)
        out.append((curve, eng.params.detach().cpu().numpy().copy()))
    assert out[0][0].tobytes() == out[1][0].tobytes(), np.abs(out[0][0] - out[1][0]).max()
    assert out[0][1].tobytes() == out[1][1].tobytes()
    assert out[0][0][-1] < out[0][0][0]


@pytest.mark.parametrize("B,H,W", [(20, 32, 128), (16, 48, 160)])
def test_bf16_step_repeats_to_atomic_order_noise(B, H, W):
    """bf16 mode: float atomics only REORDER f32 sums.  The same step from the same state, four times (B = 20: the launch-per-step
    kernels; B = 16: the persistent chains): every gradient within 5e-6 of its largest element from run to run (measured 4e-7,
    profiles/r04_bf16_repeat.txt), the loss within 1e-6.  (What a 100-step Adam trajectory makes of that noise -- a bf16 weight that
    rounds the other way is a 2^-9 step -- is the subject of tests/test_gpu_trained.py.)"""
    V = 50
    img, f, l = batch(B, H, W, V, 3, 9, seed=5)
    runs = []
    for _ in range(4):
        eng = Engine(V, dtype="bf16", seed=2)
        eng.forward(img, f)
        st = eng.loss(l, 1.0 / int(l.sum())).cpu().numpy().copy()
        eng.backward()
        torch.cuda.synchronize()
        runs.append((st[0] / st[1], eng.grad_dict()))
        del eng
    l0, g0 = runs[0]
    worst = 0.0
    for lv, g in runs[1:]:
        assert abs(lv - l0) <= 1e-6 * abs(l0), (lv, l0)
        for k in g0:
            d = float(np.abs(g[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-30))
            worst = max(worst, d)
            assert d <= 5e-6, (k, d)
    print("B=%d: bf16 gradients run to run: max |difference| / max |g| = %.1e" % (B, worst))


def _one_pass_bf16_det(V, img, f, l, dims=None, pad=True):
    eng = Engine(V, dtype="bf16", seed=3, dims=dims, deterministic=True)
    eng.pad_train = pad                                          # False: a batch the chains do not take stays on the launch-per-step kernels
    eng.forward(img, f)
    stats = eng.loss(l, 1.0 / int(l.sum())).cpu().numpy().copy()
    eng.backward()
    torch.cuda.synchronize()
    return stats, eng.grad_dict(), (eng.chain_used, eng.chain_used_bwd)


@pytest.mark.parametrize("shape", [(16, 64, 256, 120, 5, 24, True), (12, 64, 256, 120, 5, 24, False), (12, 64, 256, 120, 5, 24, True), (64, 32, 128, 50, 3, 9, True)])
def test_bf16_deterministic_mode_bit_identical_run_to_run(shape):
    n, H, W, V, lo, hi, pad = shape
    img, f, l = batch(n, H, W, V, lo, hi, seed=11)
    runs = [_one_pass_bf16_det(V, img, f, l, pad=pad) for _ in range(3)]
    s0, g0, chains = runs[0]
    # B = 16 / 64, and 12 filled up to 16 with dead rows: through xdec_fwd_kernel / xdec_bwd_kernel; B = 12 as it is: the launch-per-step kernels
    assert chains == ((n % 8 == 0 or pad), (n % 8 == 0 or pad)), chains
    assert len(g0) == 28
    for s, g, _ in runs[1:]:
        assert s.tobytes() == s0.tobytes(), (s, s0)
        bad = [k for k in g0 if g[k].tobytes() != g0[k].tobytes()]
        assert not bad, "bf16 deterministic-mode gradients differ between two runs of the same step: %s" % bad
    # and it is the same mathematics as the default (atomic) bf16 mode: loss identical to f32 rounding, gradients to summation order
    eng = Engine(V, dtype="bf16", seed=3)
    eng.pad_train = pad                                          # the same kernels (chains / launch-per-step) in both modes
    eng.forward(img, f)
    st = eng.loss(l, 1.0 / int(l.sum())).cpu().numpy().copy()
    eng.backward()
    torch.cuda.synchronize()
    assert st[1] == s0[1] and abs(st[0] - s0[0]) <= 2e-6 * abs(s0[0]), (st, s0)
    gd = eng.grad_dict()
    for k in g0:
        c = cosine(gd[k], g0[k])
        assert c > 0.999999, (k, c)


def test_bf16_deterministic_encoder_variants_bit_identical_run_to_run():
    img, f, l = batch(8, 64, 128, 50, 3, 9, seed=12)
    for dims in ({"cnn": True}, {"row_bilstm": True}):
        a = _one_pass_bf16_det(50, img, f, l, dims=dims)
        b = _one_pass_bf16_det(50, img, f, l, dims=dims)
        assert a[0].tobytes() == b[0].tobytes()
        bad = [k for k in a[1] if a[1][k].tobytes() != b[1][k].tobytes()]
        assert not bad, (dims, bad)


@pytest.mark.parametrize("B,pad", [(16, True), (20, False), (20, True)])
def test_bf16_deterministic_adam_trajectory_bit_identical_run_to_run(B, pad):
    """30 Adam steps with clipping, bf16 deterministic mode, through the chains (B = 16; B = 20 filled up to 32) and the launch-per-step kernels (B = 20 as it is)"""
    V = 50
    img, f, l = batch(B, 32, 128, V, 5, 12, seed=13)
    out = []
    for _ in range(2):
        eng = Engine(V, dtype="bf16", seed=0, deterministic=True)
        eng.pad_train = pad
        curve = np.array([eng.train_step(img, f, l, 1e-3, clip=5.0) for _ in range(30)], np.float64)
        out.append((curve, eng.params.detach().cpu().numpy().copy(), eng.chain_used and eng.chain_used_bwd))
    assert out[0][2] == (B % 8 == 0 or pad)
    assert out[0][0].tobytes() == out[1][0].tobytes(), np.abs(out[0][0] - out[1][0]).max()
    assert out[0][1].tobytes() == out[1][1].tobytes()
    assert out[0][0][-1] < out[0][0][0]

