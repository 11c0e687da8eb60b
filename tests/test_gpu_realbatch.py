"""-m gpu: the reference's OWN batch sizes and image shapes through the kernels the benchmark runs.

The reference trains at batch 3 (configs/training.json:6), buckets by image shape in groups of 20 (model/utils/data_generator.py:41,84-122)
and evaluates at 20 (evaluate_txt.py:42); its images come in the 21 sizes of configs/data.json:22-28 (halved by the build-time downsample:
50x120 ... 160x400, 800x800).  The persistent decoder chains (csrc/xdec.hip) take 8 / 16 / 32 / 64 rows, so Engine.forward fills such a
batch up with DEAD rows (token ids copied from its own samples, formula length 0: outside the loss mask of img2seq.py:68-71; lxo_shape.live_B keeps the
encoder on the live images and gives the dead rows zero features).  Held here:

* B = 3 and B = 20 report chain_used and chain_used_bwd;
* the padded bf16 chain step == the f32 parity mode on the UNPADDED batch (launch-per-step kernels) within the existing bf16 bars, and the
  loss is within the north-star 1e-3 of the CPU oracle on the unpadded batch;
* the padded chain == the unpadded launch-per-step kernels of the SAME dtype (what the padding replaces) much tighter;
* n_words / ce_words do not see the dead rows; evaluate_batch (the reference's evaluation at 20) agrees with the oracle;
* a 5-step Adam trajectory at batch 3 follows the oracle's within 1e-3 per step.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_common import *  # noqa

V = 50
# (H, W) = real buckets after the /2 downsample: [240,100] [320,80] [720,120] of configs/data.json:23-25
SHAPES = [(50, 120), (40, 160), (60, 360)]


def _threads():
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, 32)))


def _one(dtype, img, f, l, pad=True, seed=5, deterministic=None):
    eng = Engine(V, dtype=dtype, seed=seed, deterministic=deterministic)
    eng.pad_train = pad
    eng.forward(img, f)
    n = int(l.sum())
    stats = eng.loss(l, 1.0 / n).cpu().numpy().copy()
    eng.backward()
    torch.cuda.synchronize()
    return eng, stats, eng.grad_dict()


@pytest.mark.parametrize("B,H,W", [(3, 50, 120), (20, 50, 120), (20, 40, 160), (3, 60, 360), (20, 60, 360), (5, 40, 160), (40, 50, 120),
                                   (3, 32, 128), (12, 32, 128)])      # config 1's crops: 28 regions -- fewer than the 32 chunks a one-sample chain deals them over
def test_real_batches_run_the_chains_and_match_f32_and_oracle(B, H, W):
    _threads()
    img, f, l = batch(B, H, W, V, 5, 30, seed=100 + B + H)
    n = int(l.sum())
    e16, s16, g16 = _one("bf16", img, f, l)
    assert e16.chain_used and e16.chain_used_bwd, (e16.chain_used, e16.chain_used_bwd)
    assert int(e16.shape.B) in (8, 16, 32, 64) and int(e16.shape.B) >= B and e16.live_B == B
    assert s16[1] == n, (s16, n)                                  # n_words = sum of the REAL lengths (img2seq.py:75)
    e32, s32, g32 = _one("f32", img, f, l)
    assert int(e32.shape.B) == B                                  # the parity mode takes the batch as it is
    P = oracle_params(e32)
    loss_ref, G, ce, nw = R.train_grads(P, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l))
    l16, l32 = s16[0] / s16[1], s32[0] / s32[1]
    assert abs(l32 - float(loss_ref)) <= 2e-5 * float(loss_ref), (l32, float(loss_ref))
    assert abs(l16 - float(loss_ref)) <= 1e-3 * float(loss_ref), (l16, float(loss_ref))      # north star: 1e-3 relative on the training loss
    worst = (1.0, None)
    for k in G:
        c32 = cosine(g32[k], G[k].numpy())
        assert c32 > 0.9999, (k, c32)
        c = cosine(g16[k], g32[k])
        if c < worst[0]:
            worst = (c, k)
        assert np.isfinite(g16[k]).all(), k
        assert c > 0.995, (k, c)                                  # bf16 storage of six conv layers at 3 .. 20 samples; measured below
    print("B=%d (chain batch %d) %dx%d: loss bf16 %.6f f32 %.6f oracle %.6f; worst bf16-vs-f32 gradient cosine %.6f (%s)" % (
        B, int(e16.shape.B), H, W, l16, l32, float(loss_ref), worst[0], worst[1]))


@pytest.mark.parametrize("B,H,W", [(3, 50, 120), (20, 40, 160), (20, 60, 360)])
def test_padded_chain_equals_the_unpadded_launch_per_step_kernels(B, H, W):
    """Same dtype, same weights: what the padding replaces (the launch-per-step kernels on B rows) against what it runs (the chains on the
    filled-up batch).  The dead rows contribute exact zeros, so the two differ by summation order / the chain's K split only."""
    img, f, l = batch(B, H, W, V, 5, 30, seed=7 + B)
    ea, sa, ga = _one("bf16", img, f, l, pad=True)
    eb, sb, gb = _one("bf16", img, f, l, pad=False)
    assert ea.chain_used and ea.chain_used_bwd and not eb.chain_used and int(eb.shape.B) == B
    assert sa[1] == sb[1]
    la, lb = sa[0] / sa[1], sb[0] / sb[1]
    assert abs(la - lb) <= 1e-4 * abs(lb), (la, lb)
    for k in ga:
        assert cosine(ga[k], gb[k]) > 0.9999, (k, cosine(ga[k], gb[k]))


def test_dead_rows_contribute_exact_zeros():
    """Deterministic bf16 mode (no float atomics): the step on 20 samples filled up to 32 is bit-identical from run to run, and every per-sample
    gradient the dead rows own (d_y6 rows of the dead images in ws region "d_img") is exactly zero."""
    img, f, l = batch(20, 50, 120, V, 5, 30, seed=77)
    e1, s1, g1 = _one("bf16", img, f, l, deterministic=True)
    e2, s2, g2 = _one("bf16", img, f, l, deterministic=True)
    assert e1.chain_used and e1.chain_used_bwd
    assert np.array_equal(s1, s2)
    for k in g1:
        assert np.array_equal(g1[k], g2[k]), k
    from latex_ocr_amd.model.utils.image import encoder_out_hw
    Hp, Wp = encoder_out_hw(50, 120)
    d = e1.region("d_img", "ct", (32, Hp * Wp, 512)).float().cpu().numpy()
    assert np.abs(d[:20]).max() > 0 and np.all(d[20:] == 0.0)


def test_evaluate_batch_at_20_matches_the_oracle():
    """evaluate_txt.py:42 / img2seq.py:215-252: (sum CE, n_words) of a teacher-forced batch of 20."""
    _threads()
    img, f, l = batch(20, 50, 200, V, 5, 30, seed=9)
    eng = Engine(V, dtype="bf16", seed=2)
    ce, nw = eng.evaluate_batch(img, f, l)
    assert eng.chain_used and nw == int(l.sum())
    P = oracle_params(eng)
    loss_ref, G, ce_ref, nw_ref = R.train_grads(P, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l))
    assert int(nw_ref) == nw
    assert abs(ce - float(ce_ref)) <= 1e-3 * float(ce_ref), (ce, float(ce_ref))


def test_adam_trajectory_at_the_reference_batch_of_3():
    """configs/training.json: batch_size 3, Adam, lr_init 1e-3 -- five steps on five different batches of 3, per-step loss within 1e-3 of the oracle."""
    _threads()
    eng = Engine(V, dtype="bf16", seed=1)
    P = oracle_params(eng)
    opt = R.AdamTF(P)
    for s in range(5):
        img, f, l = batch(3, 50, 120, V, 5, 30, seed=300 + s)
        got = eng.train_step(img, f, l, 1e-3)
        ref = R.train_step(P, opt, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l), 1e-3)
        assert abs(got - ref) <= 1e-3 * abs(ref), (s, got, ref)
    assert eng.chain_used and eng.chain_used_bwd and eng.chain_failures == 0 and eng.adam_t == 5
