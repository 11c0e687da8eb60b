"""-m gpu: parity in the TRAINED regime (north_star: "loss-curve equivalent to the CPU reference").

At random initialisation every loss is ln V + epsilon and a 1e-3 bar says little; here the product path (bf16 engine, and the
f32 parity mode) and the CPU oracle run the reference's training loop (model/img2seq.py:144-196: minibatches -> pad -> step ->
LRSchedule.update(batch_no)) for 100 Adam steps on the SAME batches of a learnable toy set (tests/refgold.py:toy_set), through
the reference LRSchedule's warm-up -> lr_init -> exponential decay -> lr_min phases (lr_schedule.py:82-118); the loss falls
from 3.97 to ~0.07 and the whole curve is compared step by step.

What "equivalent" can mean: Adam turns rounding noise into O(lr) steps wherever |g| ~ eps, so two runs of the SAME float32
arithmetic that differ only in summation order (the oracle with two intra-op thread counts: the CONTROL below) already part
by ~1e-3 within 20 steps and by several per cent once the loss is small.  The bars: the first 10 steps (before that fork
matters) are held to north_star's 1e-3 (f32 mode: 1e-4; measured 2.2e-4 / 4e-6); over the whole curve the 5-step moving average
of |log loss - log oracle loss| must stay within 4x the control's own spread (floor 0.20 for the deterministic f32 mode, 0.30 for bf16)
and the mean loss of the last 10 steps within 10 %.  The f32 parity mode has no float atomics (round 4: every reduction ordered), so its
100-step curve is the SAME in every run: the test runs it twice and asserts bit equality.  The bf16 mode keeps atomics: its curve varies from
run to run of one binary as much as it differs from the oracle (profiles/r04_trained_spread.txt), so its whole-curve bars are held by the
geometric mean of 8 runs (see the test)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_common import *  # noqa
import refgold
from latex_ocr_amd.model.utils.general import minibatches
from latex_ocr_amd.model.utils.lr_schedule import LRSchedule

V, BS, EPOCHS, NB = 50, 20, 20, 5          # 100 crops / 20 = 5 steps per epoch, 100 steps


def _threads():
    try:
        return max(2, min(16, len(os.sched_getaffinity(0))))
    except Exception:
        return 8


def _batches():
    imgs, forms = refgold.toy_set(100, 32, 128, V, 41)
    out = []
    for epoch in range(EPOCHS):
        for i, (bi, bf) in enumerate(minibatches(zip(imgs, forms), BS)):
            f, l = pad_batch_formulas(bf, V - 2, V - 1)
            out.append((epoch * NB + i, pad_batch_images(bi), f, l))
    return imgs, out


def _schedule():
    return LRSchedule(lr_init=1e-3, lr_warm=1e-4, end_warm=2 * NB, start_decay=6 * NB, end_decay=13 * NB, lr_min=1e-4)


def _oracle_curve(P0, batches, threads):
    torch.set_num_threads(threads)
    P = {k: v.clone() for k, v in P0.items()}
    opt, s, out = R.AdamTF(P), _schedule(), []
    for no, img, f, l in batches:
        out.append(R.train_step(P, opt, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l), s.lr))
        s.update(batch_no=no)
    return np.array(out), P


def _engine_curve(dtype, P0, batches):
    eng = Engine(V, dtype=dtype, seed=0)
    eng.load_params({k: v.numpy() for k, v in P0.items()})
    s, out = _schedule(), []
    for no, img, f, l in batches:
        out.append(eng.train_step(img, f, l, s.lr))
        s.update(batch_no=no)
    return np.array(out), eng


def _spread(a, b):
    d = np.abs(np.log(a) - np.log(b))
    return float(np.convolve(d, np.ones(5) / 5, mode="valid").max())


@pytest.fixture(scope="module")
def curves():
    imgs, batches = _batches()
    P0 = {k: torch.from_numpy(v.copy()) for k, v in Engine(V, dtype="f32", seed=0).get_params().items()}
    n = _threads()
    ref, P = _oracle_curve(P0, batches, n)
    ctl, _ = _oracle_curve(P0, batches, max(1, n // 2 - 1))
    print("oracle: loss %.4f -> %.4f over %d steps; CONTROL (same arithmetic, %d vs %d threads): first 10 steps max rel %.2e, "
          "whole-curve spread %.3f, last-10 mean %.4f vs %.4f" % (ref[0], ref[-1], len(ref), n, max(1, n // 2 - 1),
          (np.abs(ref - ctl) / ref)[:10].max(), _spread(ref, ctl), ref[-10:].mean(), ctl[-10:].mean()))
    return imgs, batches, P0, ref, ctl, P


@pytest.mark.parametrize("dtype,first_bar", [("bf16", 1e-3), ("f32", 1e-4)])
def test_loss_curve_100_steps_vs_oracle(curves, dtype, first_bar):
    imgs, batches, P0, ref, ctl, P = curves
    got, eng = _engine_curve(dtype, P0, batches)
    runs = [got]
    if dtype == "f32":       # the parity mode is reproducible bit for bit: a second run of the 100 steps gives the same curve and the same weights
        again, eng2 = _engine_curve(dtype, P0, batches)
        assert got.tobytes() == again.tobytes(), np.abs(got - again).max()
        assert torch.equal(eng.params, eng2.params)
        del eng2
    else:
        # bf16 keeps f32 atomics in its reductions: ONE 100-step curve is a sample of a noisy trajectory.  57 runs of one binary in 9 processes
        # on 3 boxes (tools/trained_spread.py, profiles/r04_trained_spread.txt): spread 0.089 .. 0.374 against the oracle (mean 0.17, five
        # above 0.25), 0.03 .. 0.44 between two bf16 runs of one process, last-10 ratio 0.96 .. 1.15 -- round 3's single-run bars (0.30, 10 %)
        # were measuring that noise and failed about one run in 20.  The whole-curve bars are therefore held by the GEOMETRIC MEAN of 8 runs
        # (means of 5 measured: spread 0.095 .. 0.220, mean 0.14, sd 0.04; last-10 ratio 1.009 .. 1.074, mean 1.034, sd 0.02), every single
        # run by the first-10-steps bar and by a coarse bound (1.0 -- a factor e, max seen 0.465; learnt) that only a diverged run would miss.
        # The noise is order only: gradients repeat to 4e-7 of their largest element from run to run (tests/test_gpu_determinism.py,
        # profiles/r04_bf16_repeat.txt); a bf16 weight that rounds the other way turns such a difference into a 2^-9 step.
        for _ in range(7):
            more, e2 = _engine_curve(dtype, P0, batches)
            runs.append(more)
            del e2
        for r in runs:
            assert (np.abs(r - ref) / ref)[:10].max() <= first_bar, (np.abs(r - ref) / ref)[:10]
            assert _spread(r, ref) <= 1.0 and r[-1] < 0.1 * r[0], (_spread(r, ref), r[-5:])
        got = np.exp(np.mean(np.log(np.array(runs)), axis=0))
    rel = np.abs(got - ref) / ref
    sp, spc = _spread(got, ref), _spread(ctl, ref)
    print("%s: loss %.4f -> %.4f (oracle %.4f -> %.4f); first 10 steps max rel %.2e (bar %.0e); steps 10..19 max %.2e; whole-curve "
          "spread %.3f (control %.3f; single runs %s); last-10 mean %.4f vs %.4f" % (dtype, got[0], got[-1], ref[0], ref[-1], rel[:10].max(), first_bar,
          rel[10:20].max(), sp, spc, ["%.3f" % _spread(r, ref) for r in runs], got[-10:].mean(), ref[-10:].mean()))
    assert ref[-1] < 0.1 * ref[0], "the toy set was not learnt: %s" % ref[-5:]
    assert rel[:10].max() <= first_bar, rel[:10]
    # f32: the parity mode is deterministic -- one value per binary, measured 0.100 .. 0.112 -- and is held to 4x the control, floor 0.20.
    # bf16 (mean of 8 runs): floor 0.30, as in round 3; its plateau sits 3.4 % (sd 2 %) above the f32 oracle's: 15 % there, 10 % for f32 (measured 0.7 %)
    assert sp <= max(4.0 * spc, 0.20 if dtype == "f32" else 0.30), (sp, spc)
    assert abs(got[-10:].mean() - ref[-10:].mean()) <= (0.10 if dtype == "f32" else 0.15) * ref[-10:].mean()
    # decode: each side from its OWN 100-step weights (reported), then the engine from the ORACLE's weights (asserted)
    img = pad_batch_images(imgs[:40])
    rid = R.greedy_decode(P, torch.from_numpy(img), V - 1, max_iter=30).numpy()
    own = eng.greedy_decode(img, V - 1, max_iter=30)
    n = min(own.shape[1], rid.shape[1])
    print("%s: greedy from own weights vs oracle from its own: steps %d vs %d, token agreement %.4f" % (
        dtype, own.shape[1], rid.shape[1], float((own[:, :n] == rid[:, :n]).mean())))
    eng.load_params({k: v.numpy() for k, v in P.items()})
    ids = eng.greedy_decode(img, V - 1, max_iter=30)
    if dtype == "f32":
        assert ids.shape == rid.shape and np.array_equal(ids, rid)           # same checkpoint: token for token on the trained model
    else:
        n = min(ids.shape[1], rid.shape[1])
        agree = float((ids[:, :n] == rid[:, :n]).mean())
        print("bf16: greedy from the oracle's 100-step weights: agreement %.4f" % agree)
        assert ids.shape == rid.shape and agree >= 0.99
