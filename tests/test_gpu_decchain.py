"""-m gpu: the persistent greedy-decode chain (csrc/xdec.hip: xdec_dec_kernel -- up to 16 steps of dynamic_decode.py:34-61 over
GreedyDecoderCell.step in ONE launch: LSTM cell, attention, o projection, logits, arg-max, the token fed back, the finished flags and the
early exit, all inside the kernel) against the launch-per-step path it replaces (lxo_shape.step_kernels = 2: csrc/rstep.hip + the attention
pair + lxo_k_argmax), which the other -m gpu tests hold to the oracle and to the reference-code fixtures token for token.

Both compute the same bf16 mathematics; the contraction of a step GEMM is split over 8 waves instead of 4, so a near-tie of two logits may
fall the other way.  Weights that emit END at staggered steps (tests/test_gpu_benchcfg.py: count_set), every batch size the chain takes,
the early exit inside a launch, across launches (LXO_XDEC_DEC_CHUNK), and the step bound (random weights, 152 steps = 10 launches)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_common import *  # noqa
from test_gpu_benchcfg import count_set, V, H, W


@pytest.fixture(scope="module")
def end_params():
    eng = Engine(V, dtype="bf16", seed=0)
    for step in range(260):
        imgs, forms = count_set(16, 100 + step)
        f, l = pad_batch_formulas(forms, V - 2, V - 1)
        eng.train_step(pad_batch_images(imgs), f, l, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    return eng.get_params()


def _decode(params, img, step_kernels, max_iter, seed=0):
    eng = Engine(V, dtype="bf16", seed=seed)
    if params is not None:
        eng.load_params(params)
    eng.step_kernels = step_kernels
    ids = eng.greedy_decode(img, V - 1, max_iter=max_iter)
    used, err = eng.chain_status()
    return ids, used, err


def _first_end(ids):
    return [int(np.argmax(r == V - 1)) if (r == V - 1).any() else -1 for r in ids]


@pytest.mark.parametrize("B", [64, 32, 16, 8])
def test_decode_chain_equals_launch_per_step_early_exit(end_params, B):
    imgs, forms = count_set(B, 300 + B)
    img = pad_batch_images(imgs)
    a, used, err = _decode(end_params, img, 0, 151)
    assert used and err == 0, (used, err)                      # the chain ran (8 x 32 tickets) and no hand-over timed out
    b, used_b, _ = _decode(end_params, img, 2, 151)
    assert not used_b
    first = _first_end(b)
    print("B=%d: %d steps (chain) / %d (launch per step); first END per row: %d distinct positions, %d .. %d" % (
        B, a.shape[1], b.shape[1], len(set(first)), min(first), max(first)))
    assert b.shape[1] < 152 and min(first) >= 0                # every row emitted END: the loop stopped early ...
    assert B < 16 or len(set(first)) >= 3                      # ... at staggered steps
    assert a.shape == b.shape, (a.shape, b.shape)              # the chain stops at the same step (inside its first launch)
    assert (a == b).mean() >= 0.999, np.argwhere(a != b)[:8]   # measured: identical


@pytest.mark.parametrize("chunk", ["3", "7"])
def test_decode_chain_early_exit_across_launches(end_params, chunk):
    """10 steps in launches of 3 / 7: the step after which nothing is unfinished falls in the 4th / 2nd launch; speculative launches behind it
    find the stop word and return"""
    imgs, forms = count_set(16, 12)
    img = pad_batch_images(imgs)
    old = os.environ.get("LXO_XDEC_DEC_CHUNK")
    os.environ["LXO_XDEC_DEC_CHUNK"] = chunk
    try:
        a, used, err = _decode(end_params, img, 0, 151)
    finally:
        if old is None:
            os.environ.pop("LXO_XDEC_DEC_CHUNK", None)
        else:
            os.environ["LXO_XDEC_DEC_CHUNK"] = old
    assert used and err == 0, (used, err)
    b, _, _ = _decode(end_params, img, 2, 151)
    assert a.shape == b.shape and b.shape[1] > int(chunk), (a.shape, b.shape)
    assert (a == b).mean() >= 0.999, np.argwhere(a != b)[:8]


@pytest.mark.parametrize("B,max_iter", [(64, 151), (8, 40), (16, 16), (32, 15), (16, 0)])
def test_decode_chain_at_the_step_bound(B, max_iter):
    """random weights never emit END: max_iter + 1 steps (dynamic_decode.py:38), 152 = nine full launches and one of 8"""
    img, f, l = batch(B, H, W, V, 3, 5, seed=50 + B)
    a, used, err = _decode(None, img, 0, max_iter, seed=4)
    assert used and err == 0, (used, err)
    b, _, _ = _decode(None, img, 2, max_iter, seed=4)
    assert a.shape == b.shape == (B, max_iter + 1), (a.shape, b.shape)
    agree = float((a == b).mean())
    print("B=%d, %d steps at the bound: chain vs launch per step agreement %.4f, %d distinct tokens" % (B, max_iter + 1, agree, len(np.unique(b))))
    assert agree >= 0.99, np.argwhere(a != b)[:8]              # measured: identical


def test_decode_chain_twice_and_after_training(end_params):
    """the same engine: decode, train a step (the training chains share the sync block), decode again -> the same ids, chain both times"""
    imgs, forms = count_set(16, 21)
    img = pad_batch_images(imgs)
    eng = Engine(V, dtype="bf16", seed=0)
    eng.load_params(end_params)
    a = eng.greedy_decode(img, V - 1, max_iter=151)
    assert eng.chain_status() == (True, 0)
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    eng.train_step(img, f, l, 0.0)                             # lr 0: the weights stay
    assert eng.chain_used and eng.chain_failures == 0
    b = eng.greedy_decode(img, V - 1, max_iter=151)
    assert eng.chain_status() == (True, 0)
    assert a.shape == b.shape and np.array_equal(a, b)
    ids, alpha = eng.greedy_decode(img, V - 1, max_iter=151, return_attention=True)      # the attention maps come from the launch-per-step path
    assert ids.shape == a.shape and (ids == a).mean() >= 0.999
    assert np.abs(alpha.reshape(alpha.shape[0], alpha.shape[1], -1).sum(-1) - 1.0).max() < 1e-3


@pytest.mark.parametrize("B", [1, 5, 20, 40])
def test_odd_batches_are_filled_up_to_a_chain_batch(end_params, monkeypatch, B):
    """Engine.greedy_decode fills a batch the chain does not take (B not in {8, 16, 32, 64}) with copies of its own images: the chain runs
    (tickets taken, no error), the real rows' ids and the step count are those of the launch-per-step path on the B images alone."""
    imgs, forms = count_set(B, 500 + B)
    img = pad_batch_images(imgs)
    a, used, err = _decode(end_params, img, 0, 151)
    assert used and err == 0, (used, err)
    monkeypatch.setenv("LXO_DECODE_PAD", "0")
    b, used_b, _ = _decode(end_params, img, 0, 151)
    assert not used_b                                           # B images alone: the launch-per-step kernels
    assert a.shape == b.shape and a.shape[0] == B, (a.shape, b.shape)
    assert (a == b).mean() >= 0.999, np.argwhere(a != b)[:8]
