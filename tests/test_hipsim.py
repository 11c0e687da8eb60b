"""CPU: kernel LOGIC of the shipped HIP sources, executed by the hipsim SIMT interpreter
(tests/hipsim: same .hip files compiled for the host; test infrastructure only).  Compared
with the committed oracle golden vectors.  Real-hardware parity is in the -m gpu tests."""
import ctypes
import os

import numpy as np
import pytest

from simharness import Sim, lib, ptr
from simlib import bf16_to_f32, f32_to_bf16

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_small.npz"))
SMALL = dict(C=128, E=128, U=128, O=128, D=16)      # live-oracle comparisons run at small widths to keep the CPU suite short


@pytest.mark.parametrize("dt,a_f32,c_f32,small,M,N,K", [(0, 1, 1, 0, 70, 40, 64), (1, 0, 0, 0, 130, 140, 128),
                                                       (1, 1, 1, 1, 50, 70, 256), (0, 1, 1, 1, 64, 32, 256)])
def test_gemm_nt(dt, a_f32, c_f32, small, M, N, K):
    L = lib()
    rng = np.random.default_rng(M * N + K)
    A = rng.standard_normal((M, K)).astype(np.float32); B = rng.standard_normal((N, K)).astype(np.float32)
    if dt == 1:
        Bd = f32_to_bf16(B); Bref = bf16_to_f32(Bd)
        Ad = A if a_f32 else f32_to_bf16(A)
        Aref = bf16_to_f32(f32_to_bf16(A))
    else:
        Ad, Bd, Aref, Bref = A, B, A, B
    C = np.zeros((M, N), np.float32 if (dt == 0 or c_f32) else np.uint16)
    assert L.lxo_gemm_nt(dt, a_f32, c_f32, small, ptr(Ad), ptr(Bd), ptr(C), M, N, K, K, K, N, None, 2, ctypes.c_float(0.1), 0, None) == 0
    ref = np.tanh(0.1 * Aref.astype(np.float64) @ Bref.astype(np.float64).T)
    out = C if C.dtype == np.float32 else bf16_to_f32(C)
    assert np.abs(out - ref).max() < (1e-5 if C.dtype == np.float32 else 8e-3)


@pytest.mark.parametrize("dt,a_f32,b_f32", [(0, 1, 1), (1, 0, 0), (1, 1, 0), (1, 0, 1)])
def test_gemm_tn(dt, a_f32, b_f32):
    L = lib()
    M, I, J = 100, 72, 40
    rng = np.random.default_rng(3)
    A = rng.standard_normal((M, I)).astype(np.float32); B = rng.standard_normal((M, J)).astype(np.float32)
    if dt == 1:
        Ad = A if a_f32 else f32_to_bf16(A); Bd = B if b_f32 else f32_to_bf16(B)
        Aref, Bref = bf16_to_f32(f32_to_bf16(A)), bf16_to_f32(f32_to_bf16(B))
    else:
        Ad, Bd, Aref, Bref = A, B, A, B
    Ap = np.concatenate([Ad.reshape(-1), np.zeros(64, Ad.dtype)]); Bp = np.concatenate([Bd.reshape(-1), np.zeros(64, Bd.dtype)])
    C0 = rng.standard_normal((I, J)).astype(np.float32); C = C0.copy()
    assert L.lxo_gemm_tn(dt, a_f32, b_f32, ptr(Ap), ptr(Bp), ptr(C), M, I, J, I, J, J, 2, 1, None) == 0
    ref = C0 + Aref.astype(np.float64).T @ Bref.astype(np.float64)
    assert np.abs(C - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.parametrize("B,H,W,Cin,Cout,valid", [(2, 5, 70, 64, 72, 0), (1, 6, 66, 128, 128, 1), (3, 3, 20, 64, 136, 0)])
def test_conv3x3_wgrad_bf16(B, H, W, Cin, Cout, valid):
    """conv_wgrad_kernel under the SIMT interpreter: LDS-DMA through buffer resources (zero padding and ragged tiles = out-of-range
    offsets), transposing LDS reads, pixel ranges of unequal length, f32 atomics -- against a float64 correlation of the same
    bf16-rounded operands (partial pixel tiles in both directions, a partial 128-channel tile, SAME and VALID)."""
    L = lib()
    rng = np.random.default_rng(H * W + Cin + Cout)
    x = bf16_to_f32(f32_to_bf16(rng.standard_normal((B, H, W, Cin)).astype(np.float32)))
    Ho, Wo = (H - 2, W - 2) if valid else (H, W)
    pad = 0 if valid else 1
    dy = bf16_to_f32(f32_to_bf16(rng.standard_normal((B, Ho, Wo, Cout)).astype(np.float32)))
    dw0 = rng.standard_normal((9 * Cin, Cout)).astype(np.float32); dw = dw0.copy()
    assert L.lxo_conv3x3_wgrad(1, ptr(f32_to_bf16(x)), ptr(f32_to_bf16(dy)), ptr(dw), B, H, W, Cin, Ho, Wo, Cout, pad, None) == 0, L.lxo_last_error()
    xp = np.pad(x.astype(np.float64), ((0, 0), (pad, pad), (pad, pad), (0, 0)))
    ref = dw0.astype(np.float64)
    for kh in range(3):
        for kw in range(3):
            ref[(kh * 3 + kw) * Cin:(kh * 3 + kw + 1) * Cin] += np.einsum("byxi,byxo->io", xp[:, kh:kh + Ho, kw:kw + Wo, :], dy.astype(np.float64))
    assert np.abs(dw - ref).max() / np.abs(ref).max() < 2e-6


def _run(dtype, deterministic=0):
    img, f, l = GOLD["img"], GOLD["formula"], GOLD["lengths"]
    S = Sim(2, 32, 48, f.shape[1], 11, dtype=dtype, seed=0)
    S.shape.deterministic = deterministic
    if deterministic:      # the ordered-partials scratch (ws region "det_part") is part of this shape's workspace
        S.ws = np.zeros(S.L.lxo_workspace_bytes(ctypes.byref(S.shape)) + 256, np.uint8)
    S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
    S.ck(S.L.lxo_decoder_train_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), None), "dec")
    S.ck(S.L.lxo_ce_loss_fwd_bwd(S.sref(), ptr(S.ws), ptr(f), ptr(l), ctypes.c_float(1.0 / int(l.sum())), None), "loss")
    return S, img, f, l


def test_forward_backward_f32_vs_golden():
    lib().lxo_set_side_stream(ctypes.c_void_p(0))
    _check_fwd_bwd()


def _check_fwd_bwd():
    S, img, f, l = _run(0)
    T = f.shape[1]
    logits = S.region("logits", np.float32, (T, 2, 32))[:, :, :11].transpose(1, 0, 2)
    assert np.abs(logits - GOLD["logits"]).max() < 2e-5
    st = S.region("loss", np.float32)[:2]
    assert abs(st[0] / st[1] - float(GOLD["loss"])) < 2e-6 and st[1] == int(GOLD["n_words"])
    alpha = S.region("alpha", np.float32, (T, 2, 8))[:, :, :8].transpose(1, 0, 2)
    assert np.abs(alpha - GOLD["alpha"]).max() < 1e-6
    S.ck(S.L.lxo_decoder_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), ptr(S.grads), None), "decbwd")
    S.ck(S.L.lxo_encoder_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 1, None), "encbwd")
    checked = 0
    for k, _, _ in S.specs:
        key = k.replace("/", "__")
        if key in GOLD.files:
            g, r = S.grad(k), GOLD[key]
            assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1e-6) + 1e-9, k
            checked += 1
    assert checked >= 9


def test_row_bilstm_encoder_vs_its_specification():
    """The optional row-BiLSTM encoder (lxo_shape.encoder_rnn; not in the reference) against oracle/ref_model.py:row_bilstm:
    loss and the gradients of the four new variables and of a conv kernel upstream of it, f32."""
    import torch
    from oracle import ref_model as R
    img, f, l = GOLD["img"], GOLD["formula"], GOLD["lengths"]
    dims = dict(C=256, E=128, U=128, O=128, D=16, row_bilstm=True)     # the smallest widths the row encoder admits (C in {256, 512})
    S = Sim(2, 32, 48, f.shape[1], 11, dtype=0, seed=2, dims=dims)
    P = {k: torch.from_numpy(v.copy()) for k, v in S.P.items()}
    S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
    S.ck(S.L.lxo_decoder_train_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), None), "dec")
    S.ck(S.L.lxo_ce_loss_fwd_bwd(S.sref(), ptr(S.ws), ptr(f), ptr(l), ctypes.c_float(1.0 / int(l.sum())), None), "loss")
    S.ck(S.L.lxo_decoder_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), ptr(S.grads), None), "decbwd")
    S.ck(S.L.lxo_encoder_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 1, None), "encbwd")
    loss, G, _, _ = R.train_grads(P, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l))
    st = S.region("loss", np.float32)[:2]
    assert abs(st[0] / st[1] - float(loss)) < 2e-5 * float(loss)
    row = "Encoder/row_encoder/bidirectional_rnn/"
    for k in (row + "fw/lstm_cell/kernel", row + "fw/lstm_cell/bias", row + "bw/lstm_cell/kernel", row + "bw/lstm_cell/bias",
              "Encoder/convolutional_encoder/conv2d_5/kernel", "Encoder/convolutional_encoder/conv2d_2/bias", "Decoder/AttentionCell/att_img/kernel"):
        g, r = S.grad(k), G[k].numpy()
        assert np.abs(g - r).max() <= 5e-5 * max(np.abs(r).max(), 1e-6) + 1e-9, (k, np.abs(g - r).max(), np.abs(r).max())


def test_forward_backward_bf16_default_widths_vs_golden():
    """bf16 mode at the shipped widths (E = 256, C = 512): the loss within north_star's 1e-3 bar, then every gradient the golden file
    holds against the f32 oracle's, under the SIMT interpreter.  Runs the E-domain attention (region att_exp = e^{2 att_img}: the
    attention backward forms 1 - tanh^2 from one reciprocal per element)."""
    S, img, f, l = _run(1)
    st = S.region("loss", np.float32)[:2]
    assert abs(st[0] / st[1] - float(GOLD["loss"])) / float(GOLD["loss"]) < 1e-3       # north_star loss bar (forward, bf16)
    S.ck(S.L.lxo_decoder_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), ptr(S.grads), None), "decbwd")
    S.ck(S.L.lxo_encoder_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 1, None), "encbwd")
    checked = 0
    for k, _, _ in S.specs:
        key = k.replace("/", "__")
        if key in GOLD.files:
            g, r = S.grad(k).reshape(-1).astype(np.float64), GOLD[key].reshape(-1).astype(np.float64)
            c = float(g @ r) / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-300)
            assert c > 0.995, (k, c)
            checked += 1
    assert checked >= 9


def test_greedy_and_beam_f32_vs_golden():
    img = GOLD["img"]
    S = Sim(2, 32, 48, 1, 11, dtype=0, seed=0, max_steps=9)
    S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
    ids = np.zeros((2, 9), np.int32); steps = ctypes.c_int(0)
    S.ck(S.L.lxo_greedy_decode(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), 10, 8, ptr(ids), ctypes.byref(steps), None), "greedy")
    assert steps.value == GOLD["greedy_ids"].shape[1]
    assert np.array_equal(ids[:, :steps.value], GOLD["greedy_ids"])
    S2 = Sim(2, 32, 48, 1, 11, dtype=0, seed=0, beam=2, max_steps=9)
    S2.ck(S2.L.lxo_encoder_fwd(S2.sref(), ptr(S2.params), ptr(S2.wpack), ptr(S2.ws), ptr(img), None), "enc")
    bids = np.zeros((2, 9, 2), np.int32); bpar = np.zeros((2, 9, 2), np.int32)
    S2.ck(S2.L.lxo_beam_decode(S2.sref(), ptr(S2.params), ptr(S2.wpack), ptr(S2.ws), 10, 8, ptr(bids), ptr(bpar), ctypes.byref(steps), None), "beam")
    n = steps.value
    assert n == GOLD["beam_ids"].shape[1]
    assert np.array_equal(bids[:, :n], GOLD["beam_ids"]) and np.array_equal(bpar[:, :n], GOLD["beam_parents"])


def test_decode_one_step_at_a_time_equals_the_loops():
    """lxo_decode_begin / lxo_decode_step (what the cell protocol of model/components is bound to) against the same golden ids
    as the device-side loops, with the host doing dynamic_decode.py:38-51's loop."""
    img = GOLD["img"]
    for beam, want_ids, want_par in ((1, GOLD["greedy_ids"], None), (2, GOLD["beam_ids"], GOLD["beam_parents"])):
        S = Sim(2, 32, 48, 1, 11, dtype=0, seed=0, beam=beam, max_steps=9)
        S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
        S.ck(S.L.lxo_decode_begin(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), None), "begin")
        shp = (2, 9) if beam == 1 else (2, 9, beam)
        ids = np.zeros(shp, np.int32); par = np.zeros(shp, np.int32); fin = np.zeros(2 * beam, np.int32)
        time, finished = 0, np.zeros(2 * beam, bool)
        while not finished.all():
            un = ctypes.c_int(-1)
            S.ck(S.L.lxo_decode_step(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), 10, time, ptr(ids), ptr(par) if beam > 1 else None,
                                     ptr(fin), ctypes.byref(un), None), "step")
            assert un.value == int((fin == 0).sum())
            finished = np.logical_or(time >= 8, fin.astype(bool))
            time += 1
        assert time == want_ids.shape[1]
        assert np.array_equal(ids[:, :time], want_ids)
        if want_par is not None:
            assert np.array_equal(par[:, :time], want_par)


def test_cell_step_from_a_state_of_the_callers_choosing():
    """AttentionCell.step(embedding, state) of the reference takes ANY state (attention_cell.py:58).  lxo_decode_state_get / _set move the
    AttentionState (c, h, o) and the fed-back ids across the boundary and lxo_decode_cell_step runs the cell alone: a cell step from a
    state read back after step 1 of one decode, re-installed into a FRESH decode, must give the logits the original step 2 produced,
    bit for bit (f32 parity mode)."""
    img = GOLD["img"]
    S = Sim(2, 32, 48, 1, 11, dtype=0, seed=0, max_steps=9)
    U, O, Vp = S.dims["U"], S.dims["O"], 32
    S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
    S.ck(S.L.lxo_decode_begin(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), None), "begin")
    ids = np.zeros((2, 9), np.int32); fin = np.zeros(2, np.int32)
    logits = []
    for time in range(3):
        S.ck(S.L.lxo_decode_step(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), 10, time, ptr(ids), None, ptr(fin), None, None), "step")
        logits.append(S.region("dec_logits", np.float32, (2, Vp)).copy())
        if time == 1:       # the state step 2 starts from, and the ids it is fed
            c = np.zeros((2, U), np.float32); h = np.zeros((2, U), np.float32); o = np.zeros((2, O), np.float32)
            S.ck(S.L.lxo_decode_state_get(S.sref(), ptr(S.ws), 2, ptr(c), ptr(h), ptr(o), None), "get")
            fed = ids[:, 1].copy()
    assert np.abs(c).max() > 0 and np.abs(h).max() > 0 and np.abs(o).max() > 0
    S2 = Sim(2, 32, 48, 1, 11, dtype=0, seed=0, max_steps=9)
    S2.ck(S2.L.lxo_encoder_fwd(S2.sref(), ptr(S2.params), ptr(S2.wpack), ptr(S2.ws), ptr(img), None), "enc")
    S2.ck(S2.L.lxo_decode_begin(S2.sref(), ptr(S2.params), ptr(S2.wpack), ptr(S2.ws), None), "begin")
    S2.ck(S2.L.lxo_decode_state_set(S2.sref(), ptr(S2.ws), 2, ptr(c), ptr(h), ptr(o), ptr(np.ascontiguousarray(fed)), None), "set")
    S2.ck(S2.L.lxo_decode_cell_step(S2.sref(), ptr(S2.params), ptr(S2.wpack), ptr(S2.ws), 2, 0, None), "cell_step")
    got = S2.region("dec_logits", np.float32, (2, Vp))
    assert np.array_equal(got[:, :11], logits[2][:, :11])
    # ... and the start-token form from the initial state gives step 0's logits
    S2.ck(S2.L.lxo_decode_begin(S2.sref(), ptr(S2.params), ptr(S2.wpack), ptr(S2.ws), None), "begin")
    S2.ck(S2.L.lxo_decode_cell_step(S2.sref(), ptr(S2.params), ptr(S2.wpack), ptr(S2.ws), 0, 1, None), "cell_step")
    assert np.array_equal(S2.region("dec_logits", np.float32, (2, Vp))[:, :11], logits[0][:, :11])
    # the state after that step, read back, is the one the first decode had after ITS step 0
    c1 = np.zeros((2, U), np.float32)
    S2.ck(S2.L.lxo_decode_state_get(S2.sref(), ptr(S2.ws), 1, ptr(c1), None, None, None), "get")
    assert np.abs(c1).max() > 0


@pytest.mark.parametrize("dtype", [0, 1])
def test_chain_guard_drops_a_poisoned_step(dtype):
    """lxo_chain_guard folds the decoder chains' error words (ws region "xdec_sync", include/lxo.h) and the NaN probe element of the
    gradients into the optimizer's scale; a NaN scale makes lxo_adam_step / lxo_optimizer_step leave every buffer untouched."""
    S = Sim(2, 32, 48, 3, 11, dtype=dtype, dims=SMALL, seed=0)
    L = S.L
    n = S.params.size
    rng = np.random.default_rng(1)
    g = rng.standard_normal(n).astype(np.float32)
    sc = np.zeros(2 + 1024, np.float32); st = np.full(4, 9, np.uint32)
    # healthy: scale 1 without clip, the clip scale with it
    S.ck(L.lxo_chain_guard(S.sref(), ptr(S.ws), ptr(g), ptr(sc), 0, ptr(st), None), "guard")
    assert sc[0] == 1.0 and list(st[:3]) == [0, 0, 0]
    sc[0] = 0.25
    S.ck(L.lxo_chain_guard(S.sref(), ptr(S.ws), ptr(g), ptr(sc), 1, ptr(st), None), "guard")
    assert sc[0] == 0.25
    # another rank's failure arrives as a NaN in the last gradient element
    g2 = g.copy(); g2[-1] = np.nan
    S.ck(L.lxo_chain_guard(S.sref(), ptr(S.ws), ptr(g2), ptr(sc), 1, ptr(st), None), "guard")
    assert np.isnan(sc[0]) and list(st[:3]) == [0, 0, 1]
    p = rng.standard_normal(n).astype(np.float32); m = rng.standard_normal(n).astype(np.float32); v = np.abs(rng.standard_normal(n)).astype(np.float32)
    p0, m0, v0 = p.copy(), m.copy(), v.copy()
    assert L.lxo_adam_step(n, ptr(p), ptr(g), ptr(m), ptr(v), ctypes.c_float(1e-3), ctypes.c_float(0.9), ctypes.c_float(0.999),
                           ctypes.c_float(1e-8), ptr(sc), None) == 0
    assert np.array_equal(p, p0) and np.array_equal(m, m0) and np.array_equal(v, v0)
    for mode in (1, 2, 3):
        assert L.lxo_optimizer_step(mode, n, ptr(p), ptr(g), ptr(v), ctypes.c_float(0.01), ptr(sc), None) == 0
        assert np.array_equal(p, p0) and np.array_equal(v, v0)
    if dtype == 1:
        # this rank's own chains: the error words (forward block 0, backward block 1)
        w = S.region("xdec_sync", np.uint32)
        for blk, want in ((0, [5, 0, 1]), (1, [0, 6, 1])):
            w[:] = 0
            w[blk * (_abi_consts()[0] // 4) + _abi_consts()[1]] = want[0] or want[1]
            sc[0] = 0.5
            S.ck(L.lxo_chain_guard(S.sref(), ptr(S.ws), ptr(g), ptr(sc), 1, ptr(st), None), "guard")
            assert np.isnan(sc[0]) and list(st[:3]) == want
        w[:] = 0


def _abi_consts():
    from latex_ocr_amd import _abi
    return _abi.LXO_XDEC_BLOCK_BYTES, _abi.LXO_XDEC_ERR_WORD


def test_bf16_deterministic_mode_gives_the_default_modes_gradients():
    """lxo_shape.deterministic (bf16): every float-atomic epilogue replaced by ordered partial slots (conv weight gradients through the
    slab + ordered pass of conv_wgrad.hip, bias sums through colsum_part, the dense weight gradients with one row range per tile, conv1 on
    the slot-writing kernel, d_beta / loss / embedding scatter on the f32 mode's ordered paths).  Same mathematics, another summation order:
    loss identical, every gradient equal to the default bf16 mode's within rounding of the sums.  (That two RUNS agree bit for bit can only
    be shown on the GPU -- the interpreter runs workgroups one after the other, so its atomics are ordered anyway: tests/test_gpu_determinism.py.)"""
    S0, img, f, l = _run(1)
    S1, _, _, _ = _run(1, deterministic=1)
    a, b = S0.region("loss", np.float32)[:2].copy(), S1.region("loss", np.float32)[:2].copy()
    assert a[1] == b[1] and abs(a[0] - b[0]) <= 1e-6 * abs(a[0])
    for S in (S0, S1):
        S.ck(S.L.lxo_decoder_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), ptr(S.grads), None), "decbwd")
        S.ck(S.L.lxo_encoder_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 1, None), "encbwd")
    worst = 0.0
    for k, shp, _ in S0.specs:
        g0, g1 = S0.grad(k).astype(np.float64), S1.grad(k).astype(np.float64)
        assert np.isfinite(g1).all(), k
        d = np.abs(g0 - g1).max() / max(np.abs(g0).max(), 1e-30)
        worst = max(worst, d)
        # summation order only -- except the conv bias gradients whose fused sums add the f32 values BEFORE they are rounded to bf16
        # (conv dgrad epilogues, the d_img GEMM): the ordered pass adds the stored bf16 values (2^-9 relative per addend, averaging out)
        assert d < (3e-3 if (k.startswith("Encoder/") and k.endswith("/bias")) else 2e-5), (k, d)
    print("bf16 deterministic vs default: worst relative gradient difference %.2e" % worst)


def test_adam_and_clip():
    L = lib()
    rng = np.random.default_rng(0)
    n = 1000
    p = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32); sc = np.zeros(2 + 1024, np.float32)      # LXO_GNORM_FLOATS
    p0 = p.copy()
    assert L.lxo_global_norm_scale(n, ptr(g), ctypes.c_float(5.0), ptr(sc), None) == 0
    gn = np.sqrt((g.astype(np.float64) ** 2).sum())
    assert abs(sc[1] - gn) / gn < 1e-6 and abs(sc[0] - 5.0 / max(gn, 5.0)) < 1e-6
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert L.lxo_adam_step(n, ptr(p), ptr(g), ptr(m), ptr(v), ctypes.c_float(lr_t), ctypes.c_float(0.9), ctypes.c_float(0.999),
                           ctypes.c_float(1e-8), ptr(sc), None) == 0
    gs = g * sc[0]
    want = p0 - lr_t * (0.1 * gs) / (np.sqrt(0.001 * gs * gs) + 1e-8)
    assert np.abs(p - want).max() < 1e-6


@pytest.mark.parametrize("method,mode,init", [("sgd", 1, 0.0), ("adagrad", 2, 0.1), ("rmsprop", 3, 1.0)])
def test_other_optimizers(method, mode, init):
    import torch
    from oracle import ref_model as R
    L = lib()
    rng = np.random.default_rng(mode)
    n = 777
    p = rng.standard_normal(n).astype(np.float32); slot = np.full(n, init, np.float32)
    P = {"w": torch.from_numpy(p.copy())}
    opt = R.SimpleOptTF(P, method)
    for it in range(3):
        g = rng.standard_normal(n).astype(np.float32)
        assert L.lxo_optimizer_step(mode, n, ptr(p), ptr(g), ptr(slot), ctypes.c_float(0.01), None, None) == 0
        opt.step(P, {"w": torch.from_numpy(g)}, 0.01)
    assert np.abs(p - P["w"].numpy()).max() < 1e-6


@pytest.mark.parametrize("dual", [0, 1])
def test_dropout_forward_backward_vs_oracle(dual):
    """config.dropout < 1 (attention_cell.py:72,83): same counter-based masks in the kernels and in the oracle.
    dual = 1 also covers the half-batch / side-stream code path of the recurrent loops (streams are no-ops under hipsim)."""
    import torch
    from oracle import ref_model as R
    img, f, l = GOLD["img"], GOLD["formula"], GOLD["lengths"]
    keep, seed = 0.7, 12345
    S = Sim(2, 32, 48, f.shape[1], 11, dtype=0, seed=0, dims=SMALL)
    S.shape.keep_prob, S.shape.dropout_seed = keep, seed
    lib().lxo_set_side_stream(ctypes.c_void_p(1 if dual else 0))
    try:
        S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
        S.ck(S.L.lxo_decoder_train_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), None), "dec")
        S.ck(S.L.lxo_ce_loss_fwd_bwd(S.sref(), ptr(S.ws), ptr(f), ptr(l), ctypes.c_float(1.0 / int(l.sum())), None), "loss")
        S.ck(S.L.lxo_decoder_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), ptr(S.grads), None), "decbwd")
        S.ck(S.L.lxo_encoder_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 1, None), "encbwd")
    finally:
        lib().lxo_set_side_stream(ctypes.c_void_p(0))
    P = {k: torch.from_numpy(np.asarray(v)) for k, v in S.P.items()}
    loss, G, _, _ = R.train_grads(P, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l), dropout=(keep, seed))
    loss0 = float(R.forward_loss(P, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l))[0])
    st = S.region("loss", np.float32)[:2]
    assert abs(st[0] / st[1] - float(loss)) < 5e-6
    assert abs(float(loss) - loss0) > 1e-4            # the masks did something
    for k, _, _ in S.specs:
        g, r = S.grad(k), G[k].numpy()
        assert np.abs(g - r).max() <= 3e-5 * max(np.abs(r).max(), 1e-6) + 1e-9, k


@pytest.mark.parametrize("gamma,prob", [(0.5, 1.0), (0.3, 0.5)])
def test_beam_diversity_penalty_vs_oracle(gamma, prob):
    """add_div_penalty (beam_search_decoder_cell.py:258-287): rank penalty with shared Bernoulli draws."""
    import torch
    from oracle import ref_model as R
    img = GOLD["img"]
    S = Sim(2, 32, 48, 1, 11, dtype=0, seed=0, beam=3, max_steps=9, dims=SMALL)
    S.shape.div_gamma, S.shape.div_prob, S.shape.div_seed = gamma, prob, 4
    S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
    bids = np.zeros((2, 9, 3), np.int32); bpar = np.zeros((2, 9, 3), np.int32); steps = ctypes.c_int(0)
    S.ck(S.L.lxo_beam_decode(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), 10, 8, ptr(bids), ptr(bpar), ctypes.byref(steps), None), "beam")
    P = {k: torch.from_numpy(np.asarray(v)) for k, v in S.P.items()}
    ids, par = R.beam_decode(P, torch.from_numpy(img), 10, 3, max_iter=8, div_gamma=gamma, div_prob=prob, div_seed=4)
    n = steps.value
    assert n == ids.shape[1]
    assert np.array_equal(bids[:, :n], ids.numpy()) and np.array_equal(bpar[:, :n], par.numpy())


def _conv_ex_ref(x, w, bias, pad, relu, addend, mask):
    """float64 reference of lxo_conv3x3_ex on bf16-rounded operands: returns (out_pre, out, colsum)."""
    B, H, W, Cin = x.shape
    Cout = w.shape[3]
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    xp = np.zeros((B, H + 2 * pad, W + 2 * pad, Cin), np.float64)
    xp[:, pad:pad + H, pad:pad + W] = x
    y = np.zeros((B, Ho, Wo, Cout), np.float64)
    for kh in range(3):
        for kw in range(3):
            y += xp[:, kh:kh + Ho, kw:kw + Wo] @ w[kh, kw].astype(np.float64)
    y += bias
    if relu:
        y = np.maximum(y, 0)
    pre = y.copy()
    if addend is not None:
        y = (y.reshape(B, -1, Cout) + addend[None]).reshape(B, Ho, Wo, Cout)
    if mask is not None:
        y = np.where(mask > 0, y, 0.0)
    return pre, y, y.reshape(-1, Cout).sum(0)


@pytest.mark.parametrize("Cout,H,W,pad", [(256, 9, 35, 1), (256, 10, 34, 0), (128, 6, 70, 1), (64, 5, 67, 1)])
def test_conv3x3_ex_full_epilogue_bf16(Cout, H, W, pad):
    """fused conv epilogue (bias, ReLU, pre-addend copy, timing-signal addend, ReLU mask, bias-gradient column sums)
    on the 8x32x256, 4x64x128 and 4x64x64 halo kernels."""
    L = lib()
    rng = np.random.RandomState(Cout + H)
    B, Cin = 2, 64
    rb = lambda a: bf16_to_f32(f32_to_bf16(a.astype(np.float32)))
    x = rb(rng.randn(B, H, W, Cin)); w = rb(rng.randn(3, 3, Cin, Cout) * 0.05); bias = rng.randn(Cout).astype(np.float32)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    addend = rng.randn(Ho * Wo, Cout).astype(np.float32)
    mask = rb(rng.randn(B, Ho, Wo, Cout))
    xb, wb, mb = f32_to_bf16(x), f32_to_bf16(np.ascontiguousarray(w.reshape(9 * Cin, Cout).T)), f32_to_bf16(mask)
    out = np.zeros((B, Ho, Wo, Cout), np.uint16); pre = np.zeros_like(out); cs = np.zeros(Cout, np.float32)
    rc = L.lxo_conv3x3_ex(1, ptr(xb), ptr(wb), ptr(bias), ptr(out), B, H, W, Cin, Ho, Wo, Cout, pad, 1, ptr(addend), Ho * Wo,
                          ptr(pre), ptr(mb), ptr(cs), None)
    assert rc == 0, L.lxo_last_error()
    rpre, rout, rcs = _conv_ex_ref(x, w, bias, pad, 1, addend, mask)
    s = np.abs(rout).max()
    assert np.abs(bf16_to_f32(pre) - rpre).max() / s < 6e-3
    assert np.abs(bf16_to_f32(out) - rout).max() / s < 6e-3
    assert np.abs(cs - rcs).max() / np.abs(rcs).max() < 2e-4


def test_greedy_attention_export_vs_oracle():
    """lxo_greedy_decode_attn: per-step attention weights (the reference's py_func tap, attention_mechanism.py:96-105)."""
    import torch
    from oracle import ref_model as R
    img = GOLD["img"]
    S = Sim(2, 32, 48, 1, 11, dtype=0, seed=0, max_steps=9, dims=SMALL)
    S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
    ids = np.zeros((2, 9), np.int32); steps = ctypes.c_int(0)
    alpha = np.zeros((9, 2, 8), np.float32)                      # R = 2 x 4 regions, Rp = 8
    S.ck(S.L.lxo_greedy_decode_attn(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), 10, 8, ptr(ids), ptr(alpha), ctypes.byref(steps), None), "greedy")
    P = {k: torch.from_numpy(np.asarray(v)) for k, v in S.P.items()}
    rid, ralpha = R.greedy_decode(P, torch.from_numpy(img), 10, max_iter=8, return_alpha=True)
    n = steps.value
    assert np.array_equal(ids[:, :n], rid.numpy())
    assert np.abs(alpha[:n].transpose(1, 0, 2) - ralpha.numpy()).max() < 1e-6


@pytest.mark.parametrize("dtype", [0, 1])
def test_encoder_cnn_variant_and_no_positional_vs_oracle(dtype):
    """encoder_cnn == "cnn" (encoder.py:54-56: no late pools, (2,4) stride-2 SAME conv) with positional_embeddings
    false (encoder.py:60-65 skipped): forward loss and every gradient against the oracle."""
    import torch
    from oracle import ref_model as R
    img, f, l = GOLD["img"], GOLD["formula"], GOLD["lengths"]
    dims = dict(C=128, E=128, U=128, O=128, D=16, cnn=True, positional=False)
    S = Sim(2, 32, 48, f.shape[1], 11, dtype=dtype, seed=2, dims=dims)
    S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
    S.ck(S.L.lxo_decoder_train_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), None), "dec")
    S.ck(S.L.lxo_ce_loss_fwd_bwd(S.sref(), ptr(S.ws), ptr(f), ptr(l), ctypes.c_float(1.0 / int(l.sum())), None), "loss")
    S.ck(S.L.lxo_decoder_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), ptr(S.grads), None), "decbwd")
    S.ck(S.L.lxo_encoder_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 5, None), "encbwd")
    S.ck(S.L.lxo_encoder_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 4, 1, None), "encbwd")
    P = {k: torch.from_numpy(np.asarray(v)) for k, v in S.P.items()}
    assert "Encoder/convolutional_encoder/conv2d_6/kernel" in P and P["Encoder/convolutional_encoder/conv2d_5/kernel"].shape == (2, 4, 128, 128)
    loss, G, _, _ = R.train_grads(P, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l), positional=False)
    st = S.region("loss", np.float32)[:2]
    if dtype == 0:
        assert abs(st[0] / st[1] - float(loss)) < 5e-6
        for k, _, _ in S.specs:
            g, r = S.grad(k), G[k].numpy()
            assert np.abs(g - r).max() <= 3e-5 * max(np.abs(r).max(), 1e-6) + 1e-9, k
    else:
        assert abs(st[0] / st[1] - float(loss)) / float(loss) < 1e-3
        for k, _, _ in S.specs:
            g, r = S.grad(k).reshape(-1), G[k].numpy().reshape(-1)
            if np.abs(r).max() > 0:
                assert float(g @ r) / (np.linalg.norm(g) * np.linalg.norm(r) + 1e-30) > 0.97, k


def test_dead_padding_rows_live_B_same_loss_and_gradients():
    """lxo_shape.live_B (include/lxo.h): a batch filled up with DEAD rows -- formula length 0, so outside the loss mask of img2seq.py:68-71 -- gives the
    loss and the gradients of the batch it was filled up from; the encoder reads the live images only (the image buffer handed over here HAS only the
    live ones) and leaves zero features for the dead row.  f32, small widths: 2 live samples in a batch of 3 against the 2 alone."""
    img, f, l = GOLD["img"], GOLD["formula"], GOLD["lengths"]
    T = f.shape[1]
    res = []
    for pad in (0, 1):
        B = 2 + pad
        S = Sim(B, 32, 48, T, 11, dtype=0, seed=0, dims=SMALL)
        ff = np.ascontiguousarray(np.concatenate([f, f[:pad]]))                     # the dead row: any valid token ids
        ll = np.ascontiguousarray(np.concatenate([l, np.zeros(pad, l.dtype)]))      # ... and length 0
        S.shape.live_B = 2 if pad else 0
        S.ck(S.L.lxo_encoder_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), None), "enc")
        S.ck(S.L.lxo_decoder_train_fwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(ff), None), "dec")
        S.ck(S.L.lxo_ce_loss_fwd_bwd(S.sref(), ptr(S.ws), ptr(ff), ptr(ll), ctypes.c_float(1.0 / int(l.sum())), None), "loss")
        S.ck(S.L.lxo_decoder_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(ff), ptr(S.grads), None), "decbwd")
        S.ck(S.L.lxo_encoder_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 1, None), "encbwd")
        st = S.region("loss", np.float32)[:2].copy()
        if pad:
            R = S.region("img", np.float32).size // B
            assert np.all(S.region("img", np.float32)[2 * R:3 * R] == 0.0)                # the dead row's features
        res.append((st, {k: S.grad(k).copy() for k, _, _ in S.specs}))
    assert res[0][0][1] == res[1][0][1] == int(l.sum())                                    # n_words does not see the dead row
    assert abs(res[0][0][0] - res[1][0][0]) <= 1e-6 * abs(res[0][0][0])
    for k in res[0][1]:
        a0, a1 = res[0][1][k], res[1][1][k]
        assert np.abs(a0 - a1).max() <= 2e-6 * max(np.abs(a0).max(), 1e-6) + 1e-10, k


def test_train_bwd_and_ready_entry_points_vs_golden():
    """lxo_encoder_bwd_ready (one call for a layer range, with a per-layer event table whose NULL entries are skipped) behind
    lxo_decoder_train_bwd against the golden gradients, and lxo_train_bwd (the backward pass in one call) from the same forward state: the
    same kernels in the same order on one stream -- every gradient bit-identical.  f32."""
    S, img, f, l = _run(0)
    S.grads[:] = 0
    S.ck(S.L.lxo_decoder_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), ptr(S.grads), None), "decbwd")
    table = (ctypes.c_void_p * 7)()
    S.ck(S.L.lxo_encoder_bwd_ready(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 1, table, None), "encbwd_ready")
    checked = 0
    for k, _, _ in S.specs:
        key = k.replace("/", "__")
        if key in GOLD.files:
            g, r = S.grad(k), GOLD[key]
            assert np.abs(g - r).max() <= 2e-5 * max(np.abs(r).max(), 1e-6) + 1e-9, k
            checked += 1
    assert checked >= 9
    first = S.grads.copy()
    S.grads[:] = 0
    S.ck(S.L.lxo_train_bwd(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(f), ptr(img), ptr(S.grads), None, None), "train_bwd")
    assert first.tobytes() == S.grads.tobytes()
    assert S.L.lxo_encoder_bwd_ready(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 6, 1, None, None) != 0     # no table
    assert S.L.lxo_encoder_bwd_ready(S.sref(), ptr(S.params), ptr(S.wpack), ptr(S.ws), ptr(img), ptr(S.grads), 7, 1, table, None) != 0   # layer range
