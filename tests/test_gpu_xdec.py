"""-m gpu: the persistent XCD-local decoder chain (csrc/xdec.hip: the T teacher-forced steps of AttentionCell.step in ONE launch, 8 chains
of B / 8 samples, hand-over through each XCD's L2) against the launch-per-step chain it replaces (csrc/rstep.hip + the attention pair,
lxo_shape.step_kernels = 2), which is itself held to the oracle and to the reference-code fixtures by the other -m gpu tests.

Both compute the same mathematics in bf16 mode (same operand roundings, same transcendental forms); the contraction of a step GEMM is
split over 8 waves instead of 4, so sums differ in the last f32 bits and a bf16 mirror may round the other way here and there.  Every
batch size the chain takes (8, 16, 32, 64: one to eight samples per XCD), dropout, and the error word / ticket counters of the chain."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_common import *  # noqa

V = 120


def _run(step_kernels, B, H, W, dropout=None, seed=5):
    img, f, l = batch(B, H, W, V, 5, 24, seed=77 + B)
    eng = Engine(V, dtype="bf16", seed=seed)
    eng.step_kernels = step_kernels
    eng.forward(img, f, dropout=dropout)
    stats = eng.loss(l, 1.0 / int(l.sum())).cpu().numpy().copy()
    torch.cuda.synchronize()
    T = f.shape[1]
    Bq = img.shape[0]
    out = {"stats": stats,
           "logits": eng.region("logits", "f32", (T, Bq, (V + 31) // 32 * 32))[:, :, :V].cpu().numpy().copy(),
           "rec": eng.region("rec", "f32", (T + 1, Bq, 2048)).cpu().numpy().copy(),
           "cs": eng.region("cs", "f32", (T + 1, Bq, 512)).cpu().numpy().copy(),
           "gates": eng.region("gates", "f32", (T, Bq, 2048)).cpu().numpy().copy(),
           "att_h": eng.region("att_h", "f32", (T, Bq, 256)).cpu().numpy().copy()}
    R = (-(-H // 8) - 2) * (-(-W // 8) - 2)
    Rp = (R + 7) // 8 * 8
    out["alpha"] = eng.region("alpha", "f32", (T, Bq, Rp))[:, :, :R].cpu().numpy().copy()
    used, err = eng.chain_status()
    eng.backward()
    torch.cuda.synchronize()
    return out, eng.grad_dict(), used, err


@pytest.mark.parametrize("B,H,W", [(64, 64, 256), (32, 48, 200), (16, 64, 128), (8, 40, 150)])
def test_chain_equals_launch_per_step(B, H, W):
    a, ga, used, err = _run(0, B, H, W)
    assert used and err == 0, (used, err)                      # the chain ran: 8 x 32 tickets taken, no barrier timed out
    b, gb, used_b, _ = _run(2, B, H, W)
    assert not used_b
    assert a["stats"][1] == b["stats"][1]
    la, lb = a["stats"][0] / a["stats"][1], b["stats"][0] / b["stats"][1]
    assert abs(la - lb) <= 2e-5 * abs(lb), (la, lb)
    worst = {}
    for k in ("logits", "rec", "cs", "gates", "att_h", "alpha"):
        d = np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-30)
        worst[k] = float(d)
        # a bf16 mirror that rounds the other way moves a downstream value by ~2^-9 of its operand; states stay within a few of those
        assert d < 2e-2, (k, d)
        assert cosine(a[k], b[k]) > 0.99999, (k, cosine(a[k], b[k]))
    assert np.abs(a["alpha"].sum(-1) - 1.0).max() < 1e-4       # every step's attention weights are a distribution
    wc = (1.0, None)
    for k in ga:
        c = cosine(ga[k], gb[k])
        if c < wc[0]:
            wc = (c, k)
        assert c > 0.9999, (k, c)
    print("B=%d %dx%d: chain vs launch-per-step: loss %.6f vs %.6f; max rel %s; worst gradient cosine %.7f (%s)" % (
        B, H, W, la, lb, {k: "%.1e" % v for k, v in worst.items()}, wc[0], wc[1]))


@pytest.mark.parametrize("B,H,W,drop", [(64, 64, 256, None), (32, 48, 200, (0.85, 99)), (16, 64, 128, None), (8, 40, 150, (0.9, 7))])
def test_backward_chain_equals_launch_per_step_backward(B, H, W, drop):
    """The backward chain alone: ONE forward (the forward chain), then BPTT twice over the same record -- by the persistent backward
    chain and by the launch-per-step kernels (lxo_shape.step_kernels = 2 for the second lxo_decoder_train_bwd).  Both apply the same
    expressions to the same operands; a step GEMM's contraction is split over 8 waves instead of 4, so f32 sums differ in their last
    bits and a bf16 mirror of g_t / d_z_t may round the other way here and there.  Per-step intermediates and every gradient."""
    img, f, l = batch(B, H, W, V, 5, 24, seed=177 + B)
    eng = Engine(V, dtype="bf16", seed=9)
    eng.forward(img, f, dropout=drop)
    eng.loss(l, 1.0 / int(l.sum()))
    T = f.shape[1]
    R = (-(-H // 8) - 2) * (-(-W // 8) - 2)
    Rp = (R + 7) // 8 * 8

    def snap():
        torch.cuda.synchronize()
        out = {"g": eng.region("g", "f32", (T, B, 512)), "dhc": eng.region("dhc", "f32", (T, B, 1024)),
               "de": eng.region("de", "f32", (T, B, Rp))[:, :, :R], "datth": eng.region("datth", "f32", (T, B, 256)),
               "dz": eng.region("dz", "f32", (T, B, 2048)), "dxh": eng.region("dxh", "f32", (B, 1024)), "dcc": eng.region("dcc", "f32", (B, 512))}
        return {k: v.cpu().numpy().copy() for k, v in out.items()}, eng.grad_dict()

    eng.backward()
    used, err = eng.chain_status(backward=True)
    assert used and err == 0, (used, err)
    a, ga = snap()
    eng.shape.step_kernels = 2                                   # the same record, the launch-per-step kernels
    eng.backward()
    b, gb = snap()
    worst = {}
    for k in a:
        assert np.isfinite(a[k]).all(), k
        d = np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-30)
        worst[k] = float(d)
        assert d < 2e-2, (k, d)
        assert cosine(a[k], b[k]) > 0.99999, (k, cosine(a[k], b[k]))
    wc = (1.0, None)
    for k in ga:
        c = cosine(ga[k], gb[k])
        if c < wc[0]:
            wc = (c, k)
        assert c > 0.99995, (k, c)
    print("B=%d %dx%d drop=%s: backward chain vs launch-per-step: max rel %s; worst gradient cosine %.7f (%s)" % (
        B, H, W, drop, {k: "%.1e" % v for k, v in worst.items()}, wc[0], wc[1]))


class _FaultyEngine(Engine):
    """Engine whose chains "break" on demand.  No MI355X here has ever broken one and the library carries no fault injector, so the test
    plays the hardware: it writes the chain's ERROR WORD (include/lxo.h: LXO_XDEC_ERR_WORD of the chain's block in ws region "xdec_sync")
    into device memory behind the launch, exactly where a timed-out barrier would have left it; everything that READS the word is the
    product's own code (Engine.chain_status, lxo_chain_guard, the CE kernel's NaN backstop)."""
    fault_fwd = fault_bwd = fault_guard = 0

    def _poke(self, backward, value):
        from latex_ocr_amd import _abi
        w = self.region("xdec_sync", "i32")
        w[(_abi.LXO_XDEC_BLOCK_BYTES // 4 if backward else 0) + _abi.LXO_XDEC_ERR_WORD] = value

    def chain_status(self, backward=False):
        if backward and self.fault_bwd:
            self.fault_bwd -= 1
            self._poke(True, 7)
        if not backward and self.fault_fwd:
            self.fault_fwd -= 1
            self._poke(False, 7)
        return Engine.chain_status(self, backward)

    def _chain_health_post(self, have_scale, dp=False):
        if self.fault_guard:                                     # a later step's chain breaks: seen by the device-side guard only
            self.fault_guard -= 1
            self._poke(True, 7)
        return Engine._chain_health_post(self, have_scale, dp)


def test_engine_falls_back_when_a_chain_reports_an_error():
    """The chains rely on how the hardware places a 256-workgroup grid; a chain that does not assemble sets an error word and the engine
    switches to the launch-per-step kernels for good and repeats the call: the repeated forward must be the launch chain's bit for bit
    (no atomics in the forward), the repeated backward its gradients."""
    img, f, l = batch(16, 48, 160, V, 5, 24, seed=31)
    n = int(l.sum())
    ref = Engine(V, dtype="bf16", seed=4)
    ref.step_kernels = 2
    ref.forward(img, f)
    sr = ref.loss(l, 1.0 / n).cpu().numpy().copy()
    T = f.shape[1]
    lr = ref.region("logits", "f32", (T, 16, (V + 31) // 32 * 32)).cpu().numpy().copy()
    ref.backward()
    torch.cuda.synchronize()
    gr = ref.grad_dict()

    eng = _FaultyEngine(V, dtype="bf16", seed=4)
    eng.fault_fwd = 1                                           # the first forward chain launch "reports" a broken chain
    with pytest.warns(RuntimeWarning, match="did not assemble"):
        eng.forward(img, f)
    assert eng.step_kernels == 2 and not eng.chain_used and eng.chain_failures == 1     # noticed, switched, repeated
    s = eng.loss(l, 1.0 / n).cpu().numpy().copy()
    assert np.array_equal(eng.region("logits", "f32", (T, 16, (V + 31) // 32 * 32)).cpu().numpy(), lr)
    assert s[1] == sr[1] and abs(s[0] - sr[0]) <= 1e-6 * abs(sr[0])
    eng.backward()
    torch.cuda.synchronize()
    for k, g in eng.grad_dict().items():
        assert cosine(g, gr[k]) > 0.999999, (k, cosine(g, gr[k]))

    eng2 = _FaultyEngine(V, dtype="bf16", seed=4)               # forward chain fine, backward chain reports the error
    eng2.forward(img, f)
    assert eng2.chain_used
    eng2.loss(l, 1.0 / n)
    eng2.fault_bwd = 1
    with pytest.warns(RuntimeWarning, match="backward decoder chain"):
        eng2.backward()
    torch.cuda.synchronize()
    assert eng2.step_kernels == 2 and not eng2.chain_used_bwd
    for k, g in eng2.grad_dict().items():
        assert np.isfinite(g).all() and cosine(g, gr[k]) > 0.9999, (k, cosine(g, gr[k]))
    eng2.forward(img, f)                                        # and the engine stays on the launch chain
    assert not eng2.chain_status()[0] or eng2.step_kernels == 2


def test_a_chain_that_breaks_in_a_later_step_drops_that_step_and_falls_back():
    """After the first use the chains are watched WITHOUT a host stall: lxo_chain_guard turns the optimizer's scale into NaN on the device when
    a chain of the step left an error word (lxo_adam_step then touches nothing: garbage gradients are never applied) and the words reach the
    host through a pinned ring that the next train_step looks at.  Non-syncing caller: the step is dropped (weights bit-identical to before it,
    Adam's time step taken back) and the engine runs the launch-per-step kernels from then on; syncing caller (it reads the loss): the step is
    repeated on those kernels, so loss and update are the ones asked for."""
    img, f, l = batch(16, 48, 160, V, 5, 24, seed=33)
    eng = _FaultyEngine(V, dtype="bf16", seed=6)
    eng.train_step(img, f, l, 1e-3, sync_loss=False)            # first use: checked synchronously, fine
    torch.cuda.synchronize()
    assert eng.chain_used and eng.chain_used_bwd and eng.adam_t == 1
    before = eng.params.clone()
    m_before = eng.adam_m.clone()
    eng.fault_guard = 1
    eng.train_step(img, f, l, 1e-3, sync_loss=False)            # the backward chain of this step "breaks"
    torch.cuda.synchronize()
    assert torch.equal(eng.params, before) and torch.equal(eng.adam_m, m_before)        # dropped on the device
    with pytest.warns(RuntimeWarning, match="did not assemble"):
        eng.train_step(img, f, l, 1e-3, sync_loss=False)        # the host learns of it here (no stall) and falls back
    torch.cuda.synchronize()
    assert eng.step_kernels == 2 and eng.chain_failures == 1 and getattr(eng, "dropped_steps", 0) == 1
    assert eng.adam_t == 2 and not torch.equal(eng.params, before)      # 3 calls, 2 applied updates
    assert torch.isfinite(eng.params).all()

    # syncing caller: the failed step is redone on the launch-per-step kernels
    ref = Engine(V, dtype="bf16", seed=6)
    ref.step_kernels = 2
    l0 = ref.train_step(img, f, l, 1e-3)
    l1 = ref.train_step(img, f, l, 1e-3)
    e2 = _FaultyEngine(V, dtype="bf16", seed=6)
    a0 = e2.train_step(img, f, l, 1e-3)
    e2.fault_guard = 1
    with pytest.warns(RuntimeWarning, match="did not assemble"):
        a1 = e2.train_step(img, f, l, 1e-3)
    assert e2.step_kernels == 2 and e2.adam_t == 2 and getattr(e2, "dropped_steps", 0) == 1
    assert abs(a0 - l0) <= 2e-4 * abs(l0) and abs(a1 - l1) <= 2e-3 * abs(l1), (a0, l0, a1, l1)
    assert cosine(e2.params.cpu().numpy(), ref.params.cpu().numpy()) > 0.999999


def test_chain_with_dropout_equals_launch_per_step():
    a, ga, used, err = _run(0, 16, 48, 160, dropout=(0.8, 1234))
    assert used and err == 0
    b, gb, _, _ = _run(2, 16, 48, 160, dropout=(0.8, 1234))
    la, lb = a["stats"][0] / a["stats"][1], b["stats"][0] / b["stats"][1]
    assert abs(la - lb) <= 5e-5 * abs(lb), (la, lb)
    # the same counter-based masks in both: dropped elements are exact zeros in the same places
    assert np.array_equal(a["rec"][1:, :, :512] == 0.0, b["rec"][1:, :, :512] == 0.0)
    for k in ga:
        assert cosine(ga[k], gb[k]) > 0.9999, (k, cosine(ga[k], gb[k]))


def test_chain_vs_oracle_loss_and_greedy_weights():
    """the chain against the CPU oracle directly (not only against the other kernel chain): loss within the bf16 bar at B = 16"""
    img, f, l = batch(16, 64, 256, V, 5, 24, seed=3)
    eng = Engine(V, dtype="bf16", seed=3)
    P = oracle_params(eng)
    eng.forward(img, f)
    n = int(l.sum())
    stats = eng.loss(l, 1.0 / n).cpu().numpy()
    used, err = eng.chain_status()
    assert used and err == 0
    loss_ref, G, ce, nw = R.train_grads(P, torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l))
    assert abs(stats[0] / stats[1] - float(loss_ref)) / float(loss_ref) < 1e-3
    eng.backward()
    torch.cuda.synchronize()
    got = eng.grad_dict()
    for k in G:
        assert cosine(got[k], G[k].numpy()) > 0.9999, (k, cosine(got[k], G[k].numpy()))      # measured >= 0.99995 (B = 64: test_gpu_benchcfg)


@pytest.mark.parametrize("step_kernels", [0, 2])
def test_bf16_cell_transcendentals_over_pm20(step_kernels):
    """The bf16 cell epilogues form tanh / sigmoid from v_exp_f32 + v_rcp_f32 (rstep.hip since round 3, xdec.hip since round 4) instead of
    the library calls the f32 parity mode keeps.  Bracket them over pre-activations in [-20, 20]: with the LSTM kernel zeroed the
    pre-activation of unit u is exactly lstm_cell/bias[u] (+ forget_bias 1 on the f gate), so the stored gates of step 0 ARE
    sigmoid / tanh of a known ramp; c_1 and h_1 follow from c_0 = tanh(b_c_0).  Held against float64 to 4e-7 absolute on the gates
    (measured 1.2e-7; the saturated ends must be exact 0 / 1 / +-1), through the launch-per-step kernels (2) and the persistent chain (0)."""
    Vv = 50
    img, f, l = batch(8, 32, 128, Vv, 3, 6, seed=21)
    eng = Engine(Vv, dtype="bf16", seed=1)
    eng.step_kernels = step_kernels
    P = eng.get_params()
    U = 512
    ramp = np.linspace(-20.0, 20.0, 4 * U).astype(np.float32)
    rng = np.random.default_rng(0)
    rng.shuffle(ramp)                                            # every gate sees the whole range
    P["Decoder/AttentionCell/rnn/lstm_cell/kernel"][:] = 0.0
    P["Decoder/AttentionCell/rnn/lstm_cell/bias"][:] = ramp
    bc0 = np.linspace(-3.0, 3.0, U).astype(np.float32)
    P["Decoder/AttentionCell/att_mechanism/W_c_0"][:] = 0.0
    P["Decoder/AttentionCell/att_mechanism/b_c_0"][:] = bc0
    eng.load_params(P)
    eng.forward(img, f)
    torch.cuda.synchronize()
    used, err = eng.chain_status()
    assert err == 0 and used == (step_kernels == 0)
    T = f.shape[1]
    g = eng.region("gates", "f32", (T, 8, 4 * U))[0, 0].cpu().numpy().astype(np.float64)
    z = ramp.astype(np.float64)
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    want = np.concatenate([sig(z[:U]), np.tanh(z[U:2 * U]), sig(z[2 * U:3 * U] + 1.0), sig(z[3 * U:])])
    d = np.abs(g - want)
    print("step_kernels=%d: gates vs float64 over [-20, 20]: max abs error %.2e" % (step_kernels, d.max()))
    assert d.max() < 4e-7, d.max()
    c0 = np.tanh(bc0.astype(np.float64))
    c1 = want[2 * U:3 * U] * c0 + want[:U] * want[U:2 * U]
    h1 = want[3 * U:] * np.tanh(c1)
    cs = eng.region("cs", "f32", (T + 1, 8, U))[1, 0].cpu().numpy().astype(np.float64)
    h = eng.region("rec", "f32", (T + 1, 8, 2048))[1, 0, 512:1024].cpu().numpy().astype(np.float64)
    assert np.abs(cs - c1).max() < 2e-6 and np.abs(h - h1).max() < 2e-6, (np.abs(cs - c1).max(), np.abs(h - h1).max())


def test_weight_gradients_on_the_second_stream_agree_with_one_stream(monkeypatch):
    """Round 5: the encoder's weight gradients and the decoder's deferred ones (dense dW GEMMs, d_z's column sum, the embedding gradient,
    the init-state gradients, dW_att_img) run on a second stream beside the d_img path (Engine.enc_side, default on).  Same kernels, same
    operands: against the one-stream order (LXO_ENC_OVERLAP=0) only the order of f32 atomics differs.  B = 16: behind the backward chain."""
    img, f, l = batch(16, 64, 256, V, 5, 24, seed=91)
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("LXO_ENC_OVERLAP", flag)
        eng = Engine(V, dtype="bf16", seed=5)
        assert (eng.enc_side is not None) == (flag == "1")
        for _ in range(2):                                   # twice: the second pass reuses the events and the bound stream
            eng.forward(img, f)
            st = eng.loss(l, 1.0 / int(l.sum())).cpu().numpy().copy()
            eng.backward()
        torch.cuda.synchronize()
        assert eng.chain_status(backward=True) == (True, 0)
        out.append((st, eng.grad_dict()))
    (sa, ga), (sb, gb) = out
    assert sa[1] == sb[1] and abs(sa[0] - sb[0]) <= 1e-5 * abs(sb[0])
    for k in ga:
        c = cosine(ga[k], gb[k])
        r = float(np.abs(ga[k] - gb[k]).max() / max(np.abs(gb[k]).max(), 1e-30))
        assert c > 0.99999 and r < 1e-2, (k, c, r)


@pytest.mark.parametrize("dtype,det", [("f32", False), ("bf16", True), ("bf16", False)])
def test_train_bwd_in_one_call_equals_the_two_calls(monkeypatch, dtype, det):
    """lxo_train_bwd (decoder + encoder backward, one join of the weight-gradient stream) against lxo_decoder_train_bwd + lxo_encoder_bwd:
    the same kernels on the same operands.  In the reproducible modes (f32; bf16 deterministic) every gradient is bit-identical; in the default
    bf16 mode only the order of f32 atomics may differ.  Engine.backward takes the one-call form from the second backward of a shape on."""
    img, f, l = batch(16, 48, 160, V, 5, 24, seed=33)
    out = []
    for fused in ("1", "0"):
        monkeypatch.setenv("LXO_TRAIN_BWD_FUSED", fused)
        eng = Engine(V, dtype=dtype, seed=6, deterministic=det)
        for _ in range(2):                                   # the first backward of a shape is always the two calls (the chain is looked at)
            eng.forward(img, f)
            eng.loss(l, 1.0 / int(l.sum()))
            eng.backward()
        torch.cuda.synchronize()
        out.append(eng.grad_dict())
    ga, gb = out
    for k in ga:
        if dtype == "f32" or det:
            assert ga[k].tobytes() == gb[k].tobytes(), k
        else:
            assert cosine(ga[k], gb[k]) > 0.99999, (k, cosine(ga[k], gb[k]))


def test_ready_events_fire_when_a_buckets_gradients_are_final():
    """lxo_train_bwd with an event table (what the data-parallel exchange waits for): a third stream that waits for a bucket's event and copies
    the bucket at once must see the FINAL gradients -- the event is recorded behind both streams' work on that range (the weight gradients run
    on the side stream, bias gradients and the chain-failure probe on the compute stream)."""
    import ctypes
    from latex_ocr_amd.engine import _p
    img, f, l = batch(16, 64, 256, V, 5, 24, seed=17)
    eng = Engine(V, dtype="bf16", seed=8)
    assert eng.enc_side is not None
    for _ in range(2):
        eng.forward(img, f)
        eng.loss(l, 1.0 / int(l.sum()))
        eng.backward()
    torch.cuda.synchronize()
    ranges = {0: (eng.buckets[1][0], eng.buckets[0][1])}               # every decoder parameter: [first_dec, n_params)
    for (hi, lo), rng in eng.enc_buckets:
        ranges[lo] = rng
    for rep in range(3):
        eng.forward(img, f)
        eng.loss(l, 1.0 / int(l.sum()))
        evs = eng._enc_ready_events()
        table = (ctypes.c_void_p * 7)(*[ctypes.c_void_p(e.cuda_event) if e is not None else None for e in evs])
        eng._bind_side()
        eng.grads.zero_()
        eng._ck(eng.lib.lxo_train_bwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), _p(eng._img), _p(eng.grads), table,
                                      eng._stream()), "train_bwd")
        third = torch.cuda.Stream()
        snaps = {}
        for k in (0, 6, 5, 4, 3, 1):                                   # the order the events are recorded in
            third.wait_event(evs[k])
            with torch.cuda.stream(third):
                snaps[k] = eng.grads[ranges[k][0]:ranges[k][1]].clone()
        torch.cuda.synchronize()
        for k, (lo, hi) in ranges.items():
            final = eng.grads[lo:hi]
            assert torch.isfinite(final).all() and float(final.abs().max()) > 0, k
            assert torch.equal(snaps[k], final), (rep, k, float((snaps[k] - final).abs().max()))


def test_a_foreign_kernel_holding_a_cu_breaks_the_chain_and_the_guard_drops_exactly_that_step():
    """No injected error word here: a FOREIGN kernel (torch's one-wave spin kernel on a third stream) holds one CU for ~0.7 s while a
    training step launches its chains.  A chain workgroup needs a whole CU (8 waves x 255 VGPRs, 158 KB LDS), so one XCD gets 31 of its 32
    workgroups until the foreigner leaves: the chain's bounded waits (200 ms) give up, leave the error word, the loss kernel's backstop and
    lxo_chain_guard drop the step ON THE DEVICE (weights and Adam slots bit-identical to before it), nothing hangs; the next train_step
    learns of it without a stall, takes the dropped update's Adam time step back and continues on the launch-per-step kernels -- from there
    on the run IS the fallback's run: same weights as an engine that ran the launch-per-step kernels throughout and never saw that batch.
    (The situation at N > 1: a collective kernel of a lagging peer resident when a chain launches.)"""
    b1 = batch(16, 48, 160, V, 5, 24, seed=41)
    b2 = batch(16, 48, 160, V, 5, 24, seed=42)
    b3 = batch(16, 48, 160, V, 5, 24, seed=43)
    eng = Engine(V, dtype="bf16", seed=8, deterministic=True)
    ref = Engine(V, dtype="bf16", seed=8, deterministic=True)
    ref.step_kernels = 2
    eng.train_step(*b1, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    assert eng.chain_used and eng.chain_used_bwd and eng.adam_t == 1
    before, m_before = eng.params.clone(), eng.adam_m.clone()
    # how long one spin cycle of torch.cuda._sleep is on this box
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
    per_ms = 20_000_000 / e0.elapsed_time(e1)
    foreign = torch.cuda.Stream()
    with torch.cuda.stream(foreign):
        torch.cuda._sleep(int(700 * per_ms))                     # ~0.7 s on one CU: longer than the forward chain's 200 ms patience
    t0 = time.perf_counter()
    eng.train_step(*b2, 1e-3, sync_loss=False)                   # its chains cannot assemble
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    assert waited < 5.0, waited                                  # bounded: the chains gave up, nothing hung
    assert torch.equal(eng.params, before) and torch.equal(eng.adam_m, m_before)      # the step was dropped on the device
    with pytest.warns(RuntimeWarning, match="did not assemble"):
        eng.train_step(*b3, 1e-3, sync_loss=False)               # the host learns of it here, without a stall
    torch.cuda.synchronize()
    eng._chain_health_poll(wait=True)
    assert eng.step_kernels == 2 and eng.chain_failures == 1 and getattr(eng, "dropped_steps", 0) == 1 and eng.adam_t == 2
    # the fallback's run: launch-per-step kernels throughout, batches 1 and 3
    ref.train_step(*b1, 1e-3, sync_loss=False)
    ref.train_step(*b3, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    c = cosine(eng.params.cpu().numpy(), ref.params.cpu().numpy())
    d = float((eng.params - ref.params).abs().max())
    print("foreign kernel: the broken step took %.2f s; weights vs the fallback's run: cosine %.9f, max |diff| %.2e" % (waited, c, d))
    assert c > 0.999999 and d <= 5e-3                            # step 1 ran on the chains here, on the launch-per-step kernels there (bf16 K-split noise through one Adam step of lr 1e-3)
    la = eng.train_step(*b1, 1e-3)                               # and both continue alike
    lb = ref.train_step(*b1, 1e-3)
    assert abs(la - lb) <= 2e-3 * abs(lb), (la, lb)
