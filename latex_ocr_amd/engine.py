"""Device-side engine: owns the flat parameter / gradient / Adam buffers, the packed
compute-dtype weights and the workspace (all torch CUDA tensors -- PyTorch is the
allocator and stream provider only) and drives the C ABI of liblxo.so.

Every numeric operation happens inside liblxo.so (hand-written gfx950 kernels);
there is no eager/PyTorch compute path and no CPU fallback here.
"""
import ctypes
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _abi
from .model import params as PP


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# include/lxo.h: LXO_XDEC_BLOCK_BYTES (csrc/xdec.h: kXDecBlockBytes) -- ws region "xdec_sync" holds one such block per chain (forward, then
# backward): 4 KB of flag lines, tickets and the error word (word LXO_XDEC_ERR_WORD) + 384 KB of hand-over words (tests/test_abi.py keeps
# the definitions together)
XDEC_BLOCK_BYTES = _abi.LXO_XDEC_BLOCK_BYTES


class Engine(object):
    def __init__(self, n_tok, dims=None, dtype="bf16", device="cuda:0", seed=0, beam=1, max_steps=0, lib=None, deterministic=None):
        self.lib = lib if lib is not None else _abi.load()
        self.device = torch.device(device)
        if self.device.type != "cuda" and lib is None:
            raise RuntimeError("latex_ocr_amd.Engine needs a CUDA/HIP device (no CPU fallback)")
        self.dims = dict(PP.DEFAULT_DIMS, **(dims or {}))
        self.n_tok = int(n_tok)
        self.dtype = _abi.LXO_BF16 if dtype in ("bf16", 1) else _abi.LXO_F32
        self.beam, self.max_steps = int(beam), int(max_steps)
        # decoder step decomposition (lxo_shape.step_kernels): 0 = automatic (the persistent XCD-local chain of csrc/xdec.hip where the shape
        # qualifies, else the fused step kernels), 2 = always the fused step kernels, 1 = round 1's split-K slab path
        self.step_kernels = int(os.environ.get("LXO_STEP_KERNELS", "0"))
        if self.step_kernels not in (0, 1, 2):
            raise ValueError("LXO_STEP_KERNELS must be 0, 1 or 2 (include/lxo.h: lxo_shape.step_kernels), got %d" % self.step_kernels)
        # bf16 mode only: every reduction of a training step in a fixed order (lxo_shape.deterministic): bit-identical losses, gradients and
        # weights from run to run, for a few per cent of a step.  The f32 parity mode is always deterministic.
        self.deterministic = (os.environ.get("LXO_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        # health of the persistent decoder chains (csrc/xdec.hip): checked synchronously after the first launch that used one (forward and
        # backward separately), then every step WITHOUT a host stall: lxo_chain_guard folds the two error words into the optimizer's scale on
        # the device (a step whose chain did not assemble is dropped, not applied) and copies them into a pinned ring that the next
        # train_step reads (_chain_health_poll); a failure switches this engine to the launch-per-step kernels for good.
        self._xdec_checked = False
        self._xdec_bwd_checked = False
        self.chain_used = False
        self.chain_used_bwd = False
        self.chain_failures = 0           # steps a chain failed in (each was dropped or redone)
        self._nochain_shapes, self._nochain_shapes_bwd = set(), set()
        self._health_ring = None
        self._health_i = 0
        self.specs = PP.param_specs(self.n_tok, self.dims)
        self.n_params = PP.n_params(self.n_tok, self.dims)
        probe = self._shape(1, 32, 32, 1)
        assert self.lib.lxo_param_total(ctypes.byref(probe)) == self.n_params
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.n_params, **f32)
        self.grads = torch.zeros(self.n_params, **f32)
        self.adam_m = torch.zeros(self.n_params, **f32)
        self.adam_v = torch.zeros(self.n_params, **f32)
        self.adam_t = 0
        self.scale = torch.zeros(_abi.LXO_GNORM_FLOATS, **f32)      # {scale, norm, per-workgroup partial sums} of lxo_global_norm_scale
        self.wpack = torch.zeros(self.lib.lxo_wpack_bytes(ctypes.byref(probe)) + 256, dtype=torch.uint8, device=self.device)
        self.ws = None
        self.ws_key = None
        self.shape = None
        self._offsets = OrderedDict()
        off = 0
        for name, shp, _ in self.specs:
            n = int(np.prod(shp))
            self._offsets[name] = (off, n, shp)
            off += n
        # gradient buckets for data-parallel all-reduce, in the order backward finishes them
        first_dec = self._offsets["Decoder/embedding_table"][0]
        for i in range(self.lib.lxo_param_num()):          # the C plan and this inventory must agree slot by slot
            o, c = ctypes.c_longlong(), ctypes.c_longlong()
            self._ck(self.lib.lxo_param_info(ctypes.byref(probe), i, ctypes.byref(o), ctypes.byref(c)), "param_info")
            if c.value:
                name = self.lib.lxo_param_name_for(ctypes.byref(probe), i).decode()
                assert self._offsets[name][:2] == (o.value, c.value), (name, self._offsets[name], o.value, c.value)
        c3 = self._offsets["Encoder/convolutional_encoder/conv2d_2/kernel"][0]
        c4 = self._offsets["Encoder/convolutional_encoder/conv2d_3/kernel"][0]
        c5 = self._offsets["Encoder/convolutional_encoder/conv2d_4/kernel"][0]       # (the cnn variant's strided conv belongs to layer 5's pass)
        enc_names = [k for k in self._offsets if k.startswith("Encoder/convolutional_encoder/") and k.endswith("/kernel")]
        c6 = self._offsets[enc_names[-1]][0]                                          # the VALID conv in front of the decoder
        ywo = self._offsets["Decoder/AttentionCell/rnn/y_W_o"][0]        # the last variable: final before the recurrence runs
        # (range, encoder layers whose backward makes it final): a layer's kernel is final after its own pass, its bias no later
        self.buckets = [(ywo, self.n_params), (first_dec, ywo)]
        self.enc_buckets = [((6, 6), (c6, first_dec)), ((5, 5), (c5, c6)), ((4, 4), (c4, c5)), ((3, 3), (c3, c4)), ((2, 1), (0, c3))]
        # optional second stream for the half-batch interleave of the recurrent loop (LXO_DUAL_STREAM=1).
        # Measured slower than one stream in round 1 (18.8 vs 17.3 ms/step).  Round 2 measured why (tools/loop_probe.py,
        # DESIGN.md "launch cost"): the loop is bound by the device-side dependent-launch boundary, not by the host, and
        # two half-batch chains double the boundaries while each launch keeps its fixed cost; the interleave also forces
        # the round-1 split-K step kernels.  Off by default, kept as an A/B switch.
        self.side_stream = None
        if self.device.type == "cuda" and os.environ.get("LXO_DUAL_STREAM", "0") == "1":
            self.side_stream = torch.cuda.Stream(self.device)
        # second stream for the encoder's weight-gradient kernels (LXO_ENC_OVERLAP=0 switches it off): their prologues / epilogues and
        # the memory-bound pool-backward kernels overlap the data-gradient kernels.  Slower in rounds 1-3 (the cross-stream waits delayed
        # the launch-per-step decoder); with the decoder in two persistent launches it pays: 7.99 -> 7.87 ms per step (round 5).  The
        # library ignores it in the f32 parity mode and while bench.py records per-launch times (model_encoder.hip).
        self.enc_side = None
        if self.device.type == "cuda" and os.environ.get("LXO_ENC_OVERLAP", "1") != "0":
            self.enc_side = torch.cuda.Stream(self.device)
        self.load_params(PP.init_params(self.n_tok, seed, self.dims))

    # ------------------------------------------------------------ plumbing --
    def _shape(self, B, H, W, T):
        d = self.dims
        sh = _abi.LxoShape(B, H, W, T, self.n_tok, d["C"], d["E"], d["U"], d["O"], d["D"], self.dtype,
                           self.beam, self.max_steps)
        sh.encoder_cnn = 1 if d.get("cnn") else 0                 # configs/model.json encoder_cnn (encoder.py:46-56)
        sh.no_positional = 0 if d.get("positional", True) else 1  # positional_embeddings (encoder.py:60-65)
        sh.step_kernels = self.step_kernels
        sh.encoder_rnn = 1 if d.get("row_bilstm") else 0        # optional row-BiLSTM encoder (not in the reference; off by default)
        sh.deterministic = 1 if self.deterministic else 0
        return sh

    def _stream(self):
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return ctypes.c_void_p(0)

    def _ck(self, rc, what):
        _abi.check(self.lib, rc, what)

    def ensure(self, B, H, W, T):
        """Bind a call shape; grow the workspace when a larger batch arrives."""
        shape = self._shape(B, H, W, T)
        need = self.lib.lxo_workspace_bytes(ctypes.byref(shape)) + 256
        if self.ws is None or self.ws.numel() < need:
            self.ws = None
            self.ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        self.shape = shape
        return shape

    def sref(self):
        return ctypes.byref(self.shape)

    def region(self, name, kind="f32", shape=None):
        """View of a named workspace region as a torch tensor (float32, int32 or compute dtype)."""
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        self._ck(self.lib.lxo_ws_region(self.sref(), name.encode(), ctypes.byref(off), ctypes.byref(nb)), "ws_region")
        raw = self.ws[off.value:off.value + nb.value]
        if kind == "ct":
            t = raw.view(torch.bfloat16) if self.dtype == _abi.LXO_BF16 else raw.view(torch.float32)
        elif kind == "i32":
            t = raw.view(torch.int32)
        else:
            t = raw.view(torch.float32)
        if shape is not None:
            t = t[:int(np.prod(shape))].view(*shape)
        return t

    # ------------------------------------------------------------- params --
    def load_params(self, P):
        flat = np.concatenate([np.asarray(P[k], dtype=np.float32).reshape(-1) for k, _, _ in self.specs])
        self.params.copy_(torch.from_numpy(flat))
        self.pack()

    def get_params(self):
        flat = self.params.detach().cpu().numpy()
        return OrderedDict((k, flat[o:o + n].reshape(s).copy()) for k, (o, n, s) in self._offsets.items())

    def grad_dict(self):
        flat = self.grads.detach().cpu().numpy()
        return OrderedDict((k, flat[o:o + n].reshape(s).copy()) for k, (o, n, s) in self._offsets.items())

    def pack(self):
        shape = self.shape if self.shape is not None else self._shape(1, 32, 32, 1)
        self._ck(self.lib.lxo_pack_weights(ctypes.byref(shape), _p(self.params), _p(self.wpack), self._stream()), "pack_weights")

    def state_dict(self):
        return {"params": self.get_params(), "adam_m": self.adam_m.cpu().numpy(), "adam_v": self.adam_v.cpu().numpy(),
                "adam_t": self.adam_t}

    def load_state_dict(self, sd):
        self.load_params(sd["params"])
        if "adam_m" in sd:
            self.adam_m.copy_(torch.from_numpy(np.asarray(sd["adam_m"], np.float32)))
            self.adam_v.copy_(torch.from_numpy(np.asarray(sd["adam_v"], np.float32)))
            self.adam_t = int(sd.get("adam_t", 0))

    # -------------------------------------------------------------- inputs --
    def _to_dev(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=self.device, dtype=dtype)

    def _lengths_dev(self, lengths):
        """Formula lengths on the device WITHOUT stalling the host: a pageable host array goes through a small ring of pinned staging
        buffers and an asynchronous copy.  (A plain `.to(device)` of pageable memory blocks the host until the compute stream has
        drained -- in the middle of a step that is the whole decoder forward: the kernels behind it were then enqueued ~30 us late.)"""
        if isinstance(lengths, torch.Tensor) or self.device.type != "cuda":
            return self._to_dev(lengths, torch.int32)
        a = np.ascontiguousarray(lengths, dtype=np.int32).reshape(-1)
        n = int(a.shape[0])
        ring = getattr(self, "_len_ring", None)
        if not ring or ring[0]["host"].numel() < n:
            cap = max(64, n)
            ring = [{"host": torch.empty(cap, dtype=torch.int32).pin_memory(), "dev": torch.empty(cap, dtype=torch.int32, device=self.device), "ev": None}
                    for _ in range(4)]
            self._len_ring, self._len_i = ring, 0
        slot = ring[self._len_i % len(ring)]
        self._len_i += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()                             # the copy that last read this staging buffer (four steps ago) is long done
        slot["host"][:n].numpy()[:] = a
        slot["dev"][:n].copy_(slot["host"][:n], non_blocking=True)
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record(torch.cuda.current_stream(self.device))
        return slot["dev"][:n]

    # ---------------------------------------------------------------- steps --
    def forward(self, img, formula, dropout=None, phase_hook=None, before_decoder=None):
        """Encoder + teacher-forced decoder; leaves logits in the workspace.  dropout = (keep_prob, seed)
        applies tf.nn.dropout on h and o (attention_cell.py:72,83) with this step's counter-based masks;
        backward() regenerates the same masks from the bound shape."""
        B, H, W = int(img.shape[0]), int(img.shape[1]), int(img.shape[2])
        T = int(formula.shape[1])
        # a batch the persistent chains do not take (the reference trains at 3, buckets and evaluates at 20: configs/training.json:6,
        # data_generator.py:41, evaluate_txt.py:42) is filled up to the next chain batch with DEAD rows: copies of its own samples whose
        # formula length is 0 (loss() appends the zeros), so no token of theirs is inside the loss mask (img2seq.py:68-71), their d(logits)
        # and with it every gradient contribution is an exact 0, and n_words does not see them
        self.live_B = B
        drop_on = dropout is not None and 0.0 < float(dropout[0]) < 1.0
        # (with config.dropout < 1 the batch stays as it is: the counter-based masks are keyed by (step, row, batch size), and the oracle the
        # dropout tests compare with draws them for the caller's batch -- the reference ships keep-probability 1, configs/training.json:7)
        Bp = B if drop_on else self._train_chain_batch(B, H, W)
        if Bp != B:
            # (lxo_shape.live_B: the encoder computes the B live images only -- `img` stays as it is -- and leaves zero features for the dead rows)
            # (host rows are repeated on the host, before the upload; device rows by ONE index_select with a cached index: at batch 3 the arange / remainder /
            # index_select triple was 3 launches with their gaps -- ~30 us of a 1.25 ms step)
            if isinstance(formula, torch.Tensor):
                cache = self.__dict__.setdefault("_pad_rows", {})
                idx = cache.get((Bp, B, formula.device))
                if idx is None:
                    idx = cache[(Bp, B, formula.device)] = (torch.arange(Bp, device=formula.device) % B)
                formula = formula.index_select(0, idx)
            else:
                formula = np.ascontiguousarray(np.asarray(formula)[np.arange(Bp) % B])
        self.ensure(Bp, H, W, T)
        self.shape.live_B = B if Bp != B else 0
        B = Bp
        if dropout is not None and 0.0 < float(dropout[0]) < 1.0:
            self.shape.keep_prob = float(dropout[0])
            self.shape.dropout_seed = int(dropout[1]) & 0x7FFFFFFF
        self._img = self._to_dev(img, torch.uint8)
        self._formula = self._to_dev(formula, torch.int32)
        st = self._stream()
        self._bind_side()
        self._ck(self.lib.lxo_encoder_fwd(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._img), st), "encoder_fwd")
        if phase_hook:
            phase_hook("encoder_fwd")
        if before_decoder is not None:
            # data parallel: the token-count all-reduce (its own stream) must be OFF the GPU before the persistent decoder chain starts --
            # the chain wants every CU, and a collective kernel that waits for a late rank would hold some of them
            torch.cuda.current_stream(self.device).wait_event(before_decoder)
        self._ck(self.lib.lxo_decoder_train_fwd(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._formula), st),
                 "decoder_train_fwd")
        if (not self._xdec_checked and self.dtype == _abi.LXO_BF16 and self.step_kernels == 0 and self.device.type == "cuda"
                and (B, H, W) not in self._nochain_shapes):
            self._check_chain(st)

    def chain_status(self, backward=False):
        """(used, error) of the persistent XCD-local decoder chain (csrc/xdec.hip) in the last lxo_decoder_train_fwd (backward=True: the
        backward chain of the last lxo_decoder_train_bwd): used = its 8 x 32 workgroups took their tickets; error != 0 = a chain did not
        assemble (a barrier timed out / an XCD got the wrong number of workgroups) and the step's decoder outputs are invalid.
        Synchronises the device."""
        o = XDEC_BLOCK_BYTES // 4 if backward else 0             # the backward chain's block
        w = self.region("xdec_sync", "i32")[o:o + 8 * 64 + 1].cpu().numpy()
        return bool(w[32:512:64].any()), int(w[512])

    def _check_chain(self, st):
        """After the first forward that RAN the persistent chain (a shape that does not qualify leaves no tickets: checked again on the
        next one): the chain relies on how the hardware places a 256-workgroup grid (32 per XCD, one per CU).  If it reports an error,
        switch this engine to the launch-per-step chain for good and redo the decoder forward.  Later steps are watched without a host
        synchronisation (_chain_health_post / _chain_health_poll)."""
        used, err = self.chain_status()
        self.chain_used = used and not err
        if used or err:
            self._xdec_checked = True
        else:
            self._nochain_shapes.add((self.shape.B, self.shape.H, self.shape.W))      # this shape takes the launch-per-step kernels: no need to look again
        if err:
            self._chain_fallback("forward", err)
            self._ck(self.lib.lxo_decoder_train_fwd(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._formula), st),
                     "decoder_train_fwd")

    def _chain_fallback(self, which, err):
        import warnings
        self.chain_failures += 1
        self.step_kernels = 2
        if self.shape is not None:
            self.shape.step_kernels = 2
        self.chain_used = self.chain_used_bwd = False
        warnings.warn("latex_ocr_amd: the persistent %s decoder chain did not assemble (error word %d: a barrier / hand-over timed out or an "
                      "XCD got the wrong number of workgroups -- CUs held by another kernel or process, a CU mask); this engine now runs the "
                      "launch-per-step decoder kernels (lxo_shape.step_kernels = 2)" % (which, err), RuntimeWarning)

    def _chains_possible(self):
        return self.dtype == _abi.LXO_BF16 and self.step_kernels == 0 and self.device.type == "cuda"

    def _chain_health_post(self, have_scale, dp=False):
        """Behind backward(), before the optimizer: lxo_chain_guard (device: scale[0] = NaN when a chain of this step failed, so the optimizer
        drops the step) + an asynchronous copy of the two error words into a pinned ring slot.  No host synchronisation.
        The NaN probe element of the gradients (a PEER's chain failed) is only looked at in a data-parallel step: a single process whose
        gradients are NaN for numeric reasons is not protected from itself -- the reference would write the NaNs into its weights too
        (img2seq.py:119-123), and a silently dropped update would hide the divergence."""
        st = self._stream()
        if self._health_ring is None:
            self._health_dev = torch.zeros(4, dtype=torch.int32, device=self.device)
            self._health_ring = [{"host": torch.zeros(4, dtype=torch.int32).pin_memory(), "ev": None} for _ in range(4)]
        self._ck(self.lib.lxo_chain_guard(self.sref(), _p(self.ws), _p(self.grads if dp else None), _p(self.scale), 1 if have_scale else 0, _p(self._health_dev), st),
                 "chain_guard")
        slot = self._health_ring[self._health_i % len(self._health_ring)]
        self._health_i += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()                 # four steps old: long done (and its words were looked at by _chain_health_poll)
            self._chain_health_look(slot)
        slot["host"].copy_(self._health_dev, non_blocking=True)
        slot["seq"] = self._health_i
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record(torch.cuda.current_stream(self.device))

    def _chain_health_look(self, slot):
        ef, eb, dropped = int(slot["host"][0]), int(slot["host"][1]), int(slot["host"][2])
        slot["ev"] = None
        if dropped:
            self.dropped_steps = getattr(self, "dropped_steps", 0) + 1
            self._last_dropped_seq = slot["seq"]
            if getattr(self, "method", 0) == 0 and self.adam_t > 0:
                self.adam_t -= 1                      # the step was dropped on the device (NaN scale) on EVERY rank: its Adam time step did not happen
        if (ef or eb) and self.step_kernels == 0:
            self._chain_fallback("forward" if ef else "backward", ef or eb)
        return bool(dropped)

    def _chain_health_poll(self, wait=False):
        """Look at the error words of finished steps (wait=True: of every posted step; the caller has synchronised or accepts to).
        -> True when a failure was found (the engine has switched to the launch-per-step kernels; the failed step was dropped)."""
        bad = False
        for slot in self._health_ring or ():
            if slot["ev"] is not None and (wait or slot["ev"].query()):
                if wait:
                    slot["ev"].synchronize()
                bad = self._chain_health_look(slot) or bad
        return bad

    def _bind_side(self):
        side = ctypes.c_void_p(self.side_stream.cuda_stream) if self.side_stream is not None else ctypes.c_void_p(0)
        self._ck(self.lib.lxo_set_side_stream(side), "set_side_stream")
        es = getattr(self, "enc_side", None)
        self._ck(self.lib.lxo_set_encoder_side_stream(ctypes.c_void_p(es.cuda_stream) if es is not None else ctypes.c_void_p(0)),
                 "set_encoder_side_stream")

    def loss(self, lengths, inv_ntok=None, ntok_dev=None, ntok_event=None):
        """Loss statistics + d(logits).  Either inv_ntok (host float) or ntok_dev (device float32 [1] = the global token count,
        e.g. from DataParallel.sum_count_async; the compute stream waits for ntok_event, the host does not)."""
        dead = int(self.shape.B) - int(getattr(self, "live_B", self.shape.B))
        if dead > 0:                                               # the dead rows forward() appended: length 0 = outside the loss mask
            if isinstance(lengths, torch.Tensor):
                zc = self.__dict__.setdefault("_pad_zeros", {})
                z = zc.get((dead, lengths.device))
                if z is None:
                    z = zc[(dead, lengths.device)] = torch.zeros(dead, dtype=torch.int32, device=lengths.device)
                lengths = torch.cat([lengths.to(torch.int32), z])
            else:
                lengths = np.concatenate([np.asarray(lengths, dtype=np.int32).reshape(-1), np.zeros(dead, np.int32)])
        self._lengths = self._lengths_dev(lengths)
        if ntok_dev is not None:
            if ntok_event is not None:
                torch.cuda.current_stream(self.device).wait_event(ntok_event)
            self._ntok_dev = ntok_dev          # keep alive until the kernel has been enqueued
            if ntok_dev.is_cuda:               # allocated on the count stream, read on this one: the allocator must not hand the
                ntok_dev.record_stream(torch.cuda.current_stream(self.device))   # block to a later upload before this kernel ran
            self._ck(self.lib.lxo_ce_loss_fwd_bwd_dev(self.sref(), _p(self.ws), _p(self._formula), _p(self._lengths), _p(ntok_dev),
                                                      self._stream()), "ce_loss_dev")
        else:
            self._ck(self.lib.lxo_ce_loss_fwd_bwd(self.sref(), _p(self.ws), _p(self._formula), _p(self._lengths),
                                                  ctypes.c_float(inv_ntok), self._stream()), "ce_loss")
        return self.region("loss")[:2]

    def backward(self, comm=None, phase_hook=None):
        """BPTT + encoder backward into self.grads (zeroed first).  `comm(lo, hi)` is called as soon
        as the gradient range [lo, hi) is final (data-parallel bucket all-reduce hook); phase_hook(name) after each
        part (bench.py's per-phase table)."""
        st = self._stream()
        self._bind_side()
        # the recurrence may run as the persistent backward chain (csrc/xdec.hip): it wants every CU, so no collective kernel is put
        # beside it -- y_W_o's all-reduce then follows the recurrence instead of overlapping it
        chain = (self.dtype == _abi.LXO_BF16 and self.shape.step_kernels == 0 and self.device.type == "cuda")

        def decoder_bwd(defer_b0):
            self.grads.zero_()
            if comm:
                # y_W_o's gradient needs only d(logits): reduce it while the recurrence runs
                self._ck(self.lib.lxo_decoder_train_bwd_part(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._formula),
                                                             _p(self.grads), 1, st), "decoder_train_bwd_part")
                if not defer_b0:
                    comm(*self.buckets[0])
                self._ck(self.lib.lxo_decoder_train_bwd_part(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._formula),
                                                             _p(self.grads), 2, st), "decoder_train_bwd_part")
            else:
                self._ck(self.lib.lxo_decoder_train_bwd(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._formula),
                                                        _p(self.grads), st), "decoder_train_bwd")

        # After the first backward of this shape has been looked at (below), the whole pass is ONE call: lxo_train_bwd joins the weight-gradient
        # side stream once, at the end (the decoder's deferred weight gradients then also run beside conv6's data gradient), and records
        # an event per gradient bucket for the data-parallel exchange.  bench.py's instrumented step (phase_hook) keeps the two calls.
        key = (self.shape.B, self.shape.H, self.shape.W)
        looked = self._xdec_bwd_checked or key in self._nochain_shapes_bwd or not chain
        ready_ok = comm is None or (getattr(comm, "takes_ready", False) and chain)      # (no chain: y_W_o's bucket goes out while the recurrence runs)
        if (self.device.type == "cuda" and looked and ready_ok and phase_hook is None
                and os.environ.get("LXO_TRAIN_BWD_FUSED", "1") != "0"):
            self.grads.zero_()
            table = None
            if comm:
                evs = self._enc_ready_events()
                table = (ctypes.c_void_p * 7)(*[ctypes.c_void_p(e.cuda_event) if e is not None else None for e in evs])
            self._ck(self.lib.lxo_train_bwd(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._formula), _p(self._img),
                                            _p(self.grads), table, st), "train_bwd")
            if comm:
                comm(*self.buckets[0], ready=evs[0])
                comm(*self.buckets[1], ready=evs[0])
                for (hi, lo), rng in self.enc_buckets:
                    comm(*rng, ready=evs[lo])
            return
        decoder_bwd(chain)
        if chain and not self._xdec_bwd_checked and (self.shape.B, self.shape.H, self.shape.W) not in self._nochain_shapes_bwd:
            # as for the forward chain, after the first backward that RAN it: a chain that did not assemble leaves an error word -> launch
            # chain from now on.  (The words are cleared by every call, chain or not, so `err` is this call's.)
            used, err = self.chain_status(backward=True)
            self.chain_used_bwd = used and not err
            if used or err:
                self._xdec_bwd_checked = True
            else:
                self._nochain_shapes_bwd.add((self.shape.B, self.shape.H, self.shape.W))
            if err:
                self._chain_fallback("backward", err)
                chain = False
                decoder_bwd(False)
        if comm:
            if chain:
                comm(*self.buckets[0])
            comm(*self.buckets[1])
        if phase_hook:
            phase_hook("decoder_bwd")
        if comm and getattr(comm, "takes_ready", False) and self.device.type == "cuda":
            # ONE call for the six layers; the library records an event per layer when its gradients are final (on the weight-gradient
            # side stream when one is bound) and the communication side waits for THAT: the compute stream does not stop at the buckets
            evs = self._enc_ready_events()
            table = (ctypes.c_void_p * 7)(*[ctypes.c_void_p(e.cuda_event) if e is not None else None for e in evs])
            self._ck(self.lib.lxo_encoder_bwd_ready(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._img), _p(self.grads),
                                                    6, 1, table, st), "encoder_bwd_ready")
            for (hi, lo), rng in self.enc_buckets:
                comm(*rng, ready=evs[lo])
        elif comm:
            # one all-reduce per encoder layer as its gradients become final: only conv2 + conv1 (0.3 MB) are left for the end
            for (hi, lo), rng in self.enc_buckets:
                self._ck(self.lib.lxo_encoder_bwd(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._img), _p(self.grads),
                                                  hi, lo, st), "encoder_bwd")
                comm(*rng)
        else:
            self._ck(self.lib.lxo_encoder_bwd(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._img), _p(self.grads),
                                              6, 1, st), "encoder_bwd")

    def _enc_ready_events(self):
        """events[l] for the layers that close a gradient bucket (enc_buckets: 6, 5, 4, 3 and 1) and [0] for the decoder's parameters; recorded
        once here so that their handles exist, re-recorded by lxo_encoder_bwd_ready / lxo_train_bwd every step"""
        if getattr(self, "_enc_ready", None) is None:
            cur = torch.cuda.current_stream(self.device)
            evs = [None] * 7
            for lo in [0] + [lo for (hi, lo), _ in self.enc_buckets]:      # 0: the decoder's parameters (lxo_train_bwd)
                evs[lo] = torch.cuda.Event()
                evs[lo].record(cur)
            self._enc_ready = evs
        return self._enc_ready

    METHODS = {"adam": 0, "sgd": 1, "adagrad": 2, "rmsprop": 3}

    def set_optimizer(self, lr_method):
        """img2seq.py:95-109: 'adam' | 'adagrad' | 'sgd' | 'rmsprop' (TF-1.12 default hyper-parameters)."""
        m = lr_method.lower()
        if m not in self.METHODS:
            raise NotImplementedError("Unknown method {}".format(lr_method))
        self.method = self.METHODS[m]
        if self.method == 2:
            self.adam_v.fill_(0.1)        # Adagrad initial_accumulator_value
        elif self.method == 3:
            self.adam_v.fill_(1.0)        # RMSProp rms slot starts at ones

    def optimizer_step(self, lr, clip=-1.0, beta1=0.9, beta2=0.999, eps=1e-8, guard=False, dp=False):
        """guard: fold the decoder chains' error words into the scale first (train_step does; a failed step is then dropped on the device).
        dp: a data-parallel step -- the guard then runs on EVERY bf16 rank whatever this rank's own step_kernels: a rank that already fell
        back to the launch-per-step kernels must still drop the step a PEER's chain failed in (the peer's NaN probe element arrives through
        the all-reduce), or the replicas part (its own error words are cleared by the no-chain paths, so the check is harmless there)."""
        st = self._stream()
        possible = self._chains_possible() or (dp and self.dtype == _abi.LXO_BF16 and self.device.type == "cuda")
        guard = guard and possible and self.ws is not None and self.shape is not None
        if getattr(self, "method", 0) != 0:
            scale = None
            if clip is not None and clip > 0:
                self._ck(self.lib.lxo_global_norm_scale(self.n_params, _p(self.grads), ctypes.c_float(clip), _p(self.scale), st), "clip")
                scale = self.scale
            if guard:
                self._chain_health_post(scale is not None, dp)
                scale = self.scale
            self._ck(self.lib.lxo_optimizer_step(self.method, self.n_params, _p(self.params), _p(self.grads), _p(self.adam_v),
                                                 ctypes.c_float(float(lr)), _p(scale), st), "optimizer_step")
            self.pack()
            return
        self.adam_t += 1
        lr_t = float(lr) * math.sqrt(1.0 - beta2 ** self.adam_t) / (1.0 - beta1 ** self.adam_t)
        scale = None
        if clip is not None and clip > 0:
            self._ck(self.lib.lxo_global_norm_scale(self.n_params, _p(self.grads), ctypes.c_float(clip), _p(self.scale), st), "clip")
            scale = self.scale
        if guard:
            self._chain_health_post(scale is not None, dp)
            scale = self.scale
        self._ck(self.lib.lxo_adam_step(self.n_params, _p(self.params), _p(self.grads), _p(self.adam_m), _p(self.adam_v),
                                        ctypes.c_float(lr_t), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps),
                                        _p(scale), st), "adam")
        self.pack()

    def train_step(self, img, formula, lengths, lr, clip=-1.0, dist=None, sync_loss=True, dropout=1.0, dropout_seed=None):
        """One optimisation step of img2seq.py:_run_train's body.  Returns the batch loss
        (token mean over the global batch) or None when sync_loss is False.  dropout = config.dropout
        (keep probability, img2seq.py:166); masks are keyed by (dropout_seed or a per-engine step counter)."""
        drop = None
        if dropout is not None and 0.0 < float(dropout) < 1.0:
            self.drop_step = getattr(self, "drop_step", 0) + 1
            world, rank = (dist.world, dist.rank) if dist is not None else (1, 0)
            drop = (float(dropout), dropout_seed if dropout_seed is not None else self.drop_step * world + rank)
        n_local = int(np.asarray(lengths).sum()) if not isinstance(lengths, torch.Tensor) else int(lengths.sum().item())
        if self._health_ring is not None and dist is None:
            # error words of the steps the device has finished meanwhile (no stall).  Not under data parallelism: WHEN the host notices a
            # finished step differs from rank to rank, and the Adam time step a dropped update gives back enters lr_t -- there every rank
            # looks at a step's words at the same distance (the ring slot's reuse, four steps later, or a synchronising caller's wait)
            self._chain_health_poll()
        if dist is not None:
            # the global token count travels rank -> device -> all-reduce -> loss kernel; no host sync inside the step
            ntok, ev = dist.sum_count_async(n_local)
            self.forward(img, formula, dropout=drop, before_decoder=ev)
            stats = self.loss(lengths, ntok_dev=ntok, ntok_event=ev)
        else:
            self.forward(img, formula, dropout=drop)
            stats = self.loss(lengths, 1.0 / float(n_local))
        self.backward(comm=dist.reduce_range_fn(self.grads) if dist is not None else None)
        if dist is not None:
            dist.finish()
        self.optimizer_step(lr, clip, guard=True, dp=dist is not None)
        if not sync_loss:
            return None
        if dist is not None:
            s = stats.clone()
            dist.all_reduce(s)
            s = s.cpu().numpy()
        else:
            s = stats.cpu().numpy()
        if (self._health_ring is not None and self._chain_health_poll(wait=True) and getattr(self, "_last_dropped_seq", -1) == self._health_i
                and dist is None and not getattr(self, "_redoing", False)):
            # the host is synchronised here anyway: a chain failed in THIS step, the device dropped its update, the engine has switched to the
            # launch-per-step kernels -- run the step again so that the caller gets the loss and the update it asked for
            self._redoing = True
            try:
                return self.train_step(img, formula, lengths, lr, clip=clip, dist=None, sync_loss=True, dropout=dropout,
                                       dropout_seed=drop[1] if drop is not None else None)
            finally:
                self._redoing = False
        return float(s[0]) / float(s[1])

    def evaluate_batch(self, img, formula, lengths):
        """(sum CE, n_words) of img2seq.py:74-75 for one batch (teacher forced)."""
        self.forward(img, formula)
        n = int(np.asarray(lengths).sum())
        s = self.loss(lengths, 1.0 / max(n, 1)).cpu().numpy()
        return float(s[0]), n

    # --------------------------------------------------------------- decode --
    def _encode_only(self, img, beam):
        B, H, W = int(img.shape[0]), int(img.shape[1]), int(img.shape[2])
        if beam != self.beam:
            self.beam, self.ws = beam, None
        if self.max_steps <= 0:
            self.max_steps, self.ws = 152, None
        self.ensure(B, H, W, 1)
        self._img = self._to_dev(img, torch.uint8)
        self._ck(self.lib.lxo_encoder_fwd(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), _p(self._img), self._stream()),
                 "encoder_fwd")
        return B

    def _train_chain_batch(self, B, H=None, W=None):
        """The batch the persistent training chains (csrc/xdec.hip: xdec_fwd_kernel / xdec_bwd_kernel) take for B samples -- 8, 16, 32 or 64 --
        or B itself where they would not run anyway (mirrors lxo_launch_xdec_fwd's conditions).  LXO_TRAIN_PAD=0 / Engine.pad_train = False
        switch the padding off (the launch-per-step kernels then take the batch as it is)."""
        d = self.dims
        if (self.device.type != "cuda" or self.dtype != _abi.LXO_BF16 or self.step_kernels != 0 or B >= 64 or B in (8, 16, 32)
                or not getattr(self, "pad_train", True) or d.get("row_bilstm")
                or not (d["C"] == 512 and d["U"] == 512 and d["O"] == 512 and d["E"] == 256)
                or "0" in (os.environ.get("LXO_TRAIN_PAD", "1"), os.environ.get("LXO_XDEC", "1"))):
            return B
        Bp = next(n for n in (8, 16, 32, 64) if B < n)
        if H is not None and not d.get("cnn"):
            from .model.utils.image import encoder_out_hw
            Hp, Wp = encoder_out_hw(int(H), int(W))
            nq = 32 // (Bp // 8)                                    # attention chunks per sample; a chunk's raw scores stay in LDS (xdec.hip: SCMAX rows)
            if (Hp * Wp + nq - 1) // nq > 2432:
                return B
        return Bp

    def _decode_chain_batch(self, B):
        """The batch the persistent greedy-decode chain (csrc/xdec.hip: xdec_dec_kernel) takes for B images -- 8, 16, 32 or 64 -- or B itself where
        the chain would not run anyway (mirrors lxo_launch_xdec_dec's conditions).  LXO_DECODE_PAD=0 switches the padding off."""
        d = self.dims
        if (self.device.type != "cuda" or self.dtype != _abi.LXO_BF16 or self.step_kernels != 0 or B >= 64 or B in (8, 16, 32)
                or not (d["C"] == 512 and d["U"] == 512 and d["O"] == 512 and d["E"] == 256) or self.n_tok > 512
                or "0" in (os.environ.get("LXO_DECODE_PAD", "1"), os.environ.get("LXO_XDEC_DEC", "1"), os.environ.get("LXO_XDEC", "1"))):
            return B
        return next(n for n in (8, 16, 32, 64) if B < n)

    def greedy_decode(self, img, id_end, max_iter=151, return_attention=False):
        """ids int32 [B, T'] as pred_test.ids of the greedy graph (decoder.py:64,70).  With return_attention also the
        attention maps alpha f32 [B, T', H', W'] (what the reference collects through its py_func hook,
        attention_mechanism.py:96-105, for visualize_attention.py).
        A batch the persistent decode chain does not take (B not in {8, 16, 32, 64}) is filled up to the next such size with COPIES of its own
        images (rows are independent and a copy finishes with its original, so neither the ids of the real rows nor the step count change):
        20 images decode in 26 us per step on the chain where the launch-per-step kernels take 44."""
        if self.max_steps < max_iter + 1:
            self.max_steps, self.ws = max_iter + 1, None
        B0 = int(img.shape[0])
        Bp = B0 if return_attention else self._decode_chain_batch(B0)
        if Bp != B0:
            img = self._to_dev(img, torch.uint8)
            img = img[torch.arange(Bp, device=img.device) % B0]
        B = self._encode_only(img, 1)
        ids = torch.zeros(B, self.max_steps, dtype=torch.int32, device=self.device)
        steps = ctypes.c_int(0)
        if not return_attention:
            self._ck(self.lib.lxo_greedy_decode(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), int(id_end), int(max_iter),
                                                _p(ids), ctypes.byref(steps), self._stream()), "greedy_decode")
            return ids[:B0, :steps.value].cpu().numpy()
        from .model.utils.image import encoder_out_hw
        Hp, Wp = encoder_out_hw(int(img.shape[1]), int(img.shape[2]))
        R = Hp * Wp
        Rp = (R + 7) // 8 * 8
        alpha = torch.zeros(self.max_steps, B, Rp, dtype=torch.float32, device=self.device)
        self._ck(self.lib.lxo_greedy_decode_attn(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), int(id_end), int(max_iter),
                                                 _p(ids), _p(alpha), ctypes.byref(steps), self._stream()), "greedy_decode_attn")
        n = steps.value
        a = alpha[:n, :, :R].permute(1, 0, 2).reshape(B, n, Hp, Wp).cpu().numpy()
        return ids[:, :n].cpu().numpy(), a

    # one step at a time: what model/components (the reference's decoder-cell protocol) drives
    def decode_begin(self, img, beam_size=1, max_steps=152, div_gamma=1.0, div_prob=0.0, div_seed=0):
        """initialize(): encoder + attention set-up + initial states for beam_size hypotheses per image."""
        if self.max_steps < max_steps:
            self.max_steps, self.ws = int(max_steps), None
        B = self._encode_only(img, int(beam_size))
        self.shape.div_gamma, self.shape.div_prob = float(div_gamma or 0.0), float(div_prob or 0.0)
        self.shape.div_seed = int(div_seed) & 0x7FFFFFFF
        k = max(1, int(beam_size))
        shp = (B, self.max_steps) if k == 1 else (B, self.max_steps, k)
        self._dec_ids = torch.zeros(*shp, dtype=torch.int32, device=self.device)
        self._dec_par = torch.zeros(*shp, dtype=torch.int32, device=self.device) if k > 1 else None
        self._dec_fin = np.zeros(B * k, dtype=np.int32)
        self._ck(self.lib.lxo_decode_begin(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), self._stream()), "decode_begin")
        return B

    def decode_step(self, time, id_end):
        """step(time): -> (ids [B] or [B, k], parents or None, finished bool [B(, k)], logits f32 [B(, k), V])."""
        un = ctypes.c_int(0)
        self._ck(self.lib.lxo_decode_step(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), int(id_end), int(time),
                                          _p(self._dec_ids), _p(self._dec_par), self._dec_fin.ctypes.data_as(ctypes.c_void_p),
                                          ctypes.byref(un), self._stream()), "decode_step")
        ids = self._dec_ids[:, time].cpu().numpy()
        par = self._dec_par[:, time].cpu().numpy() if self._dec_par is not None else None
        fin = self._dec_fin.astype(bool).reshape(ids.shape)
        Vp = (self.n_tok + 31) // 32 * 32
        logits = self.region("dec_logits", "f32", (ids.size, Vp))[:, :self.n_tok].cpu().numpy().reshape(ids.shape + (self.n_tok,))
        return ids, par, fin, logits

    # the AttentionState as data, and AttentionCell.step alone (model/components/attention_cell.py drives these)
    def _dec_rows(self):
        return int(self.shape.B) * max(1, int(self.beam))

    def decode_get_state(self, time):
        """-> (c [rows, U], h [rows, U], o [rows, O]) float32 host arrays: the state that step `time` starts from."""
        n, U, O = self._dec_rows(), self.dims["U"], self.dims["O"]
        c = torch.empty(n, U, dtype=torch.float32, device=self.device)
        h = torch.empty(n, U, dtype=torch.float32, device=self.device)
        o = torch.empty(n, O, dtype=torch.float32, device=self.device)
        self._ck(self.lib.lxo_decode_state_get(self.sref(), _p(self.ws), int(time), _p(c), _p(h), _p(o), self._stream()), "decode_state_get")
        return c.cpu().numpy(), h.cpu().numpy(), o.cpu().numpy()

    def decode_set_state(self, time, c=None, h=None, o=None, ids=None):
        """Overwrite (parts of) the state that step `time` starts from, and / or the token ids fed with it."""
        n, U, O = self._dec_rows(), self.dims["U"], self.dims["O"]
        dev = []
        for a, shp, dt in ((c, (n, U), torch.float32), (h, (n, U), torch.float32), (o, (n, O), torch.float32), (ids, (n,), torch.int32)):
            if a is None:
                dev.append(None)
                continue
            a = np.ascontiguousarray(a)
            if tuple(a.shape) != shp:
                raise ValueError("decode_set_state: expected shape %s, got %s" % (shp, a.shape))
            dev.append(self._to_dev(a, dt))
        self._ck(self.lib.lxo_decode_state_set(self.sref(), _p(self.ws), int(time), _p(dev[0]), _p(dev[1]), _p(dev[2]), _p(dev[3]),
                                               self._stream()), "decode_state_set")
        torch.cuda.current_stream(self.device).synchronize() if self.device.type == "cuda" else None     # the staging tensors die with this frame

    def decode_cell_step(self, time, start_token):
        """AttentionCell.step alone: state of `time` -> state of time + 1; -> logits float32 [rows, V]."""
        self._ck(self.lib.lxo_decode_cell_step(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), int(time), 1 if start_token else 0,
                                               self._stream()), "decode_cell_step")
        Vp = (self.n_tok + 31) // 32 * 32
        return self.region("dec_logits", "f32", (self._dec_rows(), Vp))[:, :self.n_tok].cpu().numpy()

    def beam_decode(self, img, id_end, beam_size, max_iter=151, return_parents=False, div_gamma=1.0, div_prob=0.0, div_seed=0, return_attention=False):
        """ids int32 [B, T', k] as pred_test.ids of the beam graph before the transpose at img2seq.py:241.
        div_gamma / div_prob: add_div_penalty of beam_search_decoder_cell.py:258-287 (off at 1 / 0, the shipped values).
        return_attention: -> (ids, parents, alpha f32 [B, T', k, H', W']): alpha[b, t, j] = the map decoder row j of image b attended with at
        step t (lxo_beam_decode_attn; the rows the reference's py_func tap sees under config.decoding = "beam_search")."""
        if self.max_steps < max_iter + 1:
            self.max_steps, self.ws = max_iter + 1, None
        B = self._encode_only(img, int(beam_size))
        self.shape.div_gamma, self.shape.div_prob = float(div_gamma or 0.0), float(div_prob or 0.0)
        self.shape.div_seed = int(div_seed) & 0x7FFFFFFF
        ids = torch.zeros(B, self.max_steps, beam_size, dtype=torch.int32, device=self.device)
        par = torch.zeros(B, self.max_steps, beam_size, dtype=torch.int32, device=self.device)
        steps = ctypes.c_int(0)
        if return_attention:
            from .model.utils.image import encoder_out_hw
            Hp, Wp = encoder_out_hw(int(img.shape[1]), int(img.shape[2]))
            R = Hp * Wp
            Rp = (R + 7) // 8 * 8
            alpha = torch.zeros(self.max_steps, B * beam_size, Rp, dtype=torch.float32, device=self.device)
            self._ck(self.lib.lxo_beam_decode_attn(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), int(id_end), int(max_iter),
                                                   _p(ids), _p(par), _p(alpha), ctypes.byref(steps), self._stream()), "beam_decode_attn")
            n = steps.value
            a = alpha[:n, :, :R].reshape(n, B, beam_size, Hp, Wp).permute(1, 0, 2, 3, 4).contiguous().cpu().numpy()
            return ids[:, :n].cpu().numpy(), par[:, :n].cpu().numpy(), a
        self._ck(self.lib.lxo_beam_decode(self.sref(), _p(self.params), _p(self.wpack), _p(self.ws), int(id_end), int(max_iter),
                                          _p(ids), _p(par), ctypes.byref(steps), self._stream()), "beam_decode")
        out = ids[:, :steps.value].cpu().numpy()
        if return_parents:
            return out, par[:, :steps.value].cpu().numpy()
        return out
