"""Measurement aid (not part of the product): HIP-event timing of the implicit-GEMM conv kernel for the ten launches one training
step makes with it (conv2..conv6 forward and data gradient), each through the C ABI entry lxo_conv3x3_ex on its own.
bench.py's roofline uses IN-STEP timings (lxo_timing_*); this stand-alone loop exists for PMC passes (tools/pmc_conv.py)."""
import ctypes

import torch

from latex_ocr_amd import _abi
from latex_ocr_amd.engine import _p


def time_conv_gemms(self, B, H, W, reps=5):
    """HIP-event timing (on the stream the kernels are launched on) of the implicit-GEMM conv
    kernel for the ten launches one training step makes with it: conv2..conv6 forward and
    dgrad.  Returns (sum of algorithmic FLOPs, sum of average launch seconds, per-launch list)."""
    assert self.dtype == _abi.LXO_BF16
    c = lambda n: -(-n // 2)
    H1, W1 = c(H), c(W); H2, W2 = c(H1), c(W1); H4 = c(H2); W5 = c(W2)
    C = self.dims["C"]
    layers = [("conv2", H1, W1, 64, 128, 1), ("conv3", H2, W2, 128, 256, 1), ("conv4", H2, W2, 256, 256, 1),
              ("conv5", H4, W2, 256, C, 1), ("conv6", H4, W5, C, C, 0)]
    bf = dict(dtype=torch.bfloat16, device=self.device)
    st = self._stream()
    out, flops_tot, t_tot = [], 0.0, 0.0
    for name, h, w, ci, co, same in layers:
        ho, wo = (h, w) if same else (h - 2, w - 2)
        for kind in ("fwd", "dgrad"):
            # the same fused epilogues the training step uses: conv6 forward adds the timing signal and keeps the
            # pre-addend copy; conv4's dgrad applies conv3's ReLU mask and accumulates its bias gradient
            addend = pre = ref = cs = None
            if kind == "fwd":
                x = torch.randn(B, h, w, ci, **bf); wp = torch.randn(co, 9 * ci, **bf) * 0.05
                y = torch.empty(B, ho, wo, co, **bf)
                if name == "conv6":
                    addend = torch.randn(ho * wo, co, dtype=torch.float32, device=self.device); pre = torch.empty_like(y)
                args = (self.dtype, _p(x), _p(wp), None, _p(y), B, h, w, ci, ho, wo, co, 1 if same else 0, 1,
                        _p(addend), ho * wo, _p(pre), None, None, st)
                flops, n_out = 2.0 * B * ho * wo * co * 9 * ci, co
            else:
                x = torch.randn(B, ho, wo, co, **bf); wp = torch.randn(ci, 9 * co, **bf) * 0.05
                y = torch.empty(B, h, w, ci, **bf)
                if name == "conv4":
                    ref = torch.randn(B, h, w, ci, **bf); cs = torch.zeros(ci, dtype=torch.float32, device=self.device)
                args = (self.dtype, _p(x), _p(wp), None, _p(y), B, ho, wo, co, h, w, ci, 1 if same else 2, 0,
                        None, 1, None, _p(ref), _p(cs), st)
                flops, n_out = 2.0 * B * h * w * ci * 9 * co, ci
            for _ in range(2):
                self._ck(self.lib.lxo_conv3x3_ex(*args), "conv3x3_ex")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
            for _ in range(reps):
                self._ck(self.lib.lxo_conv3x3_ex(*args), "conv3x3_ex")
            e1.record(torch.cuda.current_stream(self.device))
            e1.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / reps
            out.append({"launch": name + "_" + kind, "kernel": "conv_halo2wg_kernel" if n_out % 64 == 0 else "conv_halo_kernel",
                        "us": round(sec * 1e6, 1), "tflops": round(flops / sec / 1e12, 1), "flops": flops})
            flops_tot += flops; t_tot += sec
    return flops_tot, t_tot, out

