#!/usr/bin/env python
"""Measurement aid: the five conv weight-gradient launches of a training step (B=64, 128x512) through lxo_conv3x3_wgrad, timed with
events (back-to-back repetitions), checked against a float64 einsum on a sub-sample of taps.  LXO_LIB_PATH selects the build."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latex_ocr_amd import _abi
lib = _abi.load()
B = 64
layers = [("conv2", 64, 256, 64, 128, 1), ("conv3", 32, 128, 128, 256, 1), ("conv4", 32, 128, 256, 256, 1), ("conv5", 16, 128, 256, 512, 1), ("conv6", 16, 64, 512, 512, 0)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
tot = 0.0; totf = 0.0
for name, h, w, ci, co, same in layers:
    ho, wo = (h, w) if same else (h - 2, w - 2)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, h, w, ci, dtype=torch.bfloat16, device="cuda", generator=g)
    dy = torch.randn(B, ho, wo, co, dtype=torch.bfloat16, device="cuda", generator=g)
    dw = torch.zeros(9 * ci, co, dtype=torch.float32, device="cuda")
    call = lambda: lib.lxo_conv3x3_wgrad(_abi.LXO_BF16, p(x), p(dy), p(dw), B, h, w, ci, ho, wo, co, 1 if same else 0, st)
    assert call() == 0
    torch.cuda.synchronize()
    # check: tap (kh, kw) of dW = sum in[b, y + kh - pad, x + kw - pad, ci] dy[b, y, x, co]
    pad = 1 if same else 0
    xp = torch.nn.functional.pad(x.float(), (0, 0, pad, pad, pad, pad))
    worst = 0.0
    for kh, kw in ((0, 0), (1, 2), (2, 1)):
        ref = torch.einsum("byxi,byxo->io", xp[:, kh:kh + ho, kw:kw + wo, :].double(), dy.double())
        got = dw[(kh * 3 + kw) * ci:(kh * 3 + kw + 1) * ci].double()
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * B * ho * wo * 9 * ci * co
    tot += us; totf += fl
    print("%s  %8.1f us  %7.1f TFLOP/s   max rel err %.2e" % (name, us, fl / us / 1e6, worst))
print("sum    %8.1f us  %7.1f TFLOP/s  frac of 2500: %.4f" % (tot, totf / tot / 1e6, totf / tot / 1e6 / 2500))
