#!/usr/bin/env python
"""Measurement aid: per-pixel-block timestamps of conv_wgrad_kernel (conv4: B=64, 32x128, 256 -> 256).  Per workgroup, for the
first 20 blocks: cycles waiting for the block's LDS-DMA (vmcnt), at the barrier, computing; then the atomic epilogue."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latex_ocr_amd import _abi
lib = _abi.load()
raw = ctypes.CDLL(_abi.LIB_PATH)
raw.lxo_wgrad_debug.argtypes = [ctypes.c_void_p]
B = 64
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, h, w, ci, co in (("conv4 wgrad", 32, 128, 256, 256), ("conv2 wgrad", 64, 256, 64, 128)):
    x = torch.randn(B, h, w, ci, dtype=torch.bfloat16, device="cuda")
    dy = torch.randn(B, h, w, co, dtype=torch.bfloat16, device="cuda")
    dw = torch.zeros(9 * ci, co, dtype=torch.float32, device="cuda")
    args = (_abi.LXO_BF16, p(x), p(dy), p(dw), B, h, w, ci, h, w, co, 1, st)
    for _ in range(2):
        assert lib.lxo_conv3x3_wgrad(*args) == 0
    torch.cuda.synchronize()
    dbg = torch.zeros(512 * 64, dtype=torch.int64, device="cuda")
    raw.lxo_wgrad_debug(ctypes.c_void_p(dbg.data_ptr()))
    assert lib.lxo_conv3x3_wgrad(*args) == 0
    torch.cuda.synchronize()
    raw.lxo_wgrad_debug(ctypes.c_void_p(0))
    d = dbg.cpu().numpy().reshape(512, 64)
    d = d[d[:, 0] != 0]
    nb = 19
    top = d[:, 1:1 + 3 * nb:3]; landed = d[:, 2:2 + 3 * nb:3]; bar = d[:, 3:3 + 3 * nb:3]
    nxt = d[:, 4:4 + 3 * nb:3]
    print("== %s: %d workgroups; per pixel block (blocks 1..%d), cycles: median / p90" % (name, len(d), nb - 1))
    for nm, v in (("wait for the block's LDS-DMA", (landed - top)[:, 1:]), ("barrier", (bar - landed)[:, 1:]), ("compute (8 K-steps, 72 MFMA per wave)", (nxt - bar)[:, 1:-1]),
                  ("whole block", (top[:, 2:] - top[:, 1:-1]))):
        print("   %-42s %7d / %7d" % (nm, np.median(v), np.percentile(v, 90)))
    print("   %-42s %7d" % ("first block: start -> data landed", np.median(landed[:, 0] - d[:, 0])))
    print("   %-42s %7d" % ("atomic epilogue", np.median(d[:, 62] - d[:, 61])))
