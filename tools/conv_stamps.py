#!/usr/bin/env python
"""Measurement aid: per-slice timestamps of conv_halo2wg_kernel (one launch of conv4 forward: B=64, 32x128, 256 -> 256).
For the first workgroups to start on each CU slot: cycles from kernel start to the first K-step (prologue), the gaps between
consecutive K-step barriers (split by position inside the 9-tap slice), K loop end -> epilogue end."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latex_ocr_amd import _abi
lib = _abi.load()
raw = ctypes.CDLL(_abi.LIB_PATH)
raw.lxo_conv_debug.argtypes = [ctypes.c_void_p]
B = 64
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, h, w, ci, co in (("conv4 fwd", 32, 128, 256, 256), ("conv2 fwd", 64, 256, 64, 128), ("conv5 fwd", 16, 128, 256, 512)):
    x = torch.randn(B, h, w, ci, dtype=torch.bfloat16, device="cuda")
    wp = (torch.randn(co, 9 * ci, dtype=torch.bfloat16, device="cuda") * 0.05)
    bias = torch.zeros(co, dtype=torch.float32, device="cuda")
    y = torch.empty(B, h, w, co, dtype=torch.bfloat16, device="cuda")
    nwg = B * (h // 8) * (w // 32) * (co // 128)
    dbg = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
    args = (_abi.LXO_BF16, p(x), p(wp), p(bias), p(y), B, h, w, ci, h, w, co, 1, 1, st)
    for _ in range(2):
        assert lib.lxo_conv3x3(*args) == 0
    torch.cuda.synchronize()
    raw.lxo_conv_debug(ctypes.c_void_p(dbg.data_ptr()))
    assert lib.lxo_conv3x3(*args) == 0
    torch.cuda.synchronize()
    raw.lxo_conv_debug(ctypes.c_void_p(0))
    d = dbg.cpu().numpy().reshape(nwg, 64)
    ns = ci // 32                                      # 32-channel slices: stamp 1 + c at the head of slice c, 1 + ns after the K loop
    rel = d[:, :ns + 3] - d[:, :1]
    pro = rel[:, 1]
    gaps = np.diff(rel[:, 1:ns + 2], axis=1)          # ns gaps: slice c start -> slice c + 1 start (last: -> K loop end)
    epi = rel[:, ns + 2] - rel[:, ns + 1]
    tot = rel[:, ns + 2]
    print("== %s: %d workgroups, %d slices of 9 taps x 16 MFMAs per tile" % (name, nwg, ns))
    print("   prologue (start -> first slice)   median %7d  p10 %7d  p90 %7d" % (np.median(pro), np.percentile(pro, 10), np.percentile(pro, 90)))
    for c in range(ns):
        g = gaps[:, c]
        print("   slice %d (144 MFMAs per wave)      median %7d  p10 %7d  p90 %7d" % (c, np.median(g), np.percentile(g, 10), np.percentile(g, 90)))
    print("   epilogue                          median %7d  p10 %7d  p90 %7d" % (np.median(epi), np.percentile(epi, 10), np.percentile(epi, 90)))
    for col, nm in ((40, "all waves out of the K loop"), (41, "own bf16 tile written"), (42, "all tiles in LDS"), (ns + 2, "rows stored")):
        v = d[:, col] - d[:, ns + 1]
        print("   epilogue: %-28s median %7d  (since this wave left the K loop)" % (nm, np.median(v)))
    print("   whole tile                        median %7d   (144 MFMA x 32 cycles x %d slices = %d cycles of MFMA issue per wave)" % (np.median(tot), ns, 4608 * ns))
