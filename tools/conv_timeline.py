#!/usr/bin/env python
"""Measurement aid: per-CU timeline of conv_halo2wg_kernel workgroups (conv4 forward).  Groups the stamped workgroups by
the CU they ran on (HW_ID stamp) inside one XCD (cycle counters of different XCDs are not aligned) and prints, for a few
CUs, every workgroup's start, slice boundaries and end relative to the first start on that CU."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latex_ocr_amd import _abi
lib = _abi.load()
raw = ctypes.CDLL(_abi.LIB_PATH)
raw.lxo_conv_debug.argtypes = [ctypes.c_void_p]
B, h, w, ci, co = 64, 32, 128, 256, 256
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
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
ns = ci // 32
hw = d[:, 63]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
xcd = np.arange(nwg) % 8
key = se * 32 + sh * 16 + cu
for X in (0, 3):
    ks = sorted(set(key[xcd == X]))
    print("XCD %d: %d distinct (se, sh, cu) keys" % (X, len(ks)))
    for k in ks[:3]:
        idx = np.where((xcd == X) & (key == k))[0]
        idx = idx[np.argsort(d[idx, 0])]
        t0 = d[idx, 0].min()
        print("  CU key %d: %d workgroups" % (k, len(idx)))
        for i in idx:
            r = d[i, :ns + 3] - t0
            print("    wg %5d simd %d: start %7d | slices %s | loop end %7d | end %7d" % (i, (hw[i] >> 4) & 3, r[0], " ".join("%6d" % v for v in np.diff(r[1:ns + 2])), r[ns + 1], r[ns + 2]))
