#!/usr/bin/env python
"""Measurement aid: in-kernel phase timestamps of the fused recurrent-step kernels (csrc/rstep.hip), last decoder step of
a B=64 / 128x512 / V=500 / T=101 bf16 training step.  Per workgroup the deltas to ITS OWN start stamp (the cycle counters
of different XCDs are not aligned), reduced to min / median / max over the grid."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
B, H, W, V = 64, 128, 512, 500
imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
f, l = pad_batch_formulas(forms, V - 2, V - 1)
f_d = torch.from_numpy(f).cuda()
eng = Engine(V, dtype="bf16", device="cuda:0", seed=0)
for _ in range(2):
    eng.train_step(img, f_d, l, 1e-3, sync_loss=False)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(ROOT, "latex_ocr_amd", "liblxo.so"))
lib.lxo_rstep_debug.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["start", "requests issued", "chunk0 staged", "chunk0 computed", "all computed", "reduced", "epilogue done"]
for epi, label in ((2, "LSTM_FWD"), (1, "TANH_O"), (0, "PLAIN (last launch of a step: the tall d_emb GEMM over all T*B rows)"), (3, "LSTM_BWD"), (4, "CARRY (t=0: raw)")):
    dbg = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
    lib.lxo_rstep_debug(ctypes.c_void_p(dbg.data_ptr()), epi)
    eng.forward(img, f_d)
    eng.loss(l, 1.0 / float(l.sum()))
    eng.backward()
    torch.cuda.synchronize()
    lib.lxo_rstep_debug(ctypes.c_void_p(0), -1)
    d = dbg.cpu().numpy().reshape(256, 8)
    d = d[(d[:, 0] != 0) & (d[:, 6] != 0)]           # workgroups whose stamping thread ran the epilogue
    grid = len(d)
    rel = d[:, :7] - d[:, :1]
    print("== %s, %d workgroups: cycles since the workgroup's own start, min / median / max" % (label, grid))
    for i, n in enumerate(names):
        print("   %-18s %8d %8d %8d" % (n, rel[:, i].min(), np.median(rel[:, i]), rel[:, i].max()))
