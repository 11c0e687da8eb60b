#!/usr/bin/env python
"""Measurement aid: forward / backward time of the benchmark step with and without the padded-step-skipping extension (active rows)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
B, H, W, V = 64, 128, 512, 500
eng = Engine(V, dtype="bf16", seed=0)
imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
f, l = pad_batch_formulas(forms, V - 2, V - 1)
fd = torch.from_numpy(f).cuda()
im2, f2, l2, act = eng.sort_by_length(img, fd, l)
print("T", f.shape[1], "active rows: mean %.1f" % act.mean(), "first zero step", int((act > 0).sum()))
def ev():
    return torch.cuda.Event(enable_timing=True)
for name, (i_, f_, l_, a_) in (("all rows", (img, fd, l, None)), ("active rows", (im2, f2, l2, act))):
    tf = tb = 0.0; n = 10
    for it in range(n + 2):
        e0, e1, e2, e3 = ev(), ev(), ev(), ev()
        e0.record(); eng.forward(i_, f_, active_rows=a_); e1.record()
        eng.loss(l_, 1.0 / float(np.asarray(l).sum()))
        e2.record(); eng.backward(); e3.record()
        eng.optimizer_step(1e-3)
        torch.cuda.synchronize()
        if it >= 2:
            tf += e0.elapsed_time(e1); tb += e2.elapsed_time(e3)
    t0 = time.perf_counter()
    for it in range(n):
        eng.forward(i_, f_, active_rows=a_)
    th = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    print("%-12s forward (encoder + decoder) %.3f ms, backward %.3f ms; host time to enqueue a forward %.3f ms" % (name, tf / n, tb / n, th))
