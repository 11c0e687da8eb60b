#!/usr/bin/env python
"""Measurement aid: would two half-batch chains on two HIP queues beat one full-batch chain?  Two Engines of batch 32 driven by two
host threads on two streams (no cross-stream waits at all) against one Engine of batch 64: whole training steps, and the decoder
calls alone.  (The C ABI calls enqueue whole loops without the GIL, so the two threads really launch concurrently.)"""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
H, W, V = 128, 512, 500
def mk(B, seed):
    eng = Engine(V, dtype="bf16", seed=0)
    imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=seed)
    img = torch.from_numpy(pad_batch_images(imgs)).cuda()
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    if f.shape[1] < 101:
        f = np.concatenate([f, np.full((B, 101 - f.shape[1]), V - 1, f.dtype)], 1)
    return eng, img, torch.from_numpy(f).cuda(), l
def step(e, what):
    eng, img, fd, l = e
    if what == "step":
        eng.train_step(img, fd, l, 1e-3, sync_loss=False)
    else:
        eng.forward(img, fd); eng.loss(l, 1.0 / float(np.asarray(l).sum())); eng.backward()
def timed(engs, streams, what, n=20):
    def worker(e, s, n):
        with torch.cuda.stream(s):
            for _ in range(n):
                step(e, what)
    for warm in (3, n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(e, s, warm)) for e, s in zip(engs, streams)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
full = mk(64, 1)
h1, h2 = mk(32, 2), mk(32, 3)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
for rep in range(2):
    for what in ("step", "fwd+loss+bwd"):
        a = timed([full], [s0], what)
        b = timed([h1], [s1], what)
        c = timed([h1, h2], [s1, s2], what)
        print("%-14s one engine B=64: %.3f ms | one engine B=32: %.3f ms | two engines B=32 on two threads / streams: %.3f ms per 64 images" % (what, a, b, c))
