"""Greedy / beam decode latency on synthetic 128x512 crops (random-init weights never emit END, so every run decodes
the full max_iter + 1 steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
V, B = 500, 64
imgs, _ = synthetic.make_set(B, 128, 512, V, 30, 101, seed=5)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
for beam in (1, 5):
    eng = Engine(V, dtype="bf16", beam=beam, max_steps=152)
    fn = (lambda: eng.greedy_decode(img, V - 1, max_iter=100)) if beam == 1 else (lambda: eng.beam_decode(img, V - 1, beam, max_iter=100))
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("beam %d: %d steps, %.1f ms per batch of %d (%.1f us per step, %.0f img/s)" % (beam, out.shape[1], dt * 1e3, B, dt * 1e6 / out.shape[1], B / dt))
