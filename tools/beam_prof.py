"""Beam-5 decode alone (B = 64, 128x512, V = 500, 60 steps) for a kernel trace: rocprofv3 --kernel-trace --stats -- python tools/beam_prof.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
V, B = 500, 64
imgs, _ = synthetic.make_set(B, 128, 512, V, 30, 101, seed=5)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
eng = Engine(V, dtype="bf16", beam=5, max_steps=152)
for _ in range(2):
    out = eng.beam_decode(img, -1, 5, max_iter=59)
torch.cuda.synchronize()
print("steps", out.shape[1])
