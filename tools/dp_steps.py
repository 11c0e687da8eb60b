#!/usr/bin/env python
"""N data-parallel training steps at world size 1 (profiling aid: rocprofv3 --kernel-trace -- python tools/dp_steps.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as td
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.dist import DataParallel
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29689")
td.init_process_group("gloo", rank=0, world_size=1)
V, B = 500, 64
dp = DataParallel(device="cuda:0")
eng = Engine(V, dtype="bf16", seed=0)
imgs, forms = synthetic.make_set(B, 128, 512, V, 30, 101, seed=1234)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
f, l = pad_batch_formulas(forms, V - 2, V - 1)
fd = torch.from_numpy(f).cuda()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for _ in range(4): eng.train_step(img, fd, l, 1e-3, dist=dp, sync_loss=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): eng.train_step(img, fd, l, 1e-3, dist=dp, sync_loss=False)
torch.cuda.synchronize()
print("DP step %.3f ms" % ((time.perf_counter() - t0) / N * 1e3))
dp.close(); td.destroy_process_group()
