#!/usr/bin/env python
"""Soak: N training steps of the headline workload (default path: two streams, one-call backward, persistent chains), then N data-parallel steps at
world size 1 (RCCL behind the C ABI, per-bucket ready events): every step's chain error words are looked at (Engine.chain_failures / dropped_steps),
the loss must fall and stay finite.   python tools/soak.py [N]      SOAK_SHAPE=B,H,W (default 64,128,512) SOAK_LEN=lo,hi (default 30,101)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as td
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.dist import DataParallel
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
V = 500
B, H, W = [int(x) for x in os.environ.get("SOAK_SHAPE", "64,128,512").split(",")]
LLO, LHI = [int(x) for x in os.environ.get("SOAK_LEN", "30,101").split(",")]
sets = []
for s in range(4):
    imgs, forms = synthetic.make_set(B, H, W, V, LLO, LHI, seed=100 + s)
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    sets.append((torch.from_numpy(pad_batch_images(imgs)).cuda(), torch.from_numpy(f).cuda(), l))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29688")
td.init_process_group("gloo", rank=0, world_size=1)
dp = DataParallel(device="cuda:0")
for name, dist in (("single process", None), ("data parallel, world 1", dp)):
    eng = Engine(V, dtype="bf16", seed=0)
    t0 = time.perf_counter(); first = last = None
    for i in range(N):
        img, f, l = sets[i % 4]
        loss = eng.train_step(img, f, l, 3e-4, dist=dist, sync_loss=(i % 500 == 0 or i == N - 1))
        if i % 500 == 0 or i == N - 1:
            first = loss if first is None else first; last = loss
            print("  %s step %5d loss %.4f" % (name, i, loss), flush=True)
    torch.cuda.synchronize()
    eng._chain_health_poll(wait=True)
    dt = time.perf_counter() - t0
    print("%s: %d steps in %.1f s (%.3f ms per step incl. the loss read-backs); chains fwd/bwd %s/%s, chain failures %d, dropped steps %d, loss %.4f -> %.4f" % (
        name, N, dt, dt / N * 1e3, eng.chain_used, eng.chain_used_bwd, eng.chain_failures, getattr(eng, "dropped_steps", 0), first, last), flush=True)
    assert eng.chain_failures == 0 and getattr(eng, "dropped_steps", 0) == 0 and np.isfinite(last) and last < first
dp.close(); td.destroy_process_group()
print("soak OK (batch %d, %d x %d)" % (B, H, W))
