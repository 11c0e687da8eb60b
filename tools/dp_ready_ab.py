#!/usr/bin/env python
"""The data-parallel step at world size 1 (one-rank RCCL communicator behind the C ABI): per-layer lxo_encoder_bwd calls, each joined with the
compute stream (LXO_DP_READY_EVENTS=0), against ONE lxo_encoder_bwd_ready call whose per-layer events the communication side waits for (default),
and the single-process step without the exchange.  Alternating in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as td
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.dist import DataParallel
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29677")
td.init_process_group("gloo", rank=0, world_size=1)
V, B = 500, 64
dp = DataParallel(device="cuda:0")
eng = Engine(V, dtype="bf16", seed=0)
imgs, forms = synthetic.make_set(B, 128, 512, V, 30, 101, seed=1234)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
f, l = pad_batch_formulas(forms, V - 2, V - 1)
fd = torch.from_numpy(f).cuda()
def timed(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for mode in ("0", "1"):
        os.environ["LXO_DP_READY_EVENTS"] = mode
        print("rep %d  DP step, ready events %s: %.3f ms" % (rep, mode, timed(lambda: eng.train_step(img, fd, l, 1e-3, dist=dp, sync_loss=False))), flush=True)
    print("rep %d  single-process step:       %.3f ms" % (rep, timed(lambda: eng.train_step(img, fd, l, 1e-3, sync_loss=False))), flush=True)
print("chains", eng.chain_used, eng.chain_used_bwd, "failures", eng.chain_failures, "dropped", getattr(eng, "dropped_steps", 0))
dp.close(); td.destroy_process_group()
