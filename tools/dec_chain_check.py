"""Greedy decode through the persistent decode chain (default) or the launch-per-step kernels (LXO_XDEC_DEC=0): ids + timing, for B in {8, 16, 32, 64}.
Writes gpurun_out/dec_ids_<tag>.npz; run twice (with and without LXO_XDEC_DEC=0) and compare with `python tools/dec_chain_check.py --compare a b`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 3 and sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k])
        n = min(a[k].shape[1], b[k].shape[1])
        agree = float((a[k][:, :n] == b[k][:, :n]).mean())
        print("%-14s shapes %s %s identical %s agreement %.4f" % (k, a[k].shape, b[k].shape, same, agree))
    sys.exit(0)
import torch
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
tag = sys.argv[1] if len(sys.argv) > 1 else "x"
V = 500
out = {}
for B in (64, 32, 16, 8):
    imgs, _ = synthetic.make_set(B, 128, 512, V, 30, 101, seed=5)
    img = torch.from_numpy(pad_batch_images(imgs)).cuda()
    eng = Engine(V, dtype="bf16", seed=3, max_steps=152)
    ids = eng.greedy_decode(img, V - 1, max_iter=60)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ids = eng.greedy_decode(img, V - 1, max_iter=100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out["b%d" % B] = ids
    print("chain_status (tickets taken, error word) after the last decode:", eng.chain_status())
    print("B=%d: %d steps, %.2f ms per batch (%.1f us per step incl. the encoder); distinct ids %d; first row %s" % (B, ids.shape[1], dt * 1e3, dt * 1e6 / ids.shape[1], len(np.unique(ids)), ids[0, :8]))
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/dec_ids_%s.npz" % tag, **out)
