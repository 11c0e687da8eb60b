"""What the bf16 deterministic mode costs, launch family by launch family: one instrumented training step (HIP events around every conv / chain
launch, lxo_timing_*) of the headline workload with Engine(deterministic=False) and (deterministic=True), plus free-running step times.
The weight-gradient kernel's range spread comes from LXO_WG_STAGGER as usual (run the script once per value)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
import bench

B, H, W, V = 64, 128, 512, 500
imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
f, l = pad_batch_formulas(forms, V - 2, V - 1)
f_d = torch.from_numpy(f).cuda()
print("LXO_WG_STAGGER=%s" % os.environ.get("LXO_WG_STAGGER", "(default 10)"))
for det in (False, True, False, True):
    eng = Engine(V, dtype="bf16", seed=0, deterministic=det)
    for _ in range(5):
        eng.train_step(img, f_d, l, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.train_step(img, f_d, l, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    recs, phases = bench.instrumented_step(eng, img, f_d, l, torch)
    fam = {}
    for fa, nm, w, s in recs:
        a = fam.setdefault(fa, [0, 0.0, 0.0]); a[0] += 1; a[1] += w; a[2] += s
    print("deterministic=%d: %.3f ms/step free-running; phases %s" % (det, ms, phases))
    for k, (n, w, s) in sorted(fam.items()):
        print("    %-12s x%-3d %8.1f us total%s" % (k, n, s * 1e6, ("  %.1f TFLOP/s" % (w / s / 1e12)) if k.startswith("conv") else ""))
    wg = [(nm, s) for fa, nm, w, s in recs if fa == "conv_wgrad"]
    print("    conv_wgrad per layer (us): " + "  ".join("%s %.0f" % (nm, s * 1e6) for nm, s in wg))
    del eng
