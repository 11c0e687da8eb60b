"""Workload for PMC passes over the REAL launches of a training step (every conv_halo2wg instantiation a step uses -- the pooled forwards
<4, 4, 2, 2> / <4, 4, 2, 1> / <4, 4, 1, 2> included, which the stand-alone tools/pmc_conv.py cannot reach through lxo_conv3x3_ex -- conv_wgrad,
the two decoder chains): 3 steps of the benchmark configuration (B = 64, 128 x 512, V = 500), one stream."""
import os, sys
os.environ.setdefault("LXO_ENC_OVERLAP", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
V = 500
imgs, forms = synthetic.make_set(64, 128, 512, V, 30, 101, seed=1234)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
f, l = pad_batch_formulas(forms, V - 2, V - 1)
fd = torch.from_numpy(f).cuda()
eng = Engine(V, dtype="bf16")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    eng.train_step(img, fd, l, 1e-3, sync_loss=False)
torch.cuda.synchronize()
print("done")
