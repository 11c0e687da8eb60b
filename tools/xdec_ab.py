"""A/B timing of the decoder forward and backward alone (the two persistent chains + their surrounding launches) for one setting of the
environment: B=64, 128x512, V=500, T=101.  Run it several times per box with different LXO_XDEC_* settings, alternating (boxes differ
by a few percent, and so do consecutive processes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine, _p
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas

B, H, W, V = 64, 128, 512, 500
imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
img = pad_batch_images(imgs)
f, l = pad_batch_formulas(forms, V - 2, V - 1)
eng = Engine(V, dtype="bf16", seed=0)
eng.forward(img, f)
eng.loss(l, 1.0 / int(l.sum()))
eng.backward()
torch.cuda.synchronize()
st = eng._stream()
fw = lambda: eng._ck(eng.lib.lxo_decoder_train_fwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), st), "fwd")
bw = lambda: eng._ck(eng.lib.lxo_decoder_train_bwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), _p(eng.grads), st), "bwd")
res = []
for fn in (fw, bw):
    best = 1e9
    for rep in range(4):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    res.append(best)
print("LL=%s  decoder forward %.3f ms, backward %.3f ms (best of 4 x 20 calls); chains %s %s" % (
    os.environ.get("LXO_XDEC_LL", "-"), res[0], res[1], eng.chain_status(), eng.chain_status(backward=True)))
