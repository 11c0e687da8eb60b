"""Wall time of the decoder forward alone and of whole training steps, persistent XCD-local chain (lxo_shape.step_kernels = 0) against the
launch-per-step chain (2), same process, same buffers.  B=64, 128x512, V=500, T=101 (the benchmark's configuration)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine, _p
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas

B, H, W, V = 64, 128, 512, 500
imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
f, l = pad_batch_formulas(forms, V - 2, V - 1)
fd = torch.from_numpy(f).cuda()
eng = Engine(V, dtype="bf16", seed=0)
for mode in (0, 2, 0, 2):
    eng.step_kernels = mode
    for _ in range(3):
        eng.train_step(img, fd, l, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.train_step(img, fd, l, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 20
    st = eng._stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        eng._ck(eng.lib.lxo_decoder_train_fwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), st), "fwd")
    e1.record(); e1.synchronize()
    fwd = e0.elapsed_time(e1) / 10
    e0.record()
    for _ in range(10):
        eng._ck(eng.lib.lxo_decoder_train_bwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), _p(eng.grads), st), "bwd")
    e1.record(); e1.synchronize()
    bwd = e0.elapsed_time(e1) / 10
    print("step_kernels=%d: train step %.3f ms; decoder forward alone %.3f ms; decoder backward alone %.3f ms; chain %s" % (mode, step * 1e3, fwd, bwd, eng.chain_status()))
