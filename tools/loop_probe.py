#!/usr/bin/env python
"""Is the decoder's recurrent loop bound by the host (launch enqueue rate) or by the device?

Times lxo_decoder_train_fwd / _bwd (B=64, 128x512, V=500, T=101, bf16) twice:
  live    : HIP events around the call on an idle stream (what a training step sees),
            plus the host wall time the call needs to ENQUEUE its ~700 launches;
  blocked : the same call enqueued while a spin kernel holds the stream (torch.cuda._sleep), so the host's
            enqueue cost is off the clock and the events see the device-only duration of the chain.
Prints one JSON line; tools/launch_probe.hip is the stand-alone version with empty / body kernels."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from latex_ocr_amd import synthetic                                   # noqa: E402
from latex_ocr_amd.engine import Engine, _p                            # noqa: E402
from latex_ocr_amd.model.utils.image import pad_batch_images           # noqa: E402
from latex_ocr_amd.model.utils.text import pad_batch_formulas          # noqa: E402


def main():
    B, H, W, V = 64, 128, 512, 500
    dev = "cuda:0"
    imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
    img = torch.from_numpy(pad_batch_images(imgs)).to(dev)
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    f_d = torch.from_numpy(f).to(dev)
    eng = Engine(V, dtype="bf16", device=dev, seed=0)
    for _ in range(3):
        eng.train_step(img, f_d, l, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    T = int(f.shape[1])
    st = eng._stream()
    lib = eng.lib
    spin = int(25e-3 * 2.4e9)          # ~25 ms at 2.4 GHz (cycles of the shader clock): longer than any enqueue below

    def fwd():
        eng._ck(lib.lxo_decoder_train_fwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), st), "fwd")

    def bwd():
        eng.grads.zero_()
        eng._ck(lib.lxo_decoder_train_bwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), _p(eng.grads), st), "bwd")

    def measure(fn, blocked):
        out = []
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if blocked:
                torch.cuda._sleep(spin)
            e0.record()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            e1.record()
            e1.synchronize()
            out.append((e0.elapsed_time(e1), (t1 - t0) * 1e3))
        return min(x[0] for x in out), min(x[1] for x in out)

    # a forward must precede the loss/backward so that the workspace holds a consistent step
    eng.forward(img, f_d)
    eng.loss(l, 1.0 / float(l.sum()))
    res = {"T": T, "B": B}
    for name, fn in (("decoder_train_fwd", fwd), ("decoder_train_bwd", bwd)):
        live_ms, host_ms = measure(fn, False)
        dev_ms, _ = measure(fn, True)
        res[name] = {"live_ms": round(live_ms, 3), "host_enqueue_ms": round(host_ms, 3), "device_only_ms": round(dev_ms, 3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
