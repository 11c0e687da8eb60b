#!/usr/bin/env python
"""Measurement aid: does a SECOND active HIP queue slow the dependent-launch chains of the decoder?  Times the benchmark step (a) plain,
(b) with a tiny kernel enqueued on another stream once per step, (c) with that stream also waiting on / being waited by the main one."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
B, H, W, V = 64, 128, 512, 500
eng = Engine(V, dtype="bf16", seed=0)
imgs, forms = synthetic.make_set(B, H, W, V, 30, 101, seed=1234)
img = torch.from_numpy(pad_batch_images(imgs)).cuda()
f, l = pad_batch_formulas(forms, V - 2, V - 1)
fd = torch.from_numpy(f).cuda()
other = torch.cuda.Stream()
x = torch.zeros(64, device="cuda")
def run(mode, n=30):
    for it in range(n + 5):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode >= 1:
            if mode in (2, 3):
                other.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(other):
                x.add_(1.0)
            if mode in (2, 4):
                torch.cuda.current_stream().wait_stream(other)
            if mode == 5:
                ev = torch.cuda.Event(); ev.record(other); ev.synchronize()
        eng.train_step(img, fd, l, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def run_hooked(kind, n=30):
    """the step with a per-bucket hook inside backward (what data parallel does): kind 0 = no-op hook, 1 = record an event on the compute
    stream per bucket, 2 = record + the second stream waits for it, 3 = the same + a tiny kernel there + main waits at the end"""
    n_local = int(np.asarray(l).sum())
    evs = []
    def hook(lo, hi):
        if kind >= 1:
            ev = torch.cuda.Event(); ev.record(); evs.append(ev)
            if kind >= 2:
                other.wait_event(ev)
            if kind >= 3:
                with torch.cuda.stream(other):
                    x.add_(1.0)
    for it in range(n + 5):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        evs.clear()
        eng.forward(img, fd)
        eng.loss(l, 1.0 / n_local)
        eng.backward(comm=hook)
        if kind >= 3:
            torch.cuda.current_stream().wait_stream(other)
        eng.optimizer_step(1e-3)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for kind, name in ((0, "bucket hook: no-op"), (1, "bucket hook: event record on the compute stream"), (2, "... + second stream waits for it"),
                       (3, "... + kernel there + main waits at the end")):
        print("%-50s %.3f ms/step" % (name, run_hooked(kind)))
    for mode, name in ((0, "plain"), (1, "tiny kernel on a second stream per step"), (2, "the same, with event waits both ways"), (3, "only: second stream waits for the main one"),
                       (4, "only: main stream waits for the second one"), (5, "host waits for the second stream")):
        print("%-50s %.3f ms/step" % (name, run(mode)))

# ---- alternatives to "event recorded on the compute stream + another stream waits for it" (through the HIP runtime directly) ----
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
main_s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
other_s = ctypes.c_void_p(other.cuda_stream)
def mk_event(flags):
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
    return e
DIS_TIMING, DIS_FENCE, REL_DEV = 0x2, 0x20000000, 0x40000000
sig = ctypes.c_void_p()
rc_sig = 1        # hipExtMallocWithFlags(hipMallocSignalMemory) is refused here (hipErrorInvalidValue): the stream value ops cannot be tried
counter = [0]
def run_alt(kind, n=30):
    ev = {1: mk_event(DIS_TIMING), 2: mk_event(DIS_TIMING | DIS_FENCE), 3: mk_event(DIS_TIMING | REL_DEV)}.get(kind)
    for it in range(n + 5):
        if it == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if kind in (1, 2, 3):
            assert hip.hipEventRecord(ev, main_s) == 0
            assert hip.hipStreamWaitEvent(other_s, ev, 0) == 0
        elif kind == 4:
            counter[0] += 1
            assert hip.hipStreamWriteValue32(main_s, sig, counter[0], 0) == 0
            assert hip.hipStreamWaitValue32(other_s, sig, counter[0], 0, 0xFFFFFFFF) == 0
        with torch.cuda.stream(other):
            x.add_(1.0)
        eng.train_step(img, fd, l, 1e-3, sync_loss=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("signal memory rc", rc_sig)
for rep in range(2):
    for kind, name in ((0, "no dependency"), (1, "hip event (timing off) + hipStreamWaitEvent"), (2, "... + hipEventDisableSystemFence"),
                       (3, "... + hipEventReleaseToDevice"), (4, "hipStreamWriteValue32 / hipStreamWaitValue32")):
        if kind == 4 and rc_sig != 0:
            continue
        print("%-50s %.3f ms/step" % (name, run_alt(kind)))

# ---- the same dependency when the compute stream is NOT the null stream ----
comp = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(comp):
    for rep in range(2):
        for mode, name in ((0, "non-default compute stream: plain"), (3, "non-default compute stream: second stream waits for it"), (2, "non-default compute stream: waits both ways")):
            print("%-50s %.3f ms/step" % (name, run(mode)))
