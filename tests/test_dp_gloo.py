"""CPU, world_size 2 over gloo: the data-parallel path (global token count, bucketed gradient
all-reduce hooks of Engine.backward, latex_ocr_amd/dist.py) gives the same gradients as one
process on the whole batch.  The ranks execute the shipped HIP sources under hipsim."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(V):
    from latex_ocr_amd import _abi
    from latex_ocr_amd.engine import Engine
    from simlib import SIM_SO
    return Engine(V, dtype="f32", device="cpu", seed=0, lib=_abi.bind(ctypes.CDLL(SIM_SO)))


def _data():
    from latex_ocr_amd import synthetic
    imgs, forms = synthetic.make_set(2, 32, 48, 11, 2, 5, seed=21)
    return imgs, forms


def _grads(eng, imgs, forms, n_global, dist=None):
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas
    img = pad_batch_images(imgs)
    f, l = pad_batch_formulas(forms, 9, 10)
    if dist is not None:
        # the product's data-parallel step: the global token count reaches the loss kernel through device memory
        ntok, ev = dist.sum_count_async(int(l.sum()))
        eng.forward(img, f)
        eng.loss(l, ntok_dev=ntok, ntok_event=ev)
        assert float(ntok[0]) == n_global
    else:
        eng.forward(img, f)
        eng.loss(l, 1.0 / n_global)
    eng.backward(comm=dist.reduce_range_fn(eng.grads) if dist is not None else None)
    if dist is not None:
        dist.finish()
    return eng.grads.numpy().copy(), int(l.sum())


def _worker(rank, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=2)
    from latex_ocr_amd.dist import DataParallel
    dp = DataParallel(device="cpu")
    imgs, forms = _data()
    n_local = len(forms[rank]) + 1
    n_global = dp.sum_scalar(n_local)
    g, _ = _grads(_engine(11), imgs[rank:rank + 1], forms[rank:rank + 1], n_global, dp)
    if rank == 0:
        q.put((g, n_global))
    td.barrier()
    td.destroy_process_group()


def test_two_ranks_equal_one_process():
    from simlib import build_sim
    build_sim()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    imgs, forms = _data()
    n = sum(len(f) + 1 for f in forms)
    ref, n1 = _grads(_engine(11), imgs, forms, float(n))
    g, n_global = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert n_global == n == n1
    # white padding differs (each rank pads to its own max formula length only; images are equal size here)
    err = np.abs(g - ref).max() / np.abs(ref).max()
    assert err < 1e-5, err
