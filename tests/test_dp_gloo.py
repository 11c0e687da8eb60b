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


# ---- world 4, unequal batches: ShardedBuckets gives every rank the same number of steps and ONE image shape per step ----
SMALL = dict(C=128, E=128, U=128, O=128, D=16)


def _mixed_dataset():
    """22 samples of three image shapes, interleaved, formula lengths 1..6: with batch_size 2 and world 4 the global batch is 8;
    shape A (14 samples) -> steps of 8 and 6 (ranks get 2,2,1,1), shape B (5) -> one step (2,1,1,1), shape C (3 < world) -> dropped."""
    from latex_ocr_amd import synthetic
    from latex_ocr_amd.model.utils.data_generator import ListDataset
    shapes = [(32, 32)] * 14 + [(32, 48)] * 5 + [(40, 32)] * 3
    order = np.random.Generator(np.random.PCG64(5)).permutation(len(shapes))
    imgs, forms = [], []
    for j, i in enumerate(order):
        h, w = shapes[i]
        im, fo = synthetic.make_set(1, h, w, 11, 1, 7, seed=300 + j)
        imgs.append(im[0]); forms.append(fo[0])
    return ListDataset(imgs, forms)


def _worker4(rank, port, q, bf16_grads, host_ordered=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=4)
    from latex_ocr_amd import _abi
    from latex_ocr_amd.dist import DataParallel
    from latex_ocr_amd.engine import Engine
    from latex_ocr_amd.pipeline import ShardedBuckets
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas
    from simlib import SIM_SO
    if host_ordered:
        os.environ["LXO_DP_HOST_ORDERED"] = "force"      # the helper thread that orders the buckets on a GPU, here over CPU tensors
    dp = DataParallel(device="cpu")
    assert dp.host_ordered == bool(host_ordered)
    if bf16_grads:
        dp.grad_dtype = torch.bfloat16
    eng = Engine(11, dims=SMALL, dtype="f32", device="cpu", seed=0, lib=_abi.bind(ctypes.CDLL(SIM_SO)))
    sb = ShardedBuckets(_mixed_dataset(), 2, 4, rank)
    trace = []
    for imgs, forms in sb:
        img = pad_batch_images(imgs)
        f, l = pad_batch_formulas(forms, 9, 10)
        loss = eng.train_step(img, f, l, 1e-3, dist=dp)
        trace.append((tuple(img.shape), int(f.shape[1]), len(imgs), float(loss)))
    q.put((rank, len(sb), trace, float(np.abs(eng.get_params()["Decoder/AttentionCell/rnn/y_W_o"]).sum())))
    td.barrier()
    td.destroy_process_group()


@pytest.mark.parametrize("bf16_grads,host_ordered", [(False, False), (True, False), (False, True)])
def test_four_ranks_unequal_batches_same_steps_no_hang(bf16_grads, host_ordered):
    from simlib import build_sim
    build_sim()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + (7 if bf16_grads else 0) + (13 if host_ordered else 0)
    procs = [ctx.Process(target=_worker4, args=(r, port, q, bf16_grads, host_ordered)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(4))
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    n_steps = [r[1] for r in res]
    assert n_steps == [3, 3, 3, 3]                                         # A: 8 + 6, B: 5; C (3 < world) dropped on every rank
    traces = [r[2] for r in res]
    assert all(len(t) == 3 for t in traces)
    for s in range(3):
        shapes = {t[s][0][1:] for t in traces}
        assert len(shapes) == 1, shapes                                   # one image shape per step on all ranks
        assert len({round(t[s][3], 6) for t in traces}) == 1              # the reported loss is the GLOBAL token mean
    assert sorted(t[1][2] for t in traces) == [1, 1, 2, 2] and sorted(t[2][2] for t in traces) == [1, 1, 1, 2]   # uneven per-rank batches
    assert len({t[1][1] for t in traces} | {t[2][1] for t in traces}) > 1  # ranks pad to their OWN longest formula: T differs
    w = [r[3] for r in res]
    assert max(w) - min(w) <= 1e-6 * max(w), w                            # replicated Adam: identical weights on every rank afterwards
