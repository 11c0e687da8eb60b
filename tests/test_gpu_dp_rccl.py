"""-m gpu, needs TWO GPUs (skipped on the one-GPU boxes this repository is built on; runs by itself wherever two devices are visible):
the data-parallel step over RCCL through the C ABI (lxo_comm_init / lxo_allreduce_bucket, csrc/comm.hip) with more than one rank.

* f32 parity mode: 2 ranks x B/2 == 1 rank x B -- losses 2e-5, summed gradients <= 1e-5 of their largest element,
  replicated Adam leaves bit-identical weights on the two ranks;
* bf16, the persistent decoder chains ON (256 spin-waiting workgroups per launch next to RCCL's kernels of a peer that may lag): 50 steps
  at 16 samples per rank, `rccl_ranks_seen == 2`, zero chain failures, zero dropped steps, identical weights on the two ranks, the loss falls.

Reference: the exchange replaces nothing in the reference (it is single-device); what it must preserve is the loss of
model/img2seq.py:69-71 -- the mean over ALL unmasked tokens of the global batch -- and one optimizer step per sess.run (:169)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = 40


def _data(n, seed=5):
    from latex_ocr_amd import synthetic
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas
    imgs, forms = synthetic.make_set(n, 32, 96, V, 3, 9, seed=seed)
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    return pad_batch_images(imgs), f, l


def _worker(rank, port, mode, q, host_ordered="0"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["LXO_DP_HOST_ORDERED"] = host_ordered            # "0": buckets ordered by stream waits (the default), "1": by the helper thread
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=2)      # control plane only: the 128-byte RCCL id, host barriers
    from latex_ocr_amd.dist import DataParallel
    from latex_ocr_amd.engine import Engine
    dev = "cuda:%d" % rank
    torch.cuda.set_device(rank)
    dp = DataParallel(device=dev)                                # data plane: RCCL behind the C ABI
    assert dp.lxo is not None and dp.lxo.ranks_seen == 2, "the RCCL communicator must report two ranks"
    if mode == "f32":
        eng = Engine(V, dtype="f32", device=dev, seed=0)
        img, f, l = _data(8)
        sl = slice(4 * rank, 4 * rank + 4)
        losses = [eng.train_step(img[sl], f[sl], l[sl], 1e-3, dist=dp) for _ in range(3)]
        torch.cuda.synchronize()
        q.put((rank, losses, eng.grads.cpu().numpy().copy(), eng.params.cpu().numpy().copy(), {}))
    else:
        eng = Engine(V, dtype="bf16", device=dev, seed=0)
        img, f, l = _data(32, seed=9)
        sl = slice(16 * rank, 16 * rank + 16)
        losses = []
        for s in range(50):
            out = eng.train_step(img[sl], f[sl], l[sl], 1e-3, dist=dp, sync_loss=(s % 10 == 9 or s == 0))
            if out is not None:
                losses.append(out)
        torch.cuda.synchronize()
        eng._chain_health_poll(wait=True)
        info = {"chain_used": eng.chain_used, "chain_used_bwd": eng.chain_used_bwd, "chain_failures": eng.chain_failures,
                "dropped": getattr(eng, "dropped_steps", 0), "adam_t": eng.adam_t, "step_kernels": eng.step_kernels, "ranks_seen": dp.lxo.ranks_seen}
        q.put((rank, losses, None, eng.params.cpu().numpy().copy(), info))
    td.barrier()
    dp.close()
    td.destroy_process_group()


def _run(mode, host_ordered="0"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 34500 + os.getpid() % 2000 + (31 if mode == "f32" else 0) + (7 if host_ordered == "1" else 0)
    procs = [ctx.Process(target=_worker, args=(r, port, mode, q, host_ordered)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    return res


def test_two_rccl_ranks_equal_one_process_f32():
    res = _run("f32")
    from latex_ocr_amd.engine import Engine
    eng = Engine(V, dtype="f32", seed=0)
    img, f, l = _data(8)
    ref = [eng.train_step(img, f, l, 1e-3) for _ in range(3)]
    torch.cuda.synchronize()
    g1 = eng.grads.cpu().numpy()
    w1 = eng.params.cpu().numpy()
    for rank, losses, g, w, _ in res:
        assert np.allclose(losses, ref, rtol=2e-5, atol=0), (losses, ref)
        assert np.abs(g - g1).max() <= 1e-4 * np.abs(g1).max(), np.abs(g - g1).max()      # third step's gradients (two Adam steps of f32 reordering earlier)
        assert np.abs(w - w1).max() <= 1e-5
    assert np.array_equal(res[0][3], res[1][3])                 # replicas bit-identical


@pytest.mark.parametrize("host_ordered", ["0", "1"])
def test_fifty_steps_with_the_chains_beside_rccl_bf16(host_ordered):
    res = _run("bf16", host_ordered)
    for rank, losses, _, w, info in res:
        assert info["ranks_seen"] == 2
        assert info["chain_used"] and info["chain_used_bwd"], info
        assert info["chain_failures"] == 0 and info["dropped"] == 0 and info["adam_t"] == 50 and info["step_kernels"] == 0, info
        assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    assert np.array_equal(res[0][3], res[1][3])                 # the same all-reduced gradients -> the same weights, bit for bit
    assert np.allclose(res[0][1], res[1][1], rtol=1e-6)
