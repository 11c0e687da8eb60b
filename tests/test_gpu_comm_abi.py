"""-m gpu: the data-parallel entry points of the C ABI on RCCL (include/lxo.h: lxo_comm_unique_id / lxo_comm_init / lxo_comm_info /
lxo_allreduce_bucket / lxo_comm_destroy), called the way a non-PyTorch binding would: raw device pointers, a HIP
stream and a HIP event.  One GPU per box here, so the communicator has ONE rank: the sum over ranks is the identity, which still
exercises id creation, ncclCommInitRank, every dtype, stream ordering behind the ready event, and tear-down.  The N-rank arithmetic
is covered on CPU by tests/test_dp_gloo.py (same DataParallel code, torch.distributed data plane) and on hardware by the driver's
bench.py --gpus N run, which reports rccl_ranks_seen."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from latex_ocr_amd import _abi


def test_comm_lifecycle_and_allreduce_world1():
    lib = _abi.load()
    torch.cuda.set_device(0)
    idbuf = ctypes.create_string_buffer(_abi.LXO_COMM_ID_BYTES)
    assert lib.lxo_comm_unique_id(idbuf) == 0, lib.lxo_comm_last_error()
    assert any(idbuf.raw)
    comm = ctypes.c_void_p()
    assert lib.lxo_comm_init(idbuf, 0, 1, ctypes.byref(comm)) == 0, lib.lxo_comm_last_error()
    r, w = ctypes.c_int(-1), ctypes.c_int(-1)
    assert lib.lxo_comm_info(comm, ctypes.byref(r), ctypes.byref(w)) == 0 and (r.value, w.value) == (0, 1)
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    for dt, code in ((torch.float32, _abi.LXO_F32), (torch.bfloat16, _abi.LXO_BF16), (torch.int32, _abi.LXO_I32)):
        x = (torch.arange(100003, device="cuda") % 251).to(dt)
        want = x.clone()
        y = x * 2                                            # produced on the compute stream, reduced on the side stream behind the event
        ev = torch.cuda.Event(); ev.record(main)
        rc = lib.lxo_allreduce_bucket(comm, ctypes.c_void_p(y.data_ptr()), y.numel(), code, ctypes.c_void_p(side.cuda_stream),
                                      ctypes.c_void_p(ev.cuda_event))
        assert rc == 0, lib.lxo_comm_last_error()
        main.wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(y, want * 2)
    assert lib.lxo_allreduce_bucket(comm, None, 4, _abi.LXO_F32, None, None) != 0           # null pointer refused
    assert lib.lxo_allreduce_bucket(comm, ctypes.c_void_p(x.data_ptr()), 4, 77, ctypes.c_void_p(side.cuda_stream), None) != 0   # unknown dtype
    assert b"dtype" in lib.lxo_comm_last_error()
    assert lib.lxo_comm_destroy(comm) == 0
    assert lib.lxo_comm_init(idbuf, 1, 1, ctypes.byref(comm)) != 0                           # rank outside the world


def test_dataparallel_uses_the_abi_comm_world1():
    """latex_ocr_amd.dist.DataParallel on a GPU: gradient buckets and the token count go through lxo_allreduce_bucket"""
    import os
    import torch.distributed as td
    from gpu_common import Engine, batch
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29644")
    td.init_process_group("gloo", rank=0, world_size=1)
    try:
        from latex_ocr_amd.dist import DataParallel
        dp = DataParallel(device="cuda:0")
        assert dp.lxo is not None and dp.lxo.ranks_seen == 1
        V = 50
        img, f, l = batch(6, 32, 128, V, 3, 9, seed=3)
        a = Engine(V, dtype="f32", seed=0); b = Engine(V, dtype="f32", seed=0)
        la = [a.train_step(img, f, l, 1e-3, dist=dp) for _ in range(3)]
        lb = [b.train_step(img, f, l, 1e-3) for _ in range(3)]
        torch.cuda.synchronize()
        assert np.allclose(la, lb, rtol=1e-6, atol=0), (la, lb)           # the (identity) exchange changes nothing; 1 / n_tok is formed on the device in one case, on the host in the other
        assert float((a.params - b.params).abs().max()) <= 1e-6
        dp.close()
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("host_ordered", ["1", "0"])
def test_dp_step_stress_world1_chains_beside_rccl(host_ordered):
    """200 data-parallel steps at world size 1 with the persistent decoder chains (bf16, B = 8): ONE communicator driven from two host
    threads (main: token count on its stream; helper: gradient buckets on the side stream) on three streams, the chains kept clear of
    the collectives, lxo_chain_guard behind the exchange -- in both bucket-ordering modes.  Every step's chain error words are looked at
    (Engine.chain_failures / dropped_steps stay 0), the run ends on the same weights as the same 200 steps without the exchange to
    f32-atomic noise, and nothing hangs.  (The N-rank run itself is the driver's: no box here has a second GPU.)"""
    import os
    import torch.distributed as td
    from gpu_common import Engine, batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29%03d" % (650 + int(host_ordered))
    old = os.environ.get("LXO_DP_HOST_ORDERED")
    os.environ["LXO_DP_HOST_ORDERED"] = host_ordered
    td.init_process_group("gloo", rank=0, world_size=1)
    try:
        from latex_ocr_amd.dist import DataParallel
        dp = DataParallel(device="cuda:0")
        assert dp.lxo is not None and dp.lxo.ranks_seen == 1 and dp.host_ordered == (host_ordered == "1")
        V = 50
        batches = [batch(8, 32, 128, V, 3, 9, seed=40 + i) for i in range(4)]
        a = Engine(V, dtype="bf16", seed=0); b = Engine(V, dtype="bf16", seed=0)
        for i in range(200):
            img, f, l = batches[i % 4]
            la = a.train_step(img, f, l, 1e-3, dist=dp, sync_loss=(i % 50 == 49))
        torch.cuda.synchronize()
        a._chain_health_poll(wait=True)
        assert a.chain_used and a.chain_used_bwd and a.chain_failures == 0 and getattr(a, "dropped_steps", 0) == 0 and a.adam_t == 200
        for i in range(200):
            img, f, l = batches[i % 4]
            lb = b.train_step(img, f, l, 1e-3, sync_loss=(i % 50 == 49))
        torch.cuda.synchronize()
        print("host_ordered=%s: loss after 200 steps with / without the exchange: %.4f / %.4f" % (host_ordered, la, lb))
        assert np.isfinite(la) and np.isfinite(lb) and abs(la - lb) <= 0.3 * abs(lb) + 0.05, (la, lb)       # two bf16 (atomic-order) trajectories of 200 steps
        assert la < 3.9 and torch.isfinite(a.params).all()                                # below ln 50: it trained
        dp.close()
    finally:
        td.destroy_process_group()
        if old is None:
            os.environ.pop("LXO_DP_HOST_ORDERED", None)
        else:
            os.environ["LXO_DP_HOST_ORDERED"] = old

