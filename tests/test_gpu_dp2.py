"""Two ranks on ONE GPU (gloo carries the CUDA tensors; RCCL refuses two ranks per device): the device side of the data-parallel
step -- token count on its own stream, gradient buckets handed to the side stream by the helper thread (host-ordered) or by stream
waits, compute stream joined before Adam -- with real HIP events and two processes that must issue their collectives in the same
order.  2 ranks x 4 samples give the gradients of 1 process x 8 samples; replicated Adam leaves identical weights."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = 40


def _data():
    from latex_ocr_amd import synthetic
    from latex_ocr_amd.model.utils.image import pad_batch_images
    from latex_ocr_amd.model.utils.text import pad_batch_formulas
    imgs, forms = synthetic.make_set(8, 32, 96, V, 3, 9, seed=5)
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    return pad_batch_images(imgs), f, l


def _worker(rank, port, host_ordered, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["LXO_DP_HOST_ORDERED"] = "1" if host_ordered else "0"
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=2)
    from latex_ocr_amd.dist import DataParallel
    from latex_ocr_amd.engine import Engine
    torch.cuda.set_device(0)
    dp = DataParallel(device="cuda:0", comm="torch")      # two ranks on ONE GPU: RCCL refuses duplicate devices, so the data plane is the gloo group here
    assert dp.host_ordered == bool(host_ordered) and dp.lxo is None
    eng = Engine(V, dtype="f32", device="cuda:0", seed=0)
    img, f, l = _data()
    sl = slice(4 * rank, 4 * rank + 4)
    losses = [eng.train_step(img[sl], f[sl], l[sl], 1e-3, dist=dp) for _ in range(3)]
    g = eng.grads.cpu().numpy().copy()
    w = eng.get_params()["Decoder/AttentionCell/rnn/y_W_o"].copy()
    torch.cuda.synchronize()
    q.put((rank, losses, g, w))
    td.barrier()
    dp.close()
    td.destroy_process_group()


@pytest.mark.parametrize("host_ordered", [True, False])
def test_two_ranks_on_one_gpu_equal_one_process(host_ordered):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + os.getpid() % 2000 + (17 if host_ordered else 0)
    procs = [ctx.Process(target=_worker, args=(r, port, host_ordered, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    from latex_ocr_amd.engine import Engine
    eng = Engine(V, dtype="f32", seed=0)
    img, f, l = _data()
    ref = [eng.train_step(img, f, l, 1e-3) for _ in range(3)]
    torch.cuda.synchronize()
    g1 = eng.grads.cpu().numpy()
    w1 = eng.get_params()["Decoder/AttentionCell/rnn/y_W_o"]
    for rank, losses, g, w in res:
        assert np.allclose(losses, ref, rtol=2e-5, atol=0), (losses, ref)          # the global token mean, on every rank
        assert np.abs(g - g1).max() <= 1e-4 * np.abs(g1).max(), np.abs(g - g1).max()   # summed gradients of the third step (f32 sums in another order, two Adam steps earlier)
        assert np.abs(w - w1).max() <= 1e-5                                        # replicated Adam
    assert np.array_equal(res[0][3], res[1][3])
