"""Where a step of the persistent backward chain goes (csrc/xdec.hip, xdec_bwd_kernel): per-phase durations from in-kernel 100 MHz
timestamps (lxo_xdec_debug_bwd), B=64, 128x512, V=500, T=101.  Phases: Q1 [d_h~ | d_ctx] | barrier | Q2 attention stream | barrier |
Q3 d_att_h + LSTM backward | barrier | Q4 carries + g_{t-1} | barrier.  Also times the decoder backward with and without the chain."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine, _p
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas

B, H, W, V = [int(x) for x in os.environ.get("XS_SHAPE", "64,128,512").split(",")] + [500]
LLO, LHI = [int(x) for x in os.environ.get("XS_LEN", "30,101").split(",")]
imgs, forms = synthetic.make_set(B, H, W, V, LLO, LHI, seed=1234)
img = pad_batch_images(imgs)
f, l = pad_batch_formulas(forms, V - 2, V - 1)
T = f.shape[1]
eng = Engine(V, dtype="bf16", seed=0)
eng.forward(img, f)
eng.loss(l, 1.0 / int(l.sum()))
eng.backward()
torch.cuda.synchronize()
print("backward chain status (used, error):", eng.chain_status(backward=True))


def dec_bwd():
    eng._ck(eng.lib.lxo_decoder_train_bwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), _p(eng.grads), eng._stream()), "bwd")


def timed(n=20):
    for _ in range(3):
        dec_bwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        dec_bwd()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(2):
    eng.shape.step_kernels = 0
    a = timed()
    eng.shape.step_kernels = 2
    b = timed()
    print("decoder backward (free-running, %d calls): chain %.3f ms, launch-per-step %.3f ms" % (20, a, b))
eng.shape.step_kernels = 0
buf = torch.zeros(256 * T * 16, dtype=torch.int64, device="cuda")
eng.lib.lxo_xdec_debug_bwd.argtypes = [ctypes.c_void_p]
eng.lib.lxo_xdec_debug_bwd(ctypes.c_void_p(buf.data_ptr()))
for _ in range(2):
    dec_bwd()
torch.cuda.synchronize()
eng.lib.lxo_xdec_debug_bwd(ctypes.c_void_p(0))
s = buf.cpu().numpy().reshape(256, T, 16).astype(np.float64) * 0.01      # us; slot i = the i-th step in time (t = T - 1 - i)
names = ["Q1 dhc", "barrier 1", "Q2 attention", "barrier 2", "Q3 lstm bwd", "barrier 3", "Q4 carries", "barrier 4"]
d = s[:, 2:, 1:9] - s[:, 2:, 0:8]
print("step (stamp 0 -> 8), mean over workgroups and steps: %.2f us" % (s[:, 2:, 8] - s[:, 2:, 0]).mean())
pw = s[:, 2:, 12] - s[:, 2:, 4]
if (pw > 0).any():
    print("inside Q3: the polled chunk partials arrived %.2f us after the phase start (mean; the wait that used to sit at the XCD barrier behind the stream phase)" % pw[pw > 0].mean())
for i, n in enumerate(names):
    print("%-14s mean %6.2f us   min over workgroups %6.2f   max over workgroups %6.2f" % (n, d[:, :, i].mean(), d[:, :, i].mean(1).min(), d[:, :, i].mean(1).max()))
w = s.reshape(8, 32, T, 16)
for ph, a, b in (("Q1", 0, 1), ("Q2", 2, 3), ("Q3", 4, 5), ("Q4", 6, 7)):
    dur = (w[:, :, 2:, b] - w[:, :, 2:, a])
    print("%s: slowest workgroup of an XCD per step, mean %.2f us; fastest %.2f us" % (ph, dur.max(1).mean(), dur.min(1).mean()))
