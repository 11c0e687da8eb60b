"""Where a step of the persistent XCD-local decoder chain goes: per-phase durations from in-kernel 100 MHz timestamps (lxo_xdec_debug),
B=64, 128x512, V=500, T=101.  Phases: P1 LSTM | barrier | P2 att_h | barrier | P3 attention chunk | barrier | P4 merge + o | barrier."""
import ctypes, os, sys
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
torch.cuda.synchronize()
buf = torch.zeros(256 * T * 16, dtype=torch.int64, device="cuda")
eng.lib.lxo_xdec_debug.argtypes = [ctypes.c_void_p]
eng.lib.lxo_xdec_debug(ctypes.c_void_p(buf.data_ptr()))
for _ in range(2):
    eng._ck(eng.lib.lxo_decoder_train_fwd(eng.sref(), _p(eng.params), _p(eng.wpack), _p(eng.ws), _p(eng._formula), eng._stream()), "fwd")
torch.cuda.synchronize()
eng.lib.lxo_xdec_debug(ctypes.c_void_p(0))
s = buf.cpu().numpy().reshape(256, T, 16).astype(np.float64) * 0.01      # us
names = ["P1 lstm", "barrier 1", "P2 att_h", "barrier 2", "P3 attention", "barrier 3", "P4 merge + o", "barrier 4"]
d = s[:, 2:, 1:9] - s[:, 2:, 0:8]                                          # [wg][t][phase]
print("chain status", eng.chain_status(), " step (stamp 0 -> 8), mean over workgroups and steps: %.2f us" % (s[:, 2:, 8] - s[:, 2:, 0]).mean())
pw = s[:, 2:, 12] - s[:, 2:, 6]
if (pw > 0).any():
    print("inside P4: the polled chunk partials arrived %.2f us after the phase start (mean; the wait that used to sit at the XCD barrier behind the stream phase)" % pw[pw > 0].mean())
for i, n in enumerate(names):
    print("%-14s mean %6.2f us   min over workgroups %6.2f   max over workgroups %6.2f" % (n, d[:, :, i].mean(), d[:, :, i].mean(1).min(), d[:, :, i].mean(1).max()))
# the barrier waits are the phase imbalance: work phase of the slowest workgroup of the XCD
w = s.reshape(8, 32, T, 16)
print("inside P1: A operand arrived %.2f us after the phase start, GEMM + partial tiles %.2f, workgroup barrier %.2f, epilogue + stores issued %.2f" % (
    (s[:, 2:, 9] - s[:, 2:, 0]).mean(), (s[:, 2:, 10] - s[:, 2:, 9]).mean(), (s[:, 2:, 11] - s[:, 2:, 10]).mean(), (s[:, 2:, 1] - s[:, 2:, 11]).mean()))
for ph, a, b in (("P1", 0, 1), ("P2", 2, 3), ("P3", 4, 5), ("P4", 6, 7)):
    dur = (w[:, :, 2:, b] - w[:, :, 2:, a])
    print("%s: slowest workgroup of an XCD per step, mean %.2f us; fastest %.2f us" % (ph, dur.max(1).mean(), dur.min(1).mean()))
