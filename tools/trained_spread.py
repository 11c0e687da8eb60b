"""Distribution of the whole-curve statistics of tests/test_gpu_trained.py for the bf16 engine: N runs of the 100 Adam steps on the same
batches against ONE oracle curve (and its control).  Prints, per run: spread (max of the 5-step moving average of |log loss - log oracle|),
its median over the steps, the last-10 mean ratio, the first-10 max relative difference."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_gpu_trained as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
imgs, batches = T._batches()
P0 = {k: torch.from_numpy(v.copy()) for k, v in T.Engine(T.V, dtype="f32", seed=0).get_params().items()}
n = T._threads()
ref, _ = T._oracle_curve(P0, batches, n)
ctl, _ = T._oracle_curve(P0, batches, max(1, n // 2 - 1))


def crossings(c, levels):
    """step (fractional) at which the 5-step moving geometric mean of curve c first falls to each level"""
    sm = np.convolve(np.log(c), np.ones(5) / 5, mode="valid")
    out = []
    for L in levels:
        i = int(np.argmax(sm <= np.log(L)))
        if sm[i] > np.log(L):
            out.append(float("nan")); continue
        out.append(float(i) if i == 0 else i - 1 + (sm[i - 1] - np.log(L)) / (sm[i - 1] - sm[i]))
    return np.array(out)


def shift(a, b):
    smb = np.exp(np.convolve(np.log(b), np.ones(5) / 5, mode="valid"))
    levels = smb[0] * (smb.min() / smb[0]) ** (np.arange(1, 9) / 9.0)
    return np.abs(crossings(a, levels) - crossings(b, levels))


def stats(a, b):
    d = np.abs(np.log(a) - np.log(b))
    ma = np.convolve(d, np.ones(5) / 5, mode="valid")
    return ma.max(), float(np.median(d)), float(np.mean(d)), a[-10:].mean() / b[-10:].mean(), (np.abs(a - b) / b)[:10].max(), int(ma.argmax())


print("control: spread %.3f median %.4f mean %.4f last10 ratio %.3f first10 %.1e (argmax %d)" % stats(ctl, ref), "; shift max %.2f" % np.nanmax(shift(ctl, ref)))
curves = []
for dtype, reps in (("bf16", N), ("f32", 1)):
    for r in range(reps):
        got, eng = T._engine_curve(dtype, P0, batches)
        del eng
        curves.append(got)
        sh = shift(got, ref)
        print("%s run %2d: spread %.3f median %.4f mean %.4f last10 ratio %.3f first10 %.1e (argmax step %d); level-crossing shift max %.2f steps %s" % (
            (dtype, r) + stats(got, ref) + (np.nanmax(sh), np.round(sh, 2).tolist())))
c = np.array(curves[:N])
pair = [stats(c[i], c[j])[0] for i in range(N) for j in range(i + 1, N)]
print("bf16 run vs bf16 run: spread min %.3f median %.3f max %.3f" % (min(pair), float(np.median(pair)), max(pair)))
print("geometric mean curve of the bf16 runs vs oracle: spread %.3f median %.4f mean %.4f last10 ratio %.3f first10 %.1e (argmax %d)" % stats(np.exp(np.log(c).mean(0)), ref))
