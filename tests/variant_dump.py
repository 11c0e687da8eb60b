"""Helper of tests/test_zz_gpu_variants.py (run as a subprocess: the library reads its A/B switches once per process).
Runs one forward / loss / backward (bf16 unless a dtype is given) on a fixed seeded batch and stores loss statistics + every gradient."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_common import Engine, batch  # noqa


def main(out, h, w, b, dtype="bf16"):
    V = 120
    img, f, l = batch(b, h, w, V, 5, 24, seed=77)
    eng = Engine(V, dtype=dtype, seed=5)          # LXO_STEP_KERNELS=1 (read by Engine) selects round 1's split-K step kernels
    eng.forward(img, f)
    stats = eng.loss(l, 1.0 / int(l.sum())).cpu().numpy()
    eng.backward()
    torch.cuda.synchronize()
    g = {k.replace("/", "__"): v for k, v in eng.grad_dict().items()}
    np.savez(out, stats=stats, **g)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "bf16")
