"""Run-to-run differences of ONE bf16 training step from identical state (atomics only reorder f32 sums: differences must stay at the
1e-6 level relative to each gradient's largest element; anything larger would be a race).  Toy-set batch of tests/test_gpu_trained.py."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import test_gpu_trained as T

imgs, batches = T._batches()
P0 = {k: v.copy() for k, v in T.Engine(T.V, dtype="f32", seed=0).get_params().items()}
worst = {}
for no, img, f, l in batches[:3]:
    gs, losses = [], []
    for rep in range(4):
        eng = T.Engine(T.V, dtype="bf16", seed=0)
        eng.load_params(P0)
        eng.forward(img, f)
        st = eng.loss(l, 1.0 / int(l.sum())).cpu().numpy().copy()
        eng.backward()
        torch.cuda.synchronize()
        gs.append(eng.grad_dict())
        losses.append(st[0] / st[1])
        del eng
    print("batch %d: losses %s" % (no, ["%.7f" % x for x in losses]))
    for k in gs[0]:
        ref = gs[0][k]
        d = max(np.abs(g[k] - ref).max() for g in gs[1:]) / max(np.abs(ref).max(), 1e-30)
        worst[k] = max(worst.get(k, 0.0), float(d))
for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:12]:
    print("%-60s max |run - run0| / max|g| = %.2e" % (k, v))
