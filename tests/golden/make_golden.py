#!/usr/bin/env python
"""Generates tests/golden/host_surface.json by IMPORTING the reference's own host-side
modules (model/utils/{general,text,lr_schedule,image}.py import cleanly here; the TF graph
code does not).  Run in the build container only: /root/reference does not exist on the GPU
box, so the tests read the committed JSON, never the reference.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
from model.utils import general as G          # noqa: E402
from model.utils import text as T             # noqa: E402
from model.utils import image as I            # noqa: E402
from model.utils.lr_schedule import LRSchedule  # noqa: E402

out = {}
rng = np.random.default_rng(7)

# minibatches (general.py:15-35)
data = [(i, [i, i + 1]) for i in range(11)]
out["minibatches"] = {str(bs): [[list(x), list(y)] for x, y in G.minibatches(iter(data), bs)] for bs in (1, 3, 4, 11, 20)}

# pad_batch_formulas (text.py:141-164)
forms = [[int(v) for v in rng.integers(0, 40, size=int(n))] for n in (3, 7, 1, 5)]
f, l = T.pad_batch_formulas(forms, 48, 49)
out["pad_batch_formulas"] = {"in": forms, "out": f.tolist(), "len": l.tolist(), "dtype": str(f.dtype)}

# pad_batch_images (image.py:27-44)
imgs = [rng.integers(0, 256, size=(h, w, 1), dtype=np.uint8) for h, w in ((4, 6), (5, 3), (2, 7))]
p = I.pad_batch_images(imgs)
out["pad_batch_images"] = {"in": [a.tolist() for a in imgs], "out": p.tolist(), "dtype": str(p.dtype)}

# greyscale (image.py:67-71)
rgb = rng.integers(0, 256, size=(5, 4, 3), dtype=np.uint8)
out["greyscale"] = {"in": rgb.tolist(), "out": I.greyscale(rgb).tolist()}

# Vocab / load_tok_to_id / form_prepro (text.py:5-63)
with tempfile.TemporaryDirectory() as d:
    vp = os.path.join(d, "vocab.txt")
    toks = ["\\frac", "{", "}", "x", "^", "2", "+", "y"]
    T.write_vocab(toks, vp)
    cfg = G.Config({"unk": "_UNK", "pad": "_PAD", "end": "_END", "path_vocab": vp})
    v = T.Vocab(cfg)
    out["vocab"] = {"tokens": toks, "tok_to_id": v.tok_to_id, "n_tok": v.n_tok, "id_pad": v.id_pad, "id_end": v.id_end,
                    "id_unk": v.id_unk, "prepro_in": "\\frac { x ^ 2 } + z", "prepro_out": v.form_prepro("\\frac { x ^ 2 } + z"),
                    "file": open(vp).read()}

# build_vocab on the shipped small formulas (text.py:93-115)
ds = [(None, line.strip().split(" ")) for line in open(os.path.join(REF, "data/small.formulas/train.norm.txt"))]
out["build_vocab_small_train_min2"] = T.build_vocab([ds], min_count=2)

# LRSchedule traces (lr_schedule.py:27-118), configs/training.json values with 5 batches/epoch, and a score-decay case
def trace(kw, n, scores=None):
    s = LRSchedule(**kw)
    lrs = [s.lr]
    for i in range(n):
        s.update(batch_no=i)
        if scores is not None and i % 5 == 4:
            s.update(score=scores[i // 5])
        lrs.append(s.lr)
    return {"kw": kw, "lrs": lrs, "stop": bool(s.stop_training)}
out["lr_schedule"] = [
    trace(dict(lr_init=1e-3, lr_min=1e-4, start_decay=6 * 5, end_decay=13 * 5, lr_warm=1e-4, end_warm=2 * 5), 80),
    trace(dict(lr_init=1e-3, lr_min=1e-5, start_decay=0, end_decay=20), 30),
    trace(dict(lr_init=1e-2, lr_min=1e-4, start_decay=3, end_decay=None, decay_rate=0.5, early_stopping=2), 30,
          scores=[-3.0, -2.5, -2.6, -2.7, -2.4, -2.9]),
]

# Config merge semantics (general.py:88-109)
with tempfile.TemporaryDirectory() as d:
    a, b = os.path.join(d, "a.json"), os.path.join(d, "b.json")
    json.dump({"export_name": "a.json", "x": 1, "y": 2}, open(a, "w"))
    json.dump({"export_name": "b.json", "y": 3, "z": {"k": 4}}, open(b, "w"))
    c = G.Config([a, b])
    out["config_merge"] = {"x": c.x, "y": c.y, "z": c.z, "export_name": c.export_name}

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_surface.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst)
