#!/usr/bin/env python
"""Freezes outputs of oracle/ref_model.py on a tiny seeded case into tests/golden/oracle_small.npz
so that (a) drift of the oracle is caught, (b) the NumPy float64 restatement, the hipsim build and
the real-GPU path can all be compared with the same committed numbers.

    python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from latex_ocr_amd import synthetic                      # noqa: E402
from latex_ocr_amd.model.utils.image import pad_batch_images   # noqa: E402
from latex_ocr_amd.model.utils.text import pad_batch_formulas  # noqa: E402
from oracle import ref_model as R                        # noqa: E402

V, B, H, W = 11, 2, 32, 48
imgs, forms = synthetic.make_set(B, H, W, V, 3, 6, seed=5)
img = pad_batch_images(imgs)
f, l = pad_batch_formulas(forms, V - 2, V - 1)
P = R.init_params(V, 0)
ti, tf, tl = torch.from_numpy(img), torch.from_numpy(f), torch.from_numpy(l)
enc = R.encoder(P, ti)
logits, alpha = R.decoder_train(P, enc, tf, True)
loss, G, ce, nw = R.train_grads(P, ti, tf, tl)
ids = R.greedy_decode(P, ti, V - 1, max_iter=8)
bids, bpar = R.beam_decode(P, ti, V - 1, 2, max_iter=8)
gn = {k.replace("/", "__"): v.numpy() for k, v in G.items() if k.endswith("bias") or k.endswith("att_beta") or k.endswith("start_token")}
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz"),
                    img=img, formula=f, lengths=l, enc=enc.numpy(), logits=logits.numpy(), alpha=alpha.numpy(),
                    loss=np.float64(loss), ce_words=np.float64(ce), n_words=np.int64(nw), greedy_ids=ids.numpy(),
                    beam_ids=bids.numpy(), beam_parents=bpar.numpy(), **gn)
print("loss", float(loss), "greedy", ids.shape, "beam", bids.shape)
