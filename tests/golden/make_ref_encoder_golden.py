#!/usr/bin/env python
"""Pins the encoder half of the oracle to REFERENCE CODE ACTUALLY RUN HERE.

The reference's TensorFlow graph cannot be executed in this container (tensorflow==1.12.2 needs
Python <= 3.6), but the reference also carries its own PyTorch statement of the same conv stack and
positional signal: /root/reference/model/components/seq2seq_torch.py:24-55 (`EncoderCNN("vanilla")`:
the same six 3x3 convolutions + ReLU, floor max-pools == TF SAME pools on even extents) and :113-156
(`add_timing_signal_nd_torch`, the torch twin of model/components/positional.py:10-65).  That module
imports once `torchvision` and `tensorflow` (unused by EncoderCNN) are stubbed in sys.modules.

This script imports it, loads the oracle's seed-0 conv weights into it (HWIO -> OIHW), runs it on
seeded synthetic crops at 32x128 (config 1) and 128x512 (the benchmark shape) and writes
tests/golden/ref_encoder.npz.  tests/test_oracle.py compares oracle/ref_model.py:encoder() with these
outputs.  Build container only -- /root/reference does not exist on the GPU box, the tests read the
committed .npz.

    python tests/golden/make_ref_encoder_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# stubs for the two imports of seq2seq_torch.py:6,9,11 that EncoderCNN never touches
tv = types.ModuleType("torchvision")
tv.models = types.ModuleType("torchvision.models")
sys.modules.setdefault("torchvision", tv)
sys.modules.setdefault("torchvision.models", tv.models)
sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))

from model.components import seq2seq_torch as REFT   # noqa: E402   (the reference's module)
from oracle import ref_model as R                      # noqa: E402
from latex_ocr_amd import synthetic                    # noqa: E402


class Cfg(object):
    def __init__(self, positional):
        self.encoder_cnn = "vanilla"
        self.positional_embeddings = positional


def ref_encoder(P, img_u8, positional):
    enc = REFT.EncoderCNN(Cfg(positional))
    convs = [m for m in enc.cnn if isinstance(m, torch.nn.Conv2d)]
    assert len(convs) == 6
    pre = "Encoder/convolutional_encoder/conv2d"
    with torch.no_grad():
        for i, m in enumerate(convs):
            sfx = "" if i == 0 else "_%d" % i
            m.weight.copy_(P[pre + sfx + "/kernel"].permute(3, 2, 0, 1))      # HWIO -> OIHW
            m.bias.copy_(P[pre + sfx + "/bias"])
        # model/encoder.py:26-27 normalisation, then the reference's NCHW module; it returns NHWC (seq2seq_torch.py:99)
        x = (torch.from_numpy(img_u8).to(torch.float32) - 128.0) / 128.0
        return enc(x.permute(0, 3, 1, 2).contiguous()).numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    P = R.init_params(50, seed=0)
    # non-zero conv biases so that the bias path is pinned as well (the TF initialiser leaves them at zero)
    rng = np.random.Generator(np.random.PCG64(11))
    for k in list(P):
        if k.startswith("Encoder") and k.endswith("/bias"):
            P[k] = torch.from_numpy(rng.uniform(-0.05, 0.05, size=tuple(P[k].shape)).astype(np.float32))
    out = {"bias_seed": np.int64(11)}
    small, _ = synthetic.make_set(2, 32, 128, 50, 5, 9, seed=21)
    big, _ = synthetic.make_set(1, 128, 512, 50, 5, 9, seed=22)
    s = np.stack(small); b = np.stack(big)
    out["img_32x128"] = s
    out["img_128x512"] = b
    out["feat_32x128_nopos"] = ref_encoder(P, s, False)
    out["feat_32x128_pos"] = ref_encoder(P, s, True)
    fb = ref_encoder(P, b, False)
    fbp = ref_encoder(P, b, True)
    assert fb.shape == (1, 14, 62, 512), fb.shape
    # the full-size map is 1.8 MB in f32: keep a strided sample plus per-channel float64 sums of the whole map
    out["feat_128x512_nopos_sample"] = fb[:, ::3, ::5, ::4].copy()
    out["feat_128x512_pos_sample"] = fbp[:, ::3, ::5, ::4].copy()
    out["feat_128x512_nopos_chansum"] = fb.astype(np.float64).sum(axis=(0, 1, 2))
    out["feat_128x512_pos_chansum"] = fbp.astype(np.float64).sum(axis=(0, 1, 2))
    np.savez_compressed(os.path.join(HERE, "ref_encoder.npz"), **out)
    print("wrote ref_encoder.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
