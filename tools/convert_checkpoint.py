#!/usr/bin/env python
"""Convert a checkpoint between the three containers, keyed by the TF variable names of SURVEY.md Appendix B:

  npz          this framework's file at the reference's path  results/.../model_weights/model.cpkt-<epoch>
  safetensors  <name>.safetensors
  tf           a tf.train.Saver bundle prefix (<prefix>.index + <prefix>.data-00000-of-00001), the reference's format

  python tools/convert_checkpoint.py <src> <dst> [--to npz|safetensors|tf] [--weights-only]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_any(path):
    from latex_ocr_amd.model.base import BaseModel
    return BaseModel._open_checkpoint(path)


def save_any(arrays, dst, kind):
    if kind == "npz":
        with open(dst, "wb") as f:
            np.savez(f, **arrays)
    elif kind == "safetensors":
        from safetensors.numpy import save_file
        save_file({k: np.ascontiguousarray(v) for k, v in arrays.items()}, dst)
    elif kind == "tf":
        from latex_ocr_amd.tf_checkpoint import write_bundle
        out = dict(arrays)
        if "optimize/adam_t" in out:         # tf.train.AdamOptimizer stores the powers, not the step count
            t = int(np.asarray(out.pop("optimize/adam_t")).reshape(-1)[0])
            out["optimize/beta1_power"] = np.float32(0.9 ** (t + 1))      # TF holds beta^(t+1) after t steps
            out["optimize/beta2_power"] = np.float32(0.999 ** (t + 1))
        write_bundle(dst, {k: v for k, v in out.items() if not k.startswith("lxo/")})
    else:
        raise ValueError(kind)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--to", default=None, choices=["npz", "safetensors", "tf"])
    ap.add_argument("--weights-only", action="store_true", help="drop optimizer slots and schedule state")
    a = ap.parse_args(argv)
    kind = a.to or ("safetensors" if a.dst.endswith(".safetensors") else "npz")
    arrays = {k: np.asarray(v) for k, v in load_any(a.src).items()}
    if a.weights_only:
        arrays = {k: v for k, v in arrays.items() if not k.startswith("optimize/") and not k.startswith("lxo/")}
    save_any(arrays, a.dst, kind)
    print("wrote %d tensors (%s) to %s" % (len(arrays), kind, a.dst))
    return arrays


if __name__ == "__main__":
    main()
