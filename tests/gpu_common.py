"""Shared helpers of the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np
import torch

from latex_ocr_amd import synthetic
from latex_ocr_amd.engine import Engine
from latex_ocr_amd.model.utils.image import pad_batch_images
from latex_ocr_amd.model.utils.text import pad_batch_formulas
from oracle import ref_model as R


def batch(n, H, W, V, lo, hi, seed):
    imgs, forms = synthetic.make_set(n, H, W, V, lo, hi, seed=seed)
    img = pad_batch_images(imgs)
    f, l = pad_batch_formulas(forms, V - 2, V - 1)
    return img, f, l


def oracle_params(eng):
    return {k: torch.from_numpy(v.copy()) for k, v in eng.get_params().items()}


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
