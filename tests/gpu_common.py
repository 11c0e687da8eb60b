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


def assert_flips_are_near_ties(ids, ref_ids, ref_logits, what, k=8.0):
    """Token-for-token is the f32 parity mode's guarantee; bf16 storage may flip an arg-max -- but ONLY where the reference's own logits
    leave a near-tie.  For every row, at its FIRST divergence from the reference ids (later ids of that row follow another input history):
    the token the HIP path chose must sit within k * 2^-8 * max|logit| of the reference's top-1 IN THE REFERENCE'S LOGITS (2^-8 = one bf16
    rounding; k covers the six bf16-stored conv layers and the decoder GEMM operands in front of the logit).  -> number of diverging rows."""
    ids, ref_ids = np.asarray(ids), np.asarray(ref_ids)
    ref_logits = np.asarray(ref_logits, np.float64)
    T = min(ids.shape[1], ref_ids.shape[1])
    n_div = 0
    for b in range(ids.shape[0]):
        d = np.nonzero(ids[b, :T] != ref_ids[b, :T])[0]
        if not len(d):
            continue
        n_div += 1
        t = int(d[0])
        row = ref_logits[b, t]
        gap = float(row[ref_ids[b, t]] - row[ids[b, t]])
        bar = k * 2.0 ** -8 * float(np.abs(row).max())
        print("%s: row %d diverges at step %d: hip %d reference %d; the reference's logit gap between the two %.3e (bar %.3e)" % (what, b, t, ids[b, t], ref_ids[b, t], gap, bar))
        assert 0.0 <= gap <= bar, (what, b, t, gap, bar)
    return n_div
