"""Image batching helpers (host, byte work).

Mirrors the on-path part of `model/utils/image.py` of the reference.  The
LaTeX->PNG rendering half (pdflatex / ImageMagick, image.py:164-246) is out of
scope (offline dataset build).
"""
import numpy as np


def get_max_shape(arrays):
    """Per-axis max over a list of arrays (reference: image.py:15-24)."""
    return [max(dims) for dims in zip(*[a.shape for a in arrays])]


def pad_batch_images(images, max_shape=None):
    """White (255) canvas at the batch-max shape, each image pasted top-left,
    result uint8.  Reference: model/utils/image.py:27-44 (which builds the
    canvas in float64 and casts; the bytes are identical)."""
    if max_shape is None:
        max_shape = get_max_shape(images)
    out = np.full([len(images)] + list(max_shape), 255, dtype=np.uint8)
    for i, img in enumerate(images):
        out[i, :img.shape[0], :img.shape[1]] = img
    return out


def greyscale(state):
    """RGB -> luma with the reference's weights, truncating cast to uint8,
    keepdims channel.  Reference: model/utils/image.py:67-71."""
    g = state[:, :, 0] * 0.299 + state[:, :, 1] * 0.587 + state[:, :, 2] * 0.114
    return g[:, :, np.newaxis].astype(np.uint8)


def downsample(state):
    """Every second pixel on the first two axes (reference: image.py:74-81)."""
    return state[::2, ::2, :]


def encoder_out_hw(H, W):
    """(H', W') of the encoder feature map for an H x W input: three /2 (SAME, ceil) reductions per axis and the
    VALID 3x3 conv (model/encoder.py:37-59; the same arithmetic as visualize_attention.py:22-31 getWH)."""
    c = lambda n: -(-n // 2)
    return c(c(c(H))) - 2, c(c(c(W))) - 2
