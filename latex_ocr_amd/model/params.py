"""Variable inventory and initialisation contract of the hot path (host).

Names are the TF checkpoint names the reference's graph would produce
(scopes from model/encoder.py:25-59, model/decoder.py:41-57,
model/components/attention_mechanism.py:43,79,87,149-150,
model/components/attention_cell.py:78-80).  Shapes keep TF layout: conv kernels
HWIO, dense kernels [in, out].

The reference never seeds its RNG (quirk C-13); this module DEFINES the
contract: NumPy Generator(PCG64(seed)), one float64 draw per variable in
`param_specs` order, stored float32:
  * kernels and the attention-init biases: glorot-uniform (TF-1.12 default of
    tf.get_variable / tf.layers.*), limit = sqrt(6 / (fan_in + fan_out));
  * conv biases, LSTM bias: zeros (tf.layers.conv2d / LSTMCell defaults);
  * embedding_table / start_token: uniform(-1, 1) then L2-normalised over the
    last axis (model/decoder.py:98-105).
"""
import math
from collections import OrderedDict

import numpy as np

DEFAULT_DIMS = dict(C=512, E=256, U=512, O=512, D=80)
_A = "Decoder/AttentionCell/"


def dims_from_config(config):
    ac = getattr(config, "attn_cell_config", {}) or {}
    enc = getattr(config, "encoder_cnn", "vanilla")
    if enc not in ("vanilla", "cnn"):
        raise NotImplementedError("encoder_cnn=%r (model/encoder.py knows 'vanilla' and 'cnn')" % (enc,))
    return dict(C=512, E=ac.get("dim_e", 256), U=ac.get("num_units", 512),
                O=ac.get("dim_o", 512), D=ac.get("dim_embeddings", 80), cnn=(enc == "cnn"),
                positional=bool(getattr(config, "positional_embeddings", True)),
                # optional row encoder (north_star's row-BiLSTM; NOT in the reference, off unless configs/model.json says "encoder_rnn": "bilstm")
                row_bilstm=(getattr(config, "encoder_rnn", None) in ("bilstm", "BiLSTM", True)))


def param_specs(n_tok, dims=None):
    d = dict(DEFAULT_DIMS, **(dims or {}))
    C, E, U, O, D, V = d["C"], d["E"], d["U"], d["O"], d["D"], n_tok
    chans = [(1, 64), (64, 128), (128, 256), (256, 256), (256, C), (C, C)]
    kernels = [(3, 3, ci, co) for ci, co in chans]
    if d.get("cnn"):        # encoder_cnn == "cnn" (model/encoder.py:54-56): TF numbers the strided conv conv2d_5
        kernels.insert(5, (2, 4, C, C))
    specs = []
    for i, shp in enumerate(kernels):
        scope = "Encoder/convolutional_encoder/conv2d" + ("" if i == 0 else "_%d" % i)
        specs.append((scope + "/kernel", shp, "glorot"))
        specs.append((scope + "/bias", (shp[3],), "zeros"))
    if d.get("row_bilstm"):  # optional row encoder: a bidirectional TF LSTMCell (C/2 units per direction) over every feature-map row
        R_ = "Encoder/row_encoder/bidirectional_rnn/"
        for dr in ("fw", "bw"):
            specs.append((R_ + dr + "/lstm_cell/kernel", (C + C // 2, 4 * (C // 2)), "glorot"))
            specs.append((R_ + dr + "/lstm_cell/bias", (4 * (C // 2),), "zeros"))
    specs += [
        ("Decoder/embedding_table", (V, D), "embed"),
        ("Decoder/start_token", (D,), "embed"),
        (_A + "att_img/kernel", (C, E), "glorot"),
        (_A + "att_mechanism/W_c_0", (C, U), "glorot"),
        (_A + "att_mechanism/b_c_0", (U,), "glorot"),
        (_A + "att_mechanism/W_h_0", (C, U), "glorot"),
        (_A + "att_mechanism/b_h_0", (U,), "glorot"),
        (_A + "att_mechanism/W_o_0", (C, O), "glorot"),
        (_A + "att_mechanism/b_o_0", (O,), "glorot"),
        (_A + "rnn/lstm_cell/kernel", (D + O + U, 4 * U), "glorot"),
        (_A + "rnn/lstm_cell/bias", (4 * U,), "zeros"),
        (_A + "rnn/att_mechanism/dense/kernel", (U, E), "glorot"),
        (_A + "rnn/att_mechanism/att_beta", (E, 1), "glorot"),
        (_A + "rnn/o_W_h", (U, O), "glorot"),
        (_A + "rnn/o_W_c", (C, O), "glorot"),
        (_A + "rnn/y_W_o", (O, V), "glorot"),
    ]
    return specs


def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = 1
    for s in shape[:-2]:
        rf *= s
    return shape[-2] * rf, shape[-1] * rf


def init_params(n_tok, seed=0, dims=None):
    """OrderedDict name -> float32 ndarray, per the contract above."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name, shape, kind in param_specs(n_tok, dims):
        if kind == "zeros":
            a = np.zeros(shape, dtype=np.float64)
        elif kind == "glorot":
            fi, fo = _fans(shape)
            lim = math.sqrt(6.0 / (fi + fo))
            a = rng.uniform(-lim, lim, size=shape)
        else:
            a = rng.uniform(-1.0, 1.0, size=shape)
            a = a / np.sqrt(np.maximum((a * a).sum(axis=-1, keepdims=True), 1e-12))
        out[name] = a.astype(np.float32)
    return out


def n_params(n_tok, dims=None):
    return sum(int(np.prod(s)) for _, s, _ in param_specs(n_tok, dims))


def out_hw(H, W):
    """Encoder output grid: three SAME halvings per axis then a VALID 3x3
    (model/encoder.py:34,39,47,52,59; known-answers in SURVEY.md section 4)."""
    c = lambda n: -(-n // 2)
    return c(c(c(H))) - 2, c(c(c(W))) - 2
