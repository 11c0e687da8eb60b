"""ORACLE (second opinion) -- test infrastructure, never shipped.

Independent float64 NumPy restatement of the same hot path as
`oracle/ref_model.py`, written with explicit shifted-window sums instead of
library conv/pool/softmax, so that the torch restatement is not a single point
of failure.  Small shapes only.  Same reference
citations as ref_model.py; PARITY UNPINNED for the same reason.
"""
import math

import numpy as np

A_ = "Decoder/AttentionCell/"


def conv3x3(x, k, b, same):
    """x [B,H,W,Ci] f64, k [3,3,Ci,Co] HWIO; SAME = zero pad 1 (+TF)."""
    B, H, W, Ci = x.shape
    if same:
        xp = np.zeros((B, H + 2, W + 2, Ci)); xp[:, 1:-1, 1:-1] = x
        Ho, Wo = H, W
    else:
        xp = x; Ho, Wo = H - 2, W - 2
    out = np.zeros((B, Ho, Wo, k.shape[3]))
    for kh in range(3):
        for kw in range(3):
            out += xp[:, kh:kh + Ho, kw:kw + Wo, :] @ k[kh, kw]
    return np.maximum(out + b, 0.0)


def maxpool(x, ph, pw):
    """SAME max-pool, window=stride=(ph,pw): pad bottom/right, ignore padding."""
    B, H, W, C = x.shape
    Ho, Wo = -(-H // ph), -(-W // pw)
    xp = np.full((B, Ho * ph, Wo * pw, C), -np.inf); xp[:, :H, :W] = x
    return xp.reshape(B, Ho, ph, Wo, pw, C).max(axis=(2, 4))


def encoder(P, img_u8, positional=True):
    pre = "Encoder/convolutional_encoder/conv2d"
    g = lambda n: np.asarray(P[n], dtype=np.float64)
    x = (img_u8.astype(np.float64) - 128.0) / 128.0
    x = maxpool(conv3x3(x, g(pre + "/kernel"), g(pre + "/bias"), True), 2, 2)
    x = maxpool(conv3x3(x, g(pre + "_1/kernel"), g(pre + "_1/bias"), True), 2, 2)
    x = conv3x3(x, g(pre + "_2/kernel"), g(pre + "_2/bias"), True)
    x = maxpool(conv3x3(x, g(pre + "_3/kernel"), g(pre + "_3/bias"), True), 2, 1)
    x = maxpool(conv3x3(x, g(pre + "_4/kernel"), g(pre + "_4/bias"), True), 1, 2)
    x = conv3x3(x, g(pre + "_5/kernel"), g(pre + "_5/bias"), False)
    if positional:
        B, H, W, C = x.shape
        n = C // 4
        inc = math.log(1.0e4) / (n - 1)
        inv = np.exp(-inc * np.arange(n))
        for hh in range(H):
            x[:, hh, :, 0:n] += np.sin(hh * inv)
            x[:, hh, :, n:2 * n] += np.cos(hh * inv)
        for ww in range(W):
            x[:, :, ww, 2 * n:3 * n] += np.sin(ww * inv)
            x[:, :, ww, 3 * n:4 * n] += np.cos(ww * inv)
    return x


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def decoder_train(P, enc, formula):
    g = lambda n: np.asarray(P[n], dtype=np.float64)
    B = enc.shape[0]
    img = enc.reshape(B, -1, enc.shape[-1])
    att_img = img @ g(A_ + "att_img/kernel")
    m = img.mean(axis=1)
    c = np.tanh(m @ g(A_ + "att_mechanism/W_c_0") + g(A_ + "att_mechanism/b_c_0"))
    h = np.tanh(m @ g(A_ + "att_mechanism/W_h_0") + g(A_ + "att_mechanism/b_h_0"))
    o = np.tanh(m @ g(A_ + "att_mechanism/W_o_0") + g(A_ + "att_mechanism/b_o_0"))
    tab, start = g("Decoder/embedding_table"), g("Decoder/start_token")
    K, bk = g(A_ + "rnn/lstm_cell/kernel"), g(A_ + "rnn/lstm_cell/bias")
    U = c.shape[1]
    T = formula.shape[1]
    logits = np.zeros((B, T, tab.shape[0]))
    alphas = np.zeros((B, T, img.shape[1]))
    for t in range(T):
        emb = np.tile(start, (B, 1)) if t == 0 else tab[formula[:, t - 1]]
        z = np.concatenate([emb, o, h], axis=1) @ K + bk
        i, j, f, og = z[:, :U], z[:, U:2 * U], z[:, 2 * U:3 * U], z[:, 3 * U:]
        c = _sig(f + 1.0) * c + _sig(i) * np.tanh(j)
        h = _sig(og) * np.tanh(c)
        att_h = h @ g(A_ + "rnn/att_mechanism/dense/kernel")
        e = (np.tanh(att_img + att_h[:, None, :]) @ g(A_ + "rnn/att_mechanism/att_beta"))[:, :, 0]
        e = e - e.max(axis=1, keepdims=True)
        a = np.exp(e); a /= a.sum(axis=1, keepdims=True)
        ctx = (a[:, :, None] * img).sum(axis=1)
        o = np.tanh(h @ g(A_ + "rnn/o_W_h") + ctx @ g(A_ + "rnn/o_W_c"))
        logits[:, t] = o @ g(A_ + "rnn/y_W_o")
        alphas[:, t] = a
    return logits, alphas


def loss_fn(logits, formula, lengths):
    B, T, V = logits.shape
    z = logits - logits.max(axis=-1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=-1))
    ce = lse - np.take_along_axis(z, formula[:, :, None].astype(np.int64), axis=2)[:, :, 0]
    mask = np.arange(T)[None, :] < lengths[:, None]
    return (ce * mask).sum() / mask.sum(), (ce * mask).sum(), int(lengths.sum())
