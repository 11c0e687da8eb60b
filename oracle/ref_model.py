"""ORACLE -- test infrastructure, never shipped, never imported by the package.

CPU restatement (PyTorch-CPU, float32) of the reference's image->LaTeX hot
path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this module, and only as the checker.

PINNED TO REFERENCE CODE RUN IN THE BUILD CONTAINER (both halves):
  * encoder + positional signal: the reference's own PyTorch statement of the conv stack,
    /root/reference/model/components/seq2seq_torch.py:24-55,113-156 (tests/golden/make_ref_encoder_golden.py ->
    tests/golden/ref_encoder.npz);
  * the whole training graph and both decode graphs: the reference's UNCHANGED graph-building code
    (model/img2seq.py:42-123, encoder.py, decoder.py, components/{attention_mechanism,attention_cell,dynamic_decode,
    greedy_decoder_cell,beam_search_decoder_cell,positional}.py) executed under tests/tfshim, an eager torch-CPU stand-in
    for the ~70 tf.* symbols it touches (tests/golden/make_ref_decoder_golden.py -> tests/golden/ref_decoder.npz: train
    logits, loss, every parameter gradient, greedy ids + logits, beam 2 / 3 (diversity penalty) / 5 ids + parents, Adam loss
    trajectories, the variable names the reference's scoping produces).  tests/test_oracle.py holds this module to them.
Still RESTATED (TF-1.12 primitives whose source is not under /root/reference: tensorflow==1.12.2, requirements.txt:1,
cannot be installed here -- Python 3.10, no wheel, no network): LSTMCell arithmetic (gate order i,j,f,o, forget_bias 1.0),
the conv / pool / dense / softmax / top_k / argmax / dropout / cross-entropy ops themselves, SAME-padding geometry, the
optimizer update formulas, glorot initialisation.  The reference ships no tests or golden vectors of its own for this path
(SURVEY.md section 8c).  An independent float64 NumPy restatement (`oracle/np_micro.py`) cross-checks the arithmetic.

Reference files followed (all under /root/reference/):
  model/encoder.py:25-68                      encoder()
  model/components/positional.py:42-64        timing_signal_2d()
  model/decoder.py:41-57,75-105               decoder_train(), embeddings
  model/components/attention_mechanism.py:19-43,57-94,145-153
  model/components/attention_cell.py:51-89    cell_step()
  model/img2seq.py:68-75                      loss_fn()
  model/img2seq.py:100-123                    AdamTF, clip_by_global_norm()
  model/components/dynamic_decode.py:34-73    greedy_decode()/beam_decode() loop
  model/components/greedy_decoder_cell.py:40-66
  model/components/beam_search_decoder_cell.py:98-250,353-391
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_DIMS = dict(C=512, E=256, U=512, O=512, D=80)
ROW_ = "Encoder/row_encoder/bidirectional_rnn/"
CONV_CHANNELS = [(1, 64), (64, 128), (128, 256), (256, 256), (256, 512), (512, 512)]


def param_specs(V, dims=None):
    """Ordered (name, shape, init) list; names are the TF checkpoint names of
    SURVEY.md Appendix B.  init in {"glorot", "zeros", "embed"}."""
    d = dict(DEFAULT_DIMS, **(dims or {}))
    C, E, U, O, D = d["C"], d["E"], d["U"], d["O"], d["D"]
    chans = list(CONV_CHANNELS)
    chans[4] = (256, C)
    chans[5] = (C, C)
    kernels = [(3, 3, ci, co) for ci, co in chans]
    if d.get("cnn"):        # encoder_cnn == "cnn" (encoder.py:54-56): a (2,4) stride-2 conv before the last one
        kernels.insert(5, (2, 4, C, C))
    specs = []
    for i, shp in enumerate(kernels):
        sfx = "" if i == 0 else "_%d" % i
        specs.append(("Encoder/convolutional_encoder/conv2d%s/kernel" % sfx, shp, "glorot"))
        specs.append(("Encoder/convolutional_encoder/conv2d%s/bias" % sfx, (shp[3],), "zeros"))
    if d.get("row_bilstm"):     # optional row encoder of the product (NOT in the reference: no reference line to cite)
        for dr in ("fw", "bw"):
            specs.append((ROW_ + dr + "/lstm_cell/kernel", (C + C // 2, 4 * (C // 2)), "glorot"))
            specs.append((ROW_ + dr + "/lstm_cell/bias", (4 * (C // 2),), "zeros"))
    A = "Decoder/AttentionCell/"
    specs += [
        ("Decoder/embedding_table", (V, D), "embed"),
        ("Decoder/start_token", (D,), "embed"),
        (A + "att_img/kernel", (C, E), "glorot"),
        (A + "att_mechanism/W_c_0", (C, U), "glorot"),
        (A + "att_mechanism/b_c_0", (U,), "glorot"),
        (A + "att_mechanism/W_h_0", (C, U), "glorot"),
        (A + "att_mechanism/b_h_0", (U,), "glorot"),
        (A + "att_mechanism/W_o_0", (C, O), "glorot"),
        (A + "att_mechanism/b_o_0", (O,), "glorot"),
        (A + "rnn/lstm_cell/kernel", (D + O + U, 4 * U), "glorot"),
        (A + "rnn/lstm_cell/bias", (4 * U,), "zeros"),
        (A + "rnn/att_mechanism/dense/kernel", (U, E), "glorot"),
        (A + "rnn/att_mechanism/att_beta", (E, 1), "glorot"),
        (A + "rnn/o_W_h", (U, O), "glorot"),
        (A + "rnn/o_W_c", (C, O), "glorot"),
        (A + "rnn/y_W_o", (O, V), "glorot"),
    ]
    return specs


def _fans(shape):
    """+TF variance_scaling `_compute_fans`."""
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def init_params(V, seed=0, dims=None):
    """Seed/initialisation contract (the reference sets no seed, quirk C-13):
    NumPy Generator(PCG64(seed)); one draw per variable in `param_specs` order;
    glorot-uniform(+TF default of get_variable / tf.layers), zeros for conv and
    LSTM biases (+TF), embeddings uniform(-1,1) then L2-normalised over the last
    axis (decoder.py:98-105).  Draws are float64, stored float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name, shape, kind in param_specs(V, dims):
        if kind == "zeros":
            a = np.zeros(shape, dtype=np.float64)
        elif kind == "glorot":
            fi, fo = _fans(shape)
            lim = math.sqrt(6.0 / (fi + fo))
            a = rng.uniform(-lim, lim, size=shape)
        else:
            a = rng.uniform(-1.0, 1.0, size=shape)
            a = a / np.sqrt(np.maximum((a * a).sum(axis=-1, keepdims=True), 1e-12))
        out[name] = torch.from_numpy(a.astype(np.float32))
    return out


def _conv(x, k, b, padding):
    # HWIO -> OIHW; x is NCHW
    return F.relu(F.conv2d(x, k.permute(3, 2, 0, 1), b, stride=1, padding=padding))


def timing_signal_2d(H, W, C, dtype=torch.float32):
    """positional.py:42-64: n_ts = C//4; channels [sin_h, cos_h, sin_w, cos_w]."""
    n_ts = C // 4
    inc = math.log(1.0e4 / 1.0) / (float(n_ts) - 1)
    inv = torch.exp(torch.arange(n_ts, dtype=dtype) * -inc)
    sig = torch.zeros(H, W, C, dtype=dtype)
    ph = torch.arange(H, dtype=dtype)[:, None] * inv[None, :]
    pw = torch.arange(W, dtype=dtype)[:, None] * inv[None, :]
    sig[:, :, 0:n_ts] += torch.sin(ph)[:, None, :]
    sig[:, :, n_ts:2 * n_ts] += torch.cos(ph)[:, None, :]
    sig[:, :, 2 * n_ts:3 * n_ts] += torch.sin(pw)[None, :, :]
    sig[:, :, 3 * n_ts:4 * n_ts] += torch.cos(pw)[None, :, :]
    return sig


def _same_pad(n, k, s):
    """+TF SAME padding of one axis: (before, after)."""
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def encoder(P, img_u8, positional=True, return_all=False):
    """encoder.py:25-68.  img_u8: uint8 [B,H,W,1] -> f32 [B,H',W',C].  The "cnn" variant (encoder.py:54-56; present
    when the parameter set holds conv2d_6) drops the two late pools for a (2,4) stride-2 SAME conv without activation."""
    pre = "Encoder/convolutional_encoder/conv2d"
    x = (img_u8.to(torch.float32) - 128.0) / 128.0          # encoder.py:26-27
    x = x.permute(0, 3, 1, 2)
    acts = []
    x = _conv(x, P[pre + "/kernel"], P[pre + "/bias"], 1); acts.append(x)          # :32
    x = F.max_pool2d(x, 2, 2, ceil_mode=True)                                       # :34 SAME+TF
    x = _conv(x, P[pre + "_1/kernel"], P[pre + "_1/bias"], 1); acts.append(x)      # :37
    x = F.max_pool2d(x, 2, 2, ceil_mode=True)                                       # :39
    x = _conv(x, P[pre + "_2/kernel"], P[pre + "_2/bias"], 1); acts.append(x)      # :42
    x = _conv(x, P[pre + "_3/kernel"], P[pre + "_3/bias"], 1); acts.append(x)      # :44
    cnn = (pre + "_6/kernel") in P
    if not cnn:
        x = F.max_pool2d(x, (2, 1), (2, 1), ceil_mode=True)                         # :47
    x = _conv(x, P[pre + "_4/kernel"], P[pre + "_4/bias"], 1); acts.append(x)      # :49
    if not cnn:
        x = F.max_pool2d(x, (1, 2), (1, 2), ceil_mode=True)                         # :52
        last = "_5"
    else:
        (pt, pb), (pl, pr) = _same_pad(x.shape[2], 2, 2), _same_pad(x.shape[3], 4, 2)
        x = F.conv2d(F.pad(x, (pl, pr, pt, pb)), P[pre + "_5/kernel"].permute(3, 2, 0, 1), P[pre + "_5/bias"], stride=2)   # :56
        last = "_6"
    x = _conv(x, P[pre + last + "/kernel"], P[pre + last + "/bias"], 0); acts.append(x)   # :59 VALID
    x = x.permute(0, 2, 3, 1)
    if positional:
        x = x + timing_signal_2d(x.shape[1], x.shape[2], x.shape[3])[None]          # :66
    if (ROW_ + "fw/lstm_cell/kernel") in P:
        x = row_bilstm(P, x)
    if return_all:
        return x, [a.permute(0, 2, 3, 1) for a in acts]
    return x


def row_bilstm(P, x):
    """The product's optional row encoder (lxo_shape.encoder_rnn; north_star's "row-BiLSTM encoder").  NOT part of the reference
    (encoder.py:4 imports GRUCell / LSTMCell and never uses them), so this is a SPECIFICATION of the extension, not a restatement:
    every row of the [B, H', W', C] feature map is a sequence over W'; a forward and a backward +TF LSTMCell (C/2 units, gate order
    i, j, f, o, forget_bias 1.0, zero initial state) run over it and their outputs are concatenated [forward | backward]."""
    B, H, W, C = x.shape
    U = C // 2
    X = x.reshape(B * H, W, C)
    outs = []
    for dr, rev in (("fw", False), ("bw", True)):
        K, b = P[ROW_ + dr + "/lstm_cell/kernel"], P[ROW_ + dr + "/lstm_cell/bias"]
        c = torch.zeros(B * H, U, dtype=x.dtype)
        h = torch.zeros(B * H, U, dtype=x.dtype)
        hs = [None] * W
        for w in (range(W - 1, -1, -1) if rev else range(W)):
            z = torch.cat([X[:, w], h], dim=1) @ K + b
            i, j, f, og = z[:, :U], z[:, U:2 * U], z[:, 2 * U:3 * U], z[:, 3 * U:]
            c = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
            h = torch.sigmoid(og) * torch.tanh(c)
            hs[w] = h
        outs.append(torch.stack(hs, dim=1))
    return torch.cat(outs, dim=-1).reshape(B, H, W, C)


A_ = "Decoder/AttentionCell/"


def attention_prepare(P, enc):
    """attention_mechanism.py:19-43 and :124-153 / attention_cell.py:51-56."""
    B = enc.shape[0]
    img = enc.reshape(B, -1, enc.shape[-1])                              # :22-25
    att_img = img @ P[A_ + "att_img/kernel"]                            # :43 (no bias)
    m = img.mean(dim=1)                                                  # :148
    def s0(n):
        return torch.tanh(m @ P[A_ + "att_mechanism/W_%s_0" % n] + P[A_ + "att_mechanism/b_%s_0" % n])
    return img, att_img, (s0("c"), s0("h"), s0("o"))                     # LSTMStateTuple order c,h


def drop_mask(keep, seed, which, t, rows, width, rows_total=None, row0=0):
    """Dropout scale (0 or 1/keep) of attention_cell.py:72 (which=1, on h) / :83 (which=2, on o) for decoder
    step t.  tf.nn.dropout draws from TF's stateful Philox stream, which cannot be reproduced; the product and
    this oracle share a counter-based mask instead: splitmix64 of ((t*rows_total + row)*width + col) offset by
    (seed, which), kept iff its top 24 bits < keep * 2^24."""
    thr = max(1, int(np.float32(keep) * np.float32(16777216.0)))
    kept = hash24(seed, which, t, rows, width, rows_total, row0) < thr
    return torch.from_numpy(kept.astype(np.float32) * np.float32(np.float32(1.0) / np.float32(keep)))


def hash24(seed, which, t, rows, width, rows_total=None, row0=0):
    """24-bit counter hash shared with csrc/decoder_kernels.hip drop_scale (int64 array [rows, width])."""
    rows_total = rows if rows_total is None else rows_total
    with np.errstate(over="ignore"):
        r = (np.uint64(t) * np.uint64(rows_total) + np.arange(row0, row0 + rows, dtype=np.uint64))[:, None]
        z = r * np.uint64(width) + np.arange(width, dtype=np.uint64)[None, :]
        z = z + np.uint64(0x9E3779B97F4A7C15) * np.uint64((int(seed) & 0xFFFFFFFF) << 2 | which)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.int64)


def cell_step(P, img, att_img, emb, state, return_alpha=False, drop=None):
    """attention_cell.py:58-89 with TF-1.12 LSTMCell (+TF: gate order i,j,f,o,
    forget_bias 1.0, no peepholes).  drop = (keep, seed, t) applies the two
    tf.nn.dropout calls (:72, :83); None = keep-prob 1."""
    c, h, o = state
    U = c.shape[1]
    x = torch.cat([emb, o], dim=-1)                                      # :70
    z = torch.cat([x, h], dim=-1) @ P[A_ + "rnn/lstm_cell/kernel"] + P[A_ + "rnn/lstm_cell/bias"]
    i, j, f, og = z[:, :U], z[:, U:2 * U], z[:, 2 * U:3 * U], z[:, 3 * U:]
    c2 = torch.sigmoid(f + 1.0) * c + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.sigmoid(og) * torch.tanh(c2)                              # :71
    h_carry = h2                                                         # LSTMStateTuple keeps the un-dropped h
    if drop is not None:
        keep, seed, t = drop
        h2 = h2 * drop_mask(keep, seed, 1, t, h2.shape[0], h2.shape[1]).to(h2.dtype)   # :72
    att_h = h2 @ P[A_ + "rnn/att_mechanism/dense/kernel"]               # attention_mechanism.py:79
    e = torch.tanh(att_img + att_h[:, None, :]) @ P[A_ + "rnn/att_mechanism/att_beta"]  # :83-91
    alpha = torch.softmax(e.squeeze(-1), dim=-1)                         # :94
    ctx = (alpha[:, :, None] * img).sum(dim=1)                           # :73-74
    o2 = torch.tanh(h2 @ P[A_ + "rnn/o_W_h"] + ctx @ P[A_ + "rnn/o_W_c"])   # :82
    if drop is not None:
        o2 = o2 * drop_mask(keep, seed, 2, t, o2.shape[0], o2.shape[1]).to(o2.dtype)   # :83
    logits = o2 @ P[A_ + "rnn/y_W_o"]                                    # :84
    if return_alpha:
        return logits, (c2, h_carry, o2), alpha
    return logits, (c2, h_carry, o2)


def train_embeddings(P, formula):
    """decoder.py:75-95: [start_token, table[formula[:, :-1]]]."""
    B = formula.shape[0]
    tab = P["Decoder/embedding_table"]
    e = tab[formula.long()]
    start = P["Decoder/start_token"].reshape(1, 1, -1).expand(B, 1, -1)
    return torch.cat([start, e[:, :-1, :]], dim=1)


def decoder_train(P, enc, formula, return_alpha=False, dropout=None):
    """decoder.py:50-57: T = formula.shape[1] steps, padded steps computed.
    dropout = (keep, seed) or None (config.dropout = 1, the shipped value)."""
    img, att_img, state = attention_prepare(P, enc)
    emb = train_embeddings(P, formula)
    outs, alphas = [], []
    for t in range(formula.shape[1]):
        dr = None if dropout is None else (dropout[0], dropout[1], t)
        if return_alpha:
            lg, state, a = cell_step(P, img, att_img, emb[:, t], state, True, drop=dr)
            alphas.append(a)
        else:
            lg, state = cell_step(P, img, att_img, emb[:, t], state, drop=dr)
        outs.append(lg)
    logits = torch.stack(outs, dim=1)
    if return_alpha:
        return logits, torch.stack(alphas, dim=1)
    return logits


def loss_fn(logits, formula, lengths):
    """img2seq.py:68-75: token-mean CE over unmasked positions; ce_words = sum
    of masked CE; n_words = sum(lengths)."""
    B, T, V = logits.shape
    ce = F.cross_entropy(logits.reshape(B * T, V), formula.reshape(-1).long(), reduction="none").reshape(B, T)
    mask = (torch.arange(T)[None, :] < lengths.long()[:, None])
    ce_words = (ce * mask).sum()
    n_words = lengths.long().sum()
    return ce_words / mask.sum(), ce_words, n_words


def forward_loss(P, img_u8, formula, lengths, positional=True, dropout=None):
    enc = encoder(P, img_u8, positional)
    logits = decoder_train(P, enc, formula, dropout=dropout)
    return loss_fn(logits, formula, lengths)


def train_grads(P, img_u8, formula, lengths, positional=True, dropout=None):
    """loss and d(loss)/d(param) by autograd (the reference gets BPTT from TF
    autodiff, img2seq.py:119-123)."""
    Q = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in P.items())
    loss, ce_words, n_words = forward_loss(Q, img_u8, formula, lengths, positional, dropout)
    grads = torch.autograd.grad(loss, list(Q.values()), allow_unused=True)
    G = OrderedDict((k, (g if g is not None else torch.zeros_like(P[k]))) for k, g in zip(Q.keys(), grads))
    return loss.detach(), G, ce_words.detach(), n_words


def clip_by_global_norm(G, clip):
    """+TF tf.clip_by_global_norm: g * clip / max(||g||, clip)."""
    gn = torch.sqrt(sum((g.double() ** 2).sum() for g in G.values())).float()
    scale = clip / torch.maximum(gn, torch.tensor(float(clip)))
    return OrderedDict((k, g * scale) for k, g in G.items()), gn


class AdamTF(object):
    """+TF tf.train.AdamOptimizer defaults (img2seq.py:101): beta1 .9, beta2
    .999, eps 1e-8; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)
    (epsilon OUTSIDE the bias correction, unlike torch.optim.Adam)."""

    def __init__(self, P, beta1=0.9, beta2=0.999, eps=1e-8):
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.t = 0
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in P.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in P.items())

    def step(self, P, G, lr):
        self.t += 1
        lr_t = float(lr) * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        lr_t = np.float32(lr_t)
        for k in P:
            g = G[k]
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            P[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))


def train_step(P, opt, img_u8, formula, lengths, lr, clip=-1.0, positional=True, dropout=None):
    loss, G, _, _ = train_grads(P, img_u8, formula, lengths, positional, dropout)
    if clip > 0:
        G, _ = clip_by_global_norm(G, clip)
    opt.step(P, G, lr)
    return float(loss)


@torch.no_grad()
def greedy_decode(P, img_u8, id_end, max_iter=151, positional=True, return_logits=False, return_alpha=False):
    """dynamic_decode.py:34-73 + greedy_decoder_cell.py:40-66.  Loop while not
    all finished; after the step at `time`, finished |= time >= max_iter, so at
    most max_iter+1 steps.  Finished rows keep stepping.  argmax ties -> lowest
    index (+TF).  Returns int32 [B, T']."""
    enc = encoder(P, img_u8, positional)
    img, att_img, state = attention_prepare(P, enc)
    B = img.shape[0]
    tab = P["Decoder/embedding_table"]
    emb = P["Decoder/start_token"].reshape(1, -1).expand(B, -1)
    finished = torch.zeros(B, dtype=torch.bool)
    ids_all, logits_all, alpha_all = [], [], []
    time = 0
    while not bool(finished.all()):
        logits, state, alpha = cell_step(P, img, att_img, emb, state, return_alpha=True)
        alpha_all.append(alpha)                                  # attention_mechanism.py:96-105 py_func tap
        ids = torch.argmax(logits, dim=-1)
        emb = tab[ids]
        finished = finished | (ids == id_end)
        ids_all.append(ids.to(torch.int32))
        logits_all.append(logits)
        if time >= max_iter:
            finished = torch.ones_like(finished)
        time += 1
    ids = torch.stack(ids_all, dim=1)
    if return_alpha:
        return ids, torch.stack(alpha_all, dim=1)                # [B, T', R]
    if return_logits:
        return ids, torch.stack(logits_all, dim=1)
    return ids


@torch.no_grad()
def add_div_penalty(lp, div_gamma, div_prob, div_seed, time):
    """beam_search_decoder_cell.py:258-287: lp[b,k,v] += log(div_gamma) * rank * bernoulli(div_prob), rank = the
    position of v in the descending sort of lp[b,k,:] (+TF top_k: ties -> lower index first).  The Bernoulli
    draws use the shared counter hash (stream 3) instead of TF's stateful RNG."""
    if div_gamma is None or div_prob is None or div_gamma == 1.0 or div_prob == 0.0:      # :270-273
        return lp
    B, k, V = lp.shape
    order = torch.argsort(lp.reshape(B * k, V), dim=1, descending=True, stable=True)         # :276
    rank = torch.empty_like(order)
    rank.scatter_(1, order, torch.arange(V).expand(B * k, V).contiguous())                # :278-280 invert_permutation
    thr = 16777216 if div_prob >= 1.0 else int(np.float32(div_prob) * np.float32(16777216.0))
    apply = torch.from_numpy((hash24(div_seed, 3, time, B * k, V) < thr).astype(np.float32))   # :284
    pen = np.float32(math.log(np.float32(div_gamma))) * rank.to(torch.float32) * apply     # :282-285
    return lp + pen.reshape(B, k, V)


def beam_decode(P, img_u8, id_end, beam_size, max_iter=151, positional=True, div_gamma=1.0, div_prob=0.0, div_seed=0, return_alpha=False):
    """beam_search_decoder_cell.py:98-250 (`finalize` never follows parents, quirk C-1, so hypothesis
    i is ids[:, t, i] at every t).  Returns int32 [B, T', k] and the parents; with return_alpha also the attention weights
    [B, T', k, R] of the merged batch x beam rows as each step ran (what the py_func tap of attention_mechanism.py:96-121 is handed)."""
    enc = encoder(P, img_u8, positional)
    img, att_img, (c, h, o) = attention_prepare(P, enc)
    B, k = img.shape[0], beam_size
    V = P["Decoder/embedding_table"].shape[0]
    tab = P["Decoder/embedding_table"]
    tile = lambda t: t[:, None].expand(B, k, *t.shape[1:]).reshape(B * k, *t.shape[1:])
    img_t, att_t = tile(img), tile(att_img)                # attention_mechanism.py:59-65
    state = (tile(c), tile(h), tile(o))                    # :98-109
    emb = P["Decoder/start_token"].reshape(1, -1).expand(B * k, -1)
    log_probs = torch.zeros(B, k)
    finished = torch.zeros(B, k, dtype=torch.bool)
    fmin = torch.finfo(torch.float32).min
    ids_all, par_all, alpha_all = [], [], []
    time = 0
    while not bool(finished.all()):
        logits, new_state, alpha = cell_step(P, img_t, att_t, emb, state, return_alpha=True)           # :137
        alpha_all.append(alpha.reshape(B, k, -1))
        step_lp = F.log_softmax(logits.reshape(B, k, V), dim=-1)             # :146
        one_hot = torch.full((V,), fmin); one_hot[id_end] = 0.0              # :353-367
        fin = finished.to(torch.float32)[:, :, None]
        step_lp = (1.0 - fin) * step_lp + fin * one_hot
        lp = log_probs[:, :, None] + step_lp                                 # :150
        lp = add_div_penalty(lp, div_gamma, div_prob, div_seed, time)        # :151
        flat = lp.reshape(B, k * V) if time > 0 else lp[:, 0]                # :156-160
        new_probs, idx = _top_k_lowest_index(flat, k)                        # :161
        new_ids = idx % V                                                    # :164
        parents = idx // V                                                   # :165
        emb = tab[new_ids.reshape(-1)]
        gat = lambda t: t.reshape(B, k, -1).gather(1, parents[:, :, None].expand(B, k, t.shape[-1])).reshape(B * k, -1)
        finished = finished.gather(1, parents) | (new_ids == id_end)         # :171-174
        state = tuple(gat(s) for s in new_state)                             # :176-178
        log_probs = new_probs
        ids_all.append(new_ids.to(torch.int32))
        par_all.append(parents.to(torch.int32))
        if time >= max_iter:
            finished = torch.ones_like(finished)
        time += 1
    if return_alpha:
        return torch.stack(ids_all, dim=1), torch.stack(par_all, dim=1), torch.stack(alpha_all, dim=1)
    return torch.stack(ids_all, dim=1), torch.stack(par_all, dim=1)


def _top_k_lowest_index(x, k):
    """+TF top_k: descending values, ties -> lower index first."""
    vals, idxs = [], []
    y = x.clone()
    rows = torch.arange(x.shape[0])
    for _ in range(k):
        v = y.max(dim=1).values
        is_max = (y == v[:, None])
        i = torch.argmax(is_max.to(torch.int8), dim=1)     # first index of the max
        vals.append(v); idxs.append(i)
        y[rows, i] = -float("inf")
    return torch.stack(vals, dim=1), torch.stack(idxs, dim=1)


def out_hw(H, W):
    """Encoder output geometry (SURVEY.md section 4 known-answers)."""
    c = lambda n: -(-n // 2)
    return c(c(c(H))) - 2, c(c(c(W))) - 2


class SimpleOptTF(object):
    """+TF the non-Adam branches of add_optimizer (img2seq.py:102-107) with TF-1.12 defaults:
    GradientDescent; Adagrad (initial_accumulator_value 0.1); RMSProp (decay .9, momentum 0, eps 1e-10,
    rms slot initialised to ones)."""

    def __init__(self, P, method):
        self.method = method.lower()
        init = {"sgd": 0.0, "adagrad": 0.1, "rmsprop": 1.0}[self.method]
        self.slot = OrderedDict((k, torch.full_like(v, init)) for k, v in P.items())

    def step(self, P, G, lr):
        lr = np.float32(lr)
        for k in P:
            g = G[k]
            if self.method == "sgd":
                P[k].sub_(lr * g)
            elif self.method == "adagrad":
                self.slot[k].add_(g * g)
                P[k].sub_(lr * g / self.slot[k].sqrt())
            else:
                self.slot[k].mul_(0.9).add_(0.1 * g * g)
                P[k].sub_(lr * g / (self.slot[k] + 1e-10).sqrt())
