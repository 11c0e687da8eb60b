"""TEST INFRASTRUCTURE: tf.nn of the eager TF stand-in (restated TF-1.12 primitives, see package docstring)."""
import torch
import torch.nn.functional as F

from ._core import as_t, wrap, variable_scope, int32


def relu(x, name=None):
    return wrap(torch.relu(as_t(x)))


def tanh(x, name=None):
    return wrap(torch.tanh(as_t(x)))


def sigmoid(x, name=None):
    return wrap(torch.sigmoid(as_t(x)))


def softmax(logits, axis=-1, name=None):
    return wrap(torch.softmax(as_t(logits), dim=axis))


def log_softmax(logits, axis=-1, name=None):
    return wrap(torch.log_softmax(as_t(logits), dim=axis))


def top_k(input, k=1, sorted=True, name=None):      # noqa: A002
    """Descending values; equal values keep their index order (TF: "if two elements are equal, the lower-index
    element appears first")."""
    v, i = torch.sort(as_t(input), dim=-1, descending=True, stable=True)
    k = int(k)
    return wrap(v[..., :k].contiguous()), wrap(i[..., :k].to(torch.int32).contiguous())


def embedding_lookup(params, ids, name=None, **kw):
    return wrap(as_t(params)[as_t(ids).long()])


def dropout(x, keep_prob, noise_shape=None, seed=None, name=None):
    """TF 1.12 nn_ops.dropout: x / keep_prob * floor(keep_prob + uniform[0,1))."""
    t = as_t(x)
    keep = as_t(keep_prob, like=t).to(t.dtype)
    return wrap(t / keep * torch.floor(keep + torch.rand_like(t)))


def l2_normalize(x, axis=None, epsilon=1e-12, name=None, dim=None):
    axis = dim if axis is None else axis
    t = as_t(x)
    sq = (t * t).sum(dim=axis, keepdim=True)
    return wrap(t * torch.rsqrt(torch.clamp(sq, min=epsilon)))


def sparse_softmax_cross_entropy_with_logits(_sentinel=None, labels=None, logits=None, name=None):
    lg, lb = as_t(logits), as_t(labels).long()
    lp = torch.log_softmax(lg, dim=-1)
    return wrap(-lp.gather(-1, lb.unsqueeze(-1)).squeeze(-1))


def dynamic_rnn(cell, inputs, sequence_length=None, initial_state=None, dtype=None, time_major=False, scope=None, **kw):
    """rnn.py dynamic_rnn: opens variable_scope("rnn") and calls cell(input_t, state) for every time step (no
    `sequence_length` here, so padded steps are computed).  Batch-major outputs [B, T, ...]."""
    assert sequence_length is None and not time_major
    from . import _core
    x = as_t(inputs)
    state = initial_state
    outs = []
    with variable_scope(scope or "rnn"):
        snap = _core.snapshot_counts()
        for t in range(x.shape[1]):
            _core.restore_counts(snap)
            out, state = cell(wrap(x[:, t]), state)
            outs.append(as_t(out))
    return wrap(torch.stack(outs, dim=1)), state
