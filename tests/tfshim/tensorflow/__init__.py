"""TEST INFRASTRUCTURE -- an eager, torch-CPU (float32) stand-in for the ~70 `tf.*` symbols the reference's
graph-building code touches (tensorflow==1.12.2 cannot be installed here: Python 3.10, no wheel, no network).

Purpose: run the reference's OWN, UNCHANGED graph code -- /root/reference/model/{encoder,decoder,img2seq}.py and
model/components/{attention_mechanism,attention_cell,dynamic_decode,greedy_decoder_cell,beam_search_decoder_cell,
positional}.py -- in the build container so that its outputs can be committed as golden vectors
(tests/golden/make_ref_decoder_golden.py) and the oracle's decoder half is pinned to reference code actually run.
Never imported by the package, bench.py or the oracle.

Execution model: every op executes immediately on torch tensors ("the graph is run while it is built"), so
placeholders take their value from `feed()` BEFORE the graph code is called and `tf.while_loop` is a Python loop.
Variables live in a name -> tensor store filled by `shim_reset(weights)`; names follow TF-1.12's variable-scope /
layer-naming rules (re-entered scopes, `reuse`, per-scope uniquification of default layer names that is reset when
the enclosing named scope closes), so the names the reference asks for are themselves a check of SURVEY Appendix B.

What stays RESTATED here (TF primitives whose source is not under /root/reference; semantics from TF 1.12):
  * contrib.rnn.LSTMCell arithmetic (gate order i,j,f,o; forget_bias 1.0; state tuple (c, h))
  * layers.conv2d / max_pooling2d SAME/VALID geometry, layers.dense, nn.softmax / log_softmax / top_k (ties ->
    lower index) / argmax (first index) / dropout (x / keep * floor(keep + u)) / sparse softmax cross-entropy,
    sequence_mask / boolean_mask, glorot-uniform default initialiser
  * train.*Optimizer update formulas and clip_by_global_norm
"""
import builtins as _b
import contextlib
import math

builtins_range = _b.range

import numpy as np
import torch
import torch.nn.functional as F

from . import _core
from ._core import (Tensor, DType, TensorShape, Dimension, float32, float64, int32, int64, uint8, as_t, wrap,
                    shim_reset, shim_new_graph, shim_requested, shim_variables, feed, variable_scope, get_variable_scope, get_variable,
                    VariableScope, unique_layer_scope, AUTO_REUSE)
from ._core import bool_ as bool            # noqa: A001  (tf.bool)
from . import nn, layers, contrib, train, summary, python   # noqa: F401,E402

__version__ = "1.12.2-shim"
Shape = TensorShape


class GraphKeys(object):
    UPDATE_OPS = "update_ops"
    GLOBAL_VARIABLES = "variables"
    TRAINABLE_VARIABLES = "trainable_variables"


def get_collection(key, scope=None):
    if key in (GraphKeys.GLOBAL_VARIABLES, GraphKeys.TRAINABLE_VARIABLES):
        return list(shim_variables().values())
    return []


def trainable_variables():
    return list(shim_variables().values())


def reset_default_graph():
    pass


def placeholder(dtype, shape=None, name=None):
    return _core.placeholder(dtype, shape, name)


def Variable(initial_value, trainable=True, name=None, dtype=None):
    return wrap(as_t(initial_value, dtype))


def constant(value, dtype=None, shape=None, name=None):
    t = as_t(value, dtype)
    if shape is not None:
        t = t.expand(*shape) if t.dim() == 0 else t.reshape(*shape)
    return wrap(t)


def convert_to_tensor(value, dtype=None, name=None):
    return wrap(as_t(value, dtype))


def identity(x, name=None):
    return wrap(as_t(x))


def cast(x, dtype, name=None):
    return wrap(as_t(x).to(dtype.torch))


def to_float(x, name=None):
    return cast(x, float32)


def shape(x, name=None):
    """Run-time shape.  Eager: a list of Python ints (so that H*W, batch*beam ... are plain arithmetic)."""
    return [int(d) for d in as_t(x).shape]


def _ints(seq):
    return [int(as_t(s)) if not isinstance(s, int) else s for s in seq]


def reshape(tensor, shape, name=None):      # noqa: A002
    return wrap(as_t(tensor).reshape(*_ints(shape)))


def expand_dims(input, axis=None, name=None, dim=None):     # noqa: A002
    axis = dim if axis is None else axis
    return wrap(as_t(input).unsqueeze(axis))


def tile(input, multiples, name=None):      # noqa: A002
    return wrap(as_t(input).repeat(*_ints(multiples)))


def transpose(a, perm=None, name=None):
    t = as_t(a)
    perm = list(builtins_range(t.dim()))[::-1] if perm is None else list(perm)
    return wrap(t.permute(*perm))


def zeros(shape, dtype=float32, name=None):     # noqa: A002
    return wrap(torch.zeros(*_ints(shape), dtype=dtype.torch))


def ones(shape, dtype=float32, name=None):      # noqa: A002
    return wrap(torch.ones(*_ints(shape), dtype=dtype.torch))


def range(start, limit=None, delta=1, dtype=None, name=None):   # noqa: A001
    if limit is None:
        start, limit = 0, start
    vals = [start, limit, delta]
    is_f = any(isinstance(v, float) or (isinstance(v, Tensor) and v.t.is_floating_point()) for v in vals)
    td = dtype.torch if dtype is not None else (torch.float32 if is_f else torch.int32)
    s, l, d = [v.t.item() if isinstance(v, Tensor) else v for v in vals]
    return wrap(torch.arange(s, l, d, dtype=td))


def concat(values, axis, name=None):
    return wrap(torch.cat([as_t(v) for v in values], dim=axis))


def matmul(a, b, name=None):
    return wrap(as_t(a) @ as_t(b))


def _unary(fn):
    def op(x, name=None):
        return wrap(fn(as_t(x, float32) if not isinstance(x, Tensor) else x.t))
    return op


tanh = _unary(torch.tanh)
sigmoid = _unary(torch.sigmoid)
sin = _unary(torch.sin)
cos = _unary(torch.cos)
exp = _unary(torch.exp)
log = _unary(torch.log)
sqrt = _unary(torch.sqrt)
logical_not = _unary(torch.logical_not)


def _axis_kw(axis, keepdims):
    return {} if axis is None else {"dim": axis, "keepdim": _b.bool(keepdims)}


def reduce_sum(x, axis=None, keepdims=False, name=None):
    return wrap(as_t(x).sum(**_axis_kw(axis, keepdims)))


def reduce_mean(x, axis=None, keepdims=False, name=None):
    return wrap(as_t(x).mean(**_axis_kw(axis, keepdims)))


def reduce_all(x, axis=None, keepdims=False, name=None):
    t = as_t(x)
    return wrap(t.all() if axis is None else t.all(dim=axis, keepdim=keepdims))


def _binary(fn):
    def op(a, b, name=None):
        ta = as_t(a)
        tb = as_t(b, like=ta)
        return wrap(fn(ta, tb))
    return op


logical_or = _binary(torch.logical_or)
logical_and = _binary(torch.logical_and)
equal = _binary(torch.eq)
greater = _binary(torch.gt)
greater_equal = _binary(torch.ge)
less = _binary(torch.lt)
less_equal = _binary(torch.le)
maximum = _binary(torch.maximum)
minimum = _binary(torch.minimum)


def argmax(input, axis=None, name=None, output_type=int64):    # noqa: A002
    # TF: index of the first maximum; torch.argmax documents the same
    return wrap(torch.argmax(as_t(input), dim=axis).to(output_type.torch))


def pad(tensor, paddings, mode="CONSTANT", name=None, constant_values=0):
    flat = []
    for lo, hi in reversed([tuple(p) for p in paddings]):
        flat += [int(lo), int(hi)]
    return wrap(F.pad(as_t(tensor), flat, value=constant_values))


def one_hot(indices, depth, on_value=1.0, off_value=0.0, axis=None, dtype=float32, name=None):
    idx = as_t(indices).long()
    out = torch.full(tuple(idx.shape) + (int(depth),), float(off_value), dtype=dtype.torch)
    out.scatter_(-1, idx.unsqueeze(-1), float(on_value))
    return wrap(out)


def gather(params, indices, axis=0, name=None):
    assert axis == 0
    return wrap(as_t(params)[as_t(indices).long()])


def reverse(tensor, axis, name=None):
    return wrap(torch.flip(as_t(tensor), dims=list(axis)))


def invert_permutation(x, name=None):
    t = as_t(x).long()
    out = torch.empty_like(t)
    out[t] = torch.arange(t.numel(), dtype=t.dtype)
    return wrap(out.to(as_t(x).dtype))


def map_fn(fn, elems, dtype=None, back_prop=True, name=None, **kw):
    t = as_t(elems)
    return wrap(torch.stack([as_t(fn(wrap(t[i]))) for i in builtins_range(t.shape[0])], dim=0))




def random_uniform(shape, minval=0, maxval=None, dtype=float32, seed=None, name=None):      # noqa: A002
    maxval = 1 if maxval is None else maxval
    return wrap(torch.rand(*_ints(shape), dtype=dtype.torch) * (maxval - minval) + minval)


def sequence_mask(lengths, maxlen=None, dtype=None, name=None):
    l = as_t(lengths).long()
    maxlen = int(l.max()) if maxlen is None else int(as_t(maxlen))
    m = torch.arange(maxlen)[None, :] < l[..., None]
    return wrap(m if dtype is None else m.to(dtype.torch))


def boolean_mask(tensor, mask, name=None, axis=None):
    t, m = as_t(tensor), as_t(mask)
    if tuple(t.shape[:m.dim()]) != tuple(m.shape):
        raise ValueError("boolean_mask: shapes %s and %s are incompatible" % (tuple(t.shape), tuple(m.shape)))
    return wrap(t[m])


def clip_by_global_norm(t_list, clip_norm, use_norm=None, name=None):
    ts = [as_t(g) for g in t_list]
    gn = torch.sqrt(sum((g * g).sum() for g in ts))
    scale = clip_norm / torch.maximum(gn, torch.tensor(float(clip_norm)))
    return [wrap(g * scale) for g in ts], wrap(gn)


def cond(pred, true_fn=None, false_fn=None, name=None, **kw):
    p = pred.t.item() if isinstance(pred, Tensor) else pred
    return true_fn() if p else false_fn()


def while_loop(cond, body, loop_vars, back_prop=True, **kw):     # noqa: A002
    """Eager loop.  TF traces `body` once; so that default layer names inside it do not grow with the trip count
    the per-scope name counters are rewound before every iteration."""
    vars_ = list(loop_vars)
    snap = _core.snapshot_counts()
    while True:
        c = cond(*vars_)
        if not (c.t.item() if isinstance(c, Tensor) else c):
            break
        _core.restore_counts(snap)
        ctx = contextlib.nullcontext() if back_prop else torch.no_grad()
        with ctx:
            vars_ = list(body(*vars_))
    return vars_


class TensorArray(object):
    def __init__(self, dtype, size=None, dynamic_size=None, **kw):
        self.dtype = dtype
        self._items = {}

    def write(self, index, value, name=None):
        i = int(as_t(index))
        self._items[i] = as_t(value)
        return self

    def read(self, index, name=None):
        return wrap(self._items[int(as_t(index))])

    def stack(self, name=None):
        n = len(self._items)
        assert sorted(self._items) == list(builtins_range(n)), sorted(self._items)
        return wrap(torch.stack([self._items[i] for i in builtins_range(n)], dim=0))

    def size(self):
        return len(self._items)


def py_func(func, inp, Tout, stateful=True, name=None):
    """The reference's visualisation tap (attention_mechanism.py:96-121): run the host callback on NumPy copies."""
    res = func(*[as_t(x).detach().numpy() for x in inp])
    res = res if isinstance(res, (list, tuple)) else [res]
    return [wrap(as_t(r)) for r in res]


@contextlib.contextmanager
def control_dependencies(control_inputs):
    yield


@contextlib.contextmanager
def name_scope(name, default_name=None, values=None):
    yield name


def gradients(ys, xs, grad_ys=None, **kw):
    ys = ys if isinstance(ys, (list, tuple)) else [ys]
    total = sum(as_t(y).sum() for y in ys)
    gs = torch.autograd.grad(total, [as_t(x) for x in xs], allow_unused=True, retain_graph=True)
    return [None if g is None else wrap(g) for g in gs]


def global_variables_initializer():
    return None
