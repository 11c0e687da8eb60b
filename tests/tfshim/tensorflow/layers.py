"""TEST INFRASTRUCTURE: tf.layers of the eager TF stand-in (functional layers; TF-1.12 naming: an explicit `name`
is the scope, otherwise the per-scope unique default name "conv2d", "conv2d_1", ... / "dense")."""
import torch
import torch.nn.functional as F

from ._core import as_t, wrap, variable_scope, get_variable, unique_layer_scope, float32


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def _scope(name, base, reuse):
    return variable_scope(name if name is not None else unique_layer_scope(base), reuse=reuse)


def conv2d(inputs, filters, kernel_size, strides=(1, 1), padding="valid", data_format="channels_last",
           dilation_rate=(1, 1), activation=None, use_bias=True, kernel_initializer=None, name=None, reuse=None, **kw):
    x = as_t(inputs)                      # NHWC
    kh, kw_ = _pair(kernel_size)
    sh, sw = _pair(strides)
    cin = int(x.shape[3])
    with _scope(name, "conv2d", reuse):
        k = get_variable("kernel", shape=[kh, kw_, cin, int(filters)], dtype=float32, initializer=kernel_initializer)
        b = get_variable("bias", shape=[int(filters)], dtype=float32,
                         initializer=lambda shape, dtype, partition_info=None: torch.zeros(*shape)) if use_bias else None
    xc = x.permute(0, 3, 1, 2)
    if padding.upper() == "SAME":
        (pt, pb), (pl, pr) = _same_pad(xc.shape[2], kh, sh), _same_pad(xc.shape[3], kw_, sw)
        xc = F.pad(xc, (pl, pr, pt, pb))
    y = F.conv2d(xc, as_t(k).permute(3, 2, 0, 1), None if b is None else as_t(b), stride=(sh, sw))
    y = wrap(y.permute(0, 2, 3, 1))
    return activation(y) if activation is not None else y


def max_pooling2d(inputs, pool_size, strides, padding="valid", data_format="channels_last", name=None):
    x = as_t(inputs).permute(0, 3, 1, 2)
    kh, kw_ = _pair(pool_size)
    sh, sw = _pair(strides)
    if padding.upper() == "SAME":         # padding never wins a max
        (pt, pb), (pl, pr) = _same_pad(x.shape[2], kh, sh), _same_pad(x.shape[3], kw_, sw)
        x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return wrap(F.max_pool2d(x, (kh, kw_), (sh, sw)).permute(0, 2, 3, 1))


def dense(inputs, units, activation=None, use_bias=True, kernel_initializer=None, name=None, reuse=None, **kw):
    x = as_t(inputs)
    with _scope(name, "dense", reuse):
        k = get_variable("kernel", shape=[int(x.shape[-1]), int(units)], dtype=float32, initializer=kernel_initializer)
        b = get_variable("bias", shape=[int(units)], dtype=float32,
                         initializer=lambda shape, dtype, partition_info=None: torch.zeros(*shape)) if use_bias else None
    y = x @ as_t(k)                       # rank > 2: tensordot over the last axis
    if b is not None:
        y = y + as_t(b)
    y = wrap(y)
    return activation(y) if activation is not None else y
