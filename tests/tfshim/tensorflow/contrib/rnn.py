"""TEST INFRASTRUCTURE: tf.contrib.rnn of the eager TF stand-in.  LSTMCell is a RESTATED TF-1.12 primitive
(rnn_cell_impl.LSTMCell, defaults: no peepholes, no projection, forget_bias 1.0, tanh, state_is_tuple)."""
import collections

import torch

from .._core import as_t, wrap, variable_scope, get_variable, get_variable_scope, unique_layer_scope, float32, VariableScope

LSTMStateTuple = collections.namedtuple("LSTMStateTuple", ("c", "h"))


class RNNCell(object):
    """Base class: the reference's AttentionCell derives from it and overrides __call__ itself."""

    @property
    def state_size(self):
        raise NotImplementedError

    @property
    def output_size(self):
        raise NotImplementedError


class LSTMCell(RNNCell):
    def __init__(self, num_units, forget_bias=1.0, state_is_tuple=True, reuse=None, name=None, **kw):
        assert state_is_tuple
        self._num_units = int(num_units)
        self._forget_bias = float(forget_bias)
        self._reuse = reuse
        self._state_size = LSTMStateTuple(self._num_units, self._num_units)
        self._scope = None              # Layer._set_scope: fixed at the first call, under the scope current then
        self._name = name

    @property
    def state_size(self):
        return self._state_size

    @property
    def output_size(self):
        return self._num_units

    def __call__(self, inputs, state, scope=None):
        x = as_t(inputs)
        c_prev, h_prev = state
        built = self._scope is not None
        if not built:
            cur = get_variable_scope()
            name = self._name or unique_layer_scope("lstm_cell")
            self._scope = VariableScope((cur.name + "/" + name) if cur.name else name, None)
        with variable_scope(self._scope, reuse=(True if (built or self._reuse) else None)):
            k = get_variable("kernel", shape=[int(x.shape[1]) + self._num_units, 4 * self._num_units], dtype=float32)
            b = get_variable("bias", shape=[4 * self._num_units], dtype=float32,
                             initializer=lambda shape, dtype, partition_info=None: torch.zeros(*shape))
        z = torch.cat([x, as_t(h_prev)], dim=1) @ as_t(k) + as_t(b)
        i, j, f, o = torch.split(z, self._num_units, dim=1)         # i = input gate, j = new input, f = forget, o = output
        c = torch.sigmoid(f + self._forget_bias) * as_t(c_prev) + torch.sigmoid(i) * torch.tanh(j)
        m = torch.sigmoid(o) * torch.tanh(c)
        return wrap(m), LSTMStateTuple(wrap(c), wrap(m))


class GRUCell(RNNCell):
    """Imported by the reference (decoder.py:5, encoder.py:3), never instantiated."""
    def __init__(self, *a, **k):
        raise NotImplementedError
