"""TEST INFRASTRUCTURE: tf.train optimizers of the eager TF stand-in (RESTATED TF-1.12 update rules -- these are
primitives whose source is not under /root/reference; running the reference's add_optimizer with them pins the
WIRING (which optimizer, clip_by_global_norm placement), not the formulas)."""
import math

import torch

from ._core import as_t, wrap, shim_variables, _SLOTS


class _Optimizer(object):
    def __init__(self, learning_rate, **kw):
        self._lr = learning_rate
        self._slots = _SLOTS            # slots live with the variable store, like TF's slot variables live in the graph

    def compute_gradients(self, loss, var_list=None):
        vs = list(shim_variables().values()) if var_list is None else list(var_list)
        gs = torch.autograd.grad(as_t(loss), [v.t for v in vs], allow_unused=True)
        return [(None if g is None else wrap(g), v) for g, v in zip(gs, vs)]

    def minimize(self, loss, **kw):
        return self.apply_gradients(self.compute_gradients(loss))

    def apply_gradients(self, grads_and_vars, **kw):
        self._t = self._slots["__step__"] = self._slots.get("__step__", 0) + 1
        lr = float(as_t(self._lr))
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is not None:
                    self._apply(v, as_t(g), lr)
        return None


class GradientDescentOptimizer(_Optimizer):
    def _apply(self, v, g, lr):
        v.t.sub_(lr * g)


class AdamOptimizer(_Optimizer):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **kw):
        _Optimizer.__init__(self, learning_rate)
        self.b1, self.b2, self.eps = beta1, beta2, epsilon

    def _apply(self, v, g, lr):
        m, s = self._slots.setdefault(v.name, (torch.zeros_like(v.t), torch.zeros_like(v.t)))
        lr_t = lr * math.sqrt(1.0 - self.b2 ** self._t) / (1.0 - self.b1 ** self._t)
        m.mul_(self.b1).add_(g, alpha=1 - self.b1)
        s.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
        v.t.sub_(lr_t * m / (s.sqrt() + self.eps))


class AdagradOptimizer(_Optimizer):
    def _apply(self, v, g, lr):
        acc = self._slots.setdefault(v.name, torch.full_like(v.t, 0.1))
        acc.add_(g * g)
        v.t.sub_(lr * g / acc.sqrt())


class RMSPropOptimizer(_Optimizer):
    def _apply(self, v, g, lr):
        rms = self._slots.setdefault(v.name, torch.ones_like(v.t))
        rms.mul_(0.9).add_(0.1 * g * g)
        v.t.sub_(lr * g / (rms + 1e-10).sqrt())


class Saver(object):
    def __init__(self, *a, **k):
        pass
