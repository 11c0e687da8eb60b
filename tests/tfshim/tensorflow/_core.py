"""TEST INFRASTRUCTURE (see package docstring): tensor wrapper, dtypes, static shapes, variable store and
TF-1.12 variable-scope / default-layer-name rules of the eager TF stand-in."""
import math

import numpy as np
import torch


class DType(object):
    def __init__(self, name, tdtype):
        self.name = name
        self.torch = tdtype
        self.is_floating = tdtype.is_floating_point
        if tdtype == torch.bool:
            self.min, self.max = False, True
        elif tdtype.is_floating_point:
            self.min, self.max = float(torch.finfo(tdtype).min), float(torch.finfo(tdtype).max)   # dtype.min = lowest finite
        else:
            self.min, self.max = int(torch.iinfo(tdtype).min), int(torch.iinfo(tdtype).max)

    @property
    def base_dtype(self):
        return self

    def __repr__(self):
        return "tf." + self.name


float32 = DType("float32", torch.float32)
float64 = DType("float64", torch.float64)
int32 = DType("int32", torch.int32)
int64 = DType("int64", torch.int64)
uint8 = DType("uint8", torch.uint8)
bool_ = DType("bool", torch.bool)
_BY_TORCH = {d.torch: d for d in (float32, float64, int32, int64, uint8, bool_)}


class Dimension(object):
    def __init__(self, value):
        self.value = None if value is None else int(value)

    def __int__(self):
        return self.value

    __index__ = __int__

    def __eq__(self, other):
        return self.value == (other.value if isinstance(other, Dimension) else other)

    def __hash__(self):
        return hash(self.value)

    def __repr__(self):
        return "Dimension(%s)" % self.value


class TensorShape(object):
    def __init__(self, dims):
        self._dims = None if dims is None else [d if isinstance(d, Dimension) else Dimension(d) for d in dims]

    @property
    def ndims(self):
        return None if self._dims is None else len(self._dims)

    @property
    def dims(self):
        return self._dims

    def __len__(self):
        return len(self._dims)

    def __iter__(self):
        return iter(self._dims)

    def __getitem__(self, i):
        return TensorShape(self._dims[i]) if isinstance(i, slice) else self._dims[i]

    def as_list(self):
        return [d.value for d in self._dims]

    def __repr__(self):
        return "TensorShape(%s)" % (self.as_list(),)


def as_t(x, dtype=None, like=None):
    """-> torch tensor.  Python scalars take `dtype`, else the dtype of `like` (TF converts the constant operand
    of a binary op to the tensor operand's dtype), else float32 / int32 / bool."""
    if isinstance(x, Tensor):
        t = x.t
    elif isinstance(x, torch.Tensor):
        t = x
    elif isinstance(x, np.ndarray):
        t = torch.from_numpy(x)
    elif isinstance(x, Dimension):
        t = torch.tensor(x.value, dtype=torch.int32)
    else:
        if dtype is None and like is not None and not isinstance(x, (list, tuple)):
            td = like.dtype
            if isinstance(x, float) and not td.is_floating_point:
                td = torch.float32
            return torch.tensor(x, dtype=td)
        a = np.asarray(x)
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        elif a.dtype == np.int64:
            a = a.astype(np.int32)
        t = torch.from_numpy(a) if a.ndim else torch.tensor(a.item(), dtype=torch.from_numpy(a.reshape(1)).dtype)
    if dtype is not None:
        t = t.to(dtype.torch if isinstance(dtype, DType) else dtype)
    return t


def wrap(t):
    return t if isinstance(t, Tensor) else Tensor(t)


class Tensor(object):
    """A value.  `shape` is the static shape object the reference's code inspects (`.ndims`, `[i].value`)."""
    __array_priority__ = 100

    def __init__(self, t, name=None):
        self.t = t
        self.name = name

    @property
    def shape(self):
        return TensorShape(list(self.t.shape))

    def get_shape(self):
        return self.shape

    @property
    def dtype(self):
        return _BY_TORCH[self.t.dtype]

    def numpy(self):
        return self.t.detach().numpy()

    def __repr__(self):
        return "<shim Tensor %s %s %s>" % (self.name or "", tuple(self.t.shape), self.dtype)

    def _bin(self, other, fn, rev=False):
        o = as_t(other, like=self.t)
        a, b = (o, self.t) if rev else (self.t, o)
        return Tensor(fn(a, b))

    def __add__(self, o): return self._bin(o, torch.add)
    def __radd__(self, o): return self._bin(o, torch.add, True)
    def __sub__(self, o): return self._bin(o, torch.sub)
    def __rsub__(self, o): return self._bin(o, torch.sub, True)
    def __mul__(self, o): return self._bin(o, torch.mul)
    def __rmul__(self, o): return self._bin(o, torch.mul, True)
    def __truediv__(self, o): return self._bin(o, torch.true_divide)
    def __rtruediv__(self, o): return self._bin(o, torch.true_divide, True)
    def __floordiv__(self, o): return self._bin(o, lambda a, b: torch.div(a, b, rounding_mode="floor"))
    def __mod__(self, o): return self._bin(o, torch.remainder)
    def __neg__(self): return Tensor(-self.t)
    def __gt__(self, o): return self._bin(o, torch.gt)
    def __ge__(self, o): return self._bin(o, torch.ge)
    def __lt__(self, o): return self._bin(o, torch.lt)
    def __le__(self, o): return self._bin(o, torch.le)
    def __matmul__(self, o): return self._bin(o, torch.matmul)

    def __bool__(self):
        return bool(self.t.item())

    def __int__(self):
        return int(self.t.item())

    __index__ = __int__

    def __getitem__(self, idx):
        idx = idx if isinstance(idx, tuple) else (idx,)
        idx = tuple(int(i.t.item()) if isinstance(i, Tensor) and i.t.dim() == 0 else (i.t.long() if isinstance(i, Tensor) else i)
                    for i in idx)
        return Tensor(self.t[idx])

    __hash__ = object.__hash__


# ---------------------------------------------------------------- feeds (placeholders are bound before building)
_FEEDS = {}


def feed(values):
    _FEEDS.clear()
    _FEEDS.update(values)


def placeholder(dtype, shape, name):
    if name not in _FEEDS:
        raise KeyError("placeholder %r has no value: call tensorflow.feed({...}) before the graph code runs" % name)
    t = as_t(_FEEDS[name], dtype)
    if shape is not None:
        assert t.dim() == len(tuple(shape)), (name, tuple(t.shape), shape)
    return Tensor(t, name=name)


# ---------------------------------------------------------------- variable scopes and the variable store
AUTO_REUSE = "auto_reuse"


class VariableScope(object):
    def __init__(self, name, reuse):
        self.name = name
        self.reuse = reuse

    def reuse_variables(self):
        self.reuse = True


_STACK = [VariableScope("", None)]
_COUNTS = {}            # full scope name -> times opened (TF: var_store.variable_scopes_count)
_VARS = {}              # name -> Tensor (torch leaf with requires_grad)
_REQUESTED = {}         # name -> shape tuple, in first-request order
_CREATED = []           # names first asked for while the scope was NOT in reuse mode
_STRICT = {"missing": True}
_SLOTS = {}             # optimizer slot variables ("optimize/..." in TF) + the step counter behind beta*_power


def shim_reset(weights=None, strict=True):
    """Start a fresh "graph": clears scopes, name counters and the variable store; `weights` maps TF variable
    names to arrays.  strict: a get_variable for a name outside `weights` raises (else glorot/zeros-initialised)."""
    del _STACK[1:]
    _STACK[0].reuse = None
    _COUNTS.clear(); _VARS.clear(); _REQUESTED.clear(); del _CREATED[:]
    _STRICT["missing"] = bool(strict)
    _SLOTS.clear()
    for k, v in (weights or {}).items():
        t = as_t(v).detach().clone().to(torch.float32).requires_grad_(True)
        _VARS[k] = Tensor(t, name=k)


def shim_new_graph():
    """A fresh graph over the SAME variable store (and optimizer slots): what re-running the reference's build code
    for the next batch needs in an eager stand-in."""
    del _STACK[1:]
    _STACK[0].reuse = None
    _COUNTS.clear(); _REQUESTED.clear(); del _CREATED[:]


def shim_requested():
    return dict(_REQUESTED)


def shim_variables():
    return _VARS


def snapshot_counts():
    return dict(_COUNTS)


def restore_counts(snap):
    _COUNTS.clear()
    _COUNTS.update(snap)


def get_variable_scope():
    return _STACK[-1]


class variable_scope(object):
    """TF-1.12 rules that matter here: a string name nests under the current scope, a VariableScope object re-enters
    that absolute scope; `reuse=False` means "inherit" (variable_scope.py: "We don't allow non-inheriting scopes,
    False = None here"), `reuse=True` is inherited by sub-scopes; when a scope opened BY NAME closes, the open-counts
    of all its sub-scopes are reset (close_variable_subscopes), which is why the second `AttentionCell` pass of
    decoder.py:60 asks for "lstm_cell" / "dense" again instead of "lstm_cell_1" / "dense_1"."""

    def __init__(self, name_or_scope, default_name=None, values=None, initializer=None, reuse=None, **kw):
        self._arg = name_or_scope
        self._default = default_name
        self._reuse = None if reuse is False else reuse

    def __enter__(self):
        cur = _STACK[-1]
        self._by_object = isinstance(self._arg, VariableScope)
        if self._by_object:
            full = self._arg.name
            reuse = self._reuse if self._reuse is not None else (self._arg.reuse if self._arg.reuse is not None else cur.reuse)
            self._saved_counts = dict(_COUNTS)
        else:
            name = self._arg if self._arg is not None else unique_layer_scope(self._default, peek=True)
            full = (cur.name + "/" + name) if cur.name else name
            reuse = self._reuse if self._reuse is not None else cur.reuse
        _COUNTS[full] = _COUNTS.get(full, 0) + 1
        self._scope = VariableScope(full, reuse)
        _STACK.append(self._scope)
        return self._scope

    def __exit__(self, *exc):
        _STACK.pop()
        if self._by_object:
            _COUNTS.clear(); _COUNTS.update(self._saved_counts)
        else:
            pre = self._scope.name + "/"
            for k in [k for k in _COUNTS if k.startswith(pre)]:
                _COUNTS[k] = 0
        return False


def unique_layer_scope(base, peek=False):
    """variable_scope(None, default_name=base): `base`, `base_1`, ... -- the first whose full name was not opened yet."""
    cur = _STACK[-1]
    full = (cur.name + "/" + base) if cur.name else base
    if _COUNTS.get(full, 0) == 0:
        return base
    i = 1
    while _COUNTS.get("%s_%d" % (full, i), 0) > 0:
        i += 1
    return "%s_%d" % (base, i)


def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, **kw):
    cur = _STACK[-1]
    full = (cur.name + "/" + name) if cur.name else name
    shp = None if shape is None else tuple(int(s.value) if isinstance(s, Dimension) else int(s) for s in
                                           (shape if not isinstance(shape, TensorShape) else shape.as_list()))
    if cur.reuse is True and full not in _CREATED:
        # TF: a reusing scope may only fetch what a creating scope of THIS graph has made (loops of the eager
        # stand-in re-request names in creating scopes; that leniency is the only departure)
        raise ValueError("Variable %s does not exist, or was not created with tf.get_variable()" % full)
    if full not in _REQUESTED:
        _REQUESTED[full] = shp
        _CREATED.append(full)
    if full not in _VARS:
        if _STRICT["missing"]:
            raise KeyError("get_variable(%r): not among the supplied weights %s" % (full, sorted(_VARS)[:4]))
        if initializer is not None:
            t = as_t(initializer(list(shp), dtype or float32)).to(torch.float32)
        else:                                   # TF default for float variables: glorot_uniform_initializer
            fi, fo = _fans(shp)
            lim = math.sqrt(6.0 / (fi + fo))
            t = (torch.rand(*shp) * 2 - 1) * lim
        _VARS[full] = Tensor(t.detach().clone().requires_grad_(True), name=full)
    v = _VARS[full]
    if shp is not None and tuple(v.t.shape) != shp:
        raise ValueError("Trying to share variable %s, but specified shape %s and found shape %s" % (full, shp, tuple(v.t.shape)))
    return v
