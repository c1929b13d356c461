"""Eager, torch-backed stand-in for the parts of Theano the reference's hot path uses (TEST INFRASTRUCTURE).

Theano is an un-vendored, un-pinned dependency of the reference and cannot be installed here.  This
module restates the *published semantics* of the few dozen `theano.*` / `theano.tensor.*` calls made by
/root/reference/{model.py, sampleRNN/lib/ops.py, sampleRNN/lib/__init__.py, sampleRNN/models/conditional/
three_tier.py}, so that those files can be executed UNMODIFIED (see loader.py) and their outputs committed as golden
vectors (tests/golden/make_ref_golden.py).  Every "symbolic" variable is a `Var` holding a concrete torch tensor
(float64 when theano.config.floatX = 'float64'); ops run immediately and record their inputs (`owner.inputs`), so the
reference's own graph walkers (lib.search / lib.get_params) work, and torch autograd yields the gradients of the
reference's own graph code.

Semantics restated (Theano 0.9/1.0 documentation):
  * arithmetic / comparison operators broadcast like NumPy; `==` is identity (tensor.eq is the elementwise test);
  * `x.std()` / `tensor.std` use ddof = 0; `nnet.softmax` is row-wise on a matrix;
  * `scan(fn, sequences, outputs_info, non_sequences)`: fn(*sequence slices, *recurrent outputs, *non_sequences),
    outputs stacked along a new leading axis, a single output is returned bare;
  * `nnet.categorical_crossentropy(p, idx)` = -log p[i, idx[i]];
  * `nnet.neighbours.images2neibs(x[1,B,1,L], (1,n), (1,1), 'valid')` = all length-n windows, image-major;
  * `ifelse(c, a, b)` / `switch(c, a, b)`; `set_subtensor(x[idx], y)` returns a modified copy.
"""
from __future__ import annotations

import collections
import types

import numpy
import torch

_DTYPES = {'float64': torch.float64, 'float32': torch.float32, 'float16': torch.float16, 'int64': torch.int64,
           'int32': torch.int32, 'int16': torch.int16, 'int8': torch.int8, 'uint8': torch.uint8, 'bool': torch.bool}
_NAMES = {v: k for k, v in _DTYPES.items()}


class _Config:
    floatX = 'float64'


config = _Config()


def _fx():
    return _DTYPES[config.floatX]


class Apply:
    """Owner record of a computed Var (only `.inputs` is used, by lib.search)."""

    def __init__(self, inputs):
        self.inputs = list(inputs)


def _t(x):
    """Anything -> torch tensor (python / numpy scalars become 0-d tensors, which torch promotes weakly)."""
    if isinstance(x, Var):
        return x.data
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, numpy.ndarray):
        return torch.from_numpy(numpy.ascontiguousarray(x))
    if isinstance(x, (bool, numpy.bool_)):
        return torch.tensor(bool(x))
    if isinstance(x, (int, numpy.integer)):
        return torch.tensor(int(x), dtype=torch.int64)
    if isinstance(x, (float, numpy.floating)):
        return torch.tensor(float(x), dtype=torch.float64 if isinstance(x, (float, numpy.float64)) else _fx())
    if isinstance(x, (list, tuple)):
        return torch.stack([_t(e) for e in x])
    raise TypeError(f"cannot convert {type(x)} to a tensor")


def _promote(a, b):
    if a.dtype == b.dtype:
        return a, b
    if a.dim() == 0 and b.dim() > 0 and (b.dtype.is_floating_point or not a.dtype.is_floating_point):
        return a.to(b.dtype), b
    if b.dim() == 0 and a.dim() > 0 and (a.dtype.is_floating_point or not b.dtype.is_floating_point):
        return a, b.to(a.dtype)
    dt = torch.promote_types(a.dtype, b.dtype)
    if a.dim() == 0 and b.dim() == 0 and dt.is_floating_point:
        dt = torch.float64
    return a.to(dt), b.to(dt)


def _mk(data, *inputs, name=None):
    return Var(data, [i for i in inputs if isinstance(i, Var)], name)


def _bin(fn):
    def op(a, b):
        ta, tb = _promote(_t(a), _t(b))
        return _mk(fn(ta, tb), a, b)
    return op


def _idx(i):
    if isinstance(i, Var):
        d = i.data
        return d.long() if not d.dtype.is_floating_point and d.dtype != torch.bool else d
    if isinstance(i, tuple):
        return tuple(_idx(e) for e in i)
    if isinstance(i, numpy.ndarray):
        return torch.from_numpy(i).long()
    if isinstance(i, numpy.integer):
        return int(i)
    return i


def _getitem(d, idx):
    """d[idx] with NumPy semantics for negative-step slices (torch has none): x[::-1] etc."""
    items = idx if isinstance(idx, tuple) else (idx,)
    if not any(isinstance(i, slice) and i.step is not None and i.step < 0 for i in items):
        return d[idx]
    assert all(isinstance(i, (slice, int)) for i in items), "negative steps only with basic indexing"
    pos, flips, out_dim = [], [], 0
    for dim, i in enumerate(items):
        if isinstance(i, slice) and i.step is not None and i.step < 0:
            pos.append(slice(None))
            flips.append((out_dim, torch.tensor(list(range(d.shape[dim])[i]), dtype=torch.long)))
        else:
            pos.append(i)
        if isinstance(i, slice):
            out_dim += 1
    out = d[tuple(pos)]
    for od, sel in flips:
        out = out.index_select(od, sel)
    return out


class Var:
    """A 'symbolic' variable evaluated eagerly."""
    __array_ufunc__ = None  # numpy scalars (lib.floatX(1.0) - x) defer to the reflected operators below

    def __init__(self, data, inputs=(), name=None):
        self.data = data
        self.owner = Apply(inputs) if inputs else None
        self.name = name
        self._sub = None

    # --- static-ish properties
    @property
    def shape(self):
        return tuple(int(s) for s in self.data.shape)

    @property
    def ndim(self):
        return self.data.dim()

    @property
    def dtype(self):
        return _NAMES[self.data.dtype]

    @property
    def T(self):
        return _mk(self.data.t() if self.ndim == 2 else self.data.permute(*reversed(range(self.ndim))), self)

    def __len__(self):
        return self.data.shape[0]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __repr__(self):
        return f"Var({self.name}, shape={self.shape}, {self.dtype})"

    # --- shared-variable API
    def get_value(self, borrow=False):
        return self.data.detach().cpu().numpy().copy()

    def set_value(self, v, borrow=False):
        with torch.no_grad():
            self.data.copy_(_t(v).to(self.data.dtype).reshape(self.data.shape))

    def eval(self, *a, **k):
        return self.get_value()

    # --- operators (no in-place forms: `x += y` rebinds, as on a symbolic variable)
    __add__ = _bin(torch.add)
    __radd__ = lambda s, o: Var.__add__(o if isinstance(o, Var) else _mk(_t(o)), s)
    __sub__ = _bin(torch.sub)
    __rsub__ = lambda s, o: _bin(torch.sub)(o, s)
    __mul__ = _bin(torch.mul)
    __rmul__ = lambda s, o: _bin(torch.mul)(o, s)
    __truediv__ = _bin(torch.true_divide)
    __rtruediv__ = lambda s, o: _bin(torch.true_divide)(o, s)
    __floordiv__ = _bin(lambda a, b: torch.div(a, b, rounding_mode='floor'))
    __mod__ = _bin(torch.remainder)
    __pow__ = _bin(torch.pow)
    __rpow__ = lambda s, o: _bin(torch.pow)(o, s)
    __lt__ = _bin(torch.lt)
    __le__ = _bin(torch.le)
    __gt__ = _bin(torch.gt)
    __ge__ = _bin(torch.ge)

    def __neg__(self):
        return _mk(-self.data, self)

    def __abs__(self):
        return _mk(self.data.abs(), self)

    def __bool__(self):
        if self.data.numel() != 1:
            raise ValueError("truth value of a non-scalar Var")
        return bool(self.data.item() != 0)

    def __int__(self):
        return int(self.data.item())

    def __float__(self):
        return float(self.data.item())

    __index__ = __int__
    __hash__ = object.__hash__

    def __getitem__(self, idx):
        parents = [self] + [i for i in (idx if isinstance(idx, tuple) else (idx,)) if isinstance(i, Var)]
        out = Var(_getitem(self.data, _idx(idx)), parents)
        out._sub = (self, idx)
        return out

    # --- methods
    def dimshuffle(self, *pattern):
        if len(pattern) == 1 and isinstance(pattern[0], (list, tuple)):
            pattern = tuple(pattern[0])
        d = self.data
        perm = [p for p in pattern if p != 'x']
        dropped = [i for i in range(d.dim()) if i not in perm]
        for i in dropped:
            assert d.shape[i] == 1, "dimshuffle can only drop broadcastable axes"
        d = d.permute(*(perm + dropped)).reshape([d.shape[p] for p in perm])
        for pos, p in enumerate(pattern):
            if p == 'x':
                d = d.unsqueeze(pos)
        return _mk(d, self)

    def reshape(self, shape, ndim=None):
        if isinstance(shape, Var):
            shape = [int(s) for s in shape.data.tolist()]
        shape = [int(s) for s in shape]
        return _mk(self.data.reshape(shape), self)

    def flatten(self, ndim=1):
        d = self.data
        if ndim == 1:
            return _mk(d.reshape(-1), self)
        return _mk(d.reshape(list(d.shape[:ndim - 1]) + [-1]), self)

    def astype(self, dtype):
        return _mk(self.data.to(_DTYPES[str(dtype)]), self)

    def dot(self, other):
        return dot(self, other)

    def sum(self, axis=None, keepdims=False, dtype=None):
        return sum(self, axis=axis, keepdims=keepdims)

    def mean(self, axis=None, keepdims=False):
        return mean(self, axis=axis, keepdims=keepdims)

    def std(self, axis=None, keepdims=False):
        return std(self, axis=axis, keepdims=keepdims)

    def var(self, axis=None, keepdims=False):
        d = self.data
        return _mk(d.var(unbiased=False) if axis is None else d.var(dim=axis, unbiased=False, keepdim=keepdims), self)

    def max(self, axis=None, keepdims=False):
        return max(self, axis=axis, keepdims=keepdims)

    def min(self, axis=None, keepdims=False):
        d = self.data
        return _mk(d.min() if axis is None else d.amin(dim=axis, keepdim=keepdims), self)

    def argmax(self, axis=None):
        return argmax(self, axis=axis)

    def norm(self, L, axis=None, keepdims=False):
        d = self.data
        if L == 2:
            return _mk((d * d).sum() .sqrt() if axis is None else (d * d).sum(dim=axis, keepdim=keepdims).sqrt(), self)
        if L == 1:
            return _mk(d.abs().sum() if axis is None else d.abs().sum(dim=axis, keepdim=keepdims), self)
        raise NotImplementedError(L)

    def repeat(self, repeats, axis=None):
        return repeat(self, repeats, axis)

    def copy(self, name=None):
        return _mk(self.data.clone(), self, name=name)

    def clip(self, a, b):
        return clip(self, a, b)

    def nonzero(self):
        return tuple(_mk(t, self) for t in self.data.nonzero(as_tuple=True))


# ------------------------------------------------------------------ theano.tensor functions
def as_tensor_variable(x, name=None, ndim=None):
    return x if isinstance(x, Var) else Var(_t(x), name=name)


def constant(x, name=None, dtype=None):
    t = _t(x)
    return Var(t.to(_DTYPES[str(dtype)]) if dtype else t, name=name)


def _un(fn):
    def op(x):
        t = _t(x)
        if not t.dtype.is_floating_point and fn not in (torch.abs, torch.sign):
            t = t.to(_fx())
        return _mk(fn(t), x)
    return op


exp, log, sqrt, tanh, cos, sin, floor, ceil, sgn = (_un(f) for f in (
    torch.exp, torch.log, torch.sqrt, torch.tanh, torch.cos, torch.sin, torch.floor, torch.ceil, torch.sign))
abs_ = _un(torch.abs)
log1p = _un(torch.log1p)


def sqr(x):
    return x * x


def dot(a, b):
    ta, tb = _t(a), _t(b)
    dt = torch.promote_types(ta.dtype, tb.dtype)
    if not dt.is_floating_point:
        dt = _fx()
    ta, tb = ta.to(dt), tb.to(dt)
    if ta.dim() == 1 and tb.dim() == 1:
        out = torch.dot(ta, tb)
    elif tb.dim() == 1:
        out = torch.mv(ta.reshape(-1, ta.shape[-1]), tb).reshape(ta.shape[:-1])
    else:
        out = torch.tensordot(ta, tb, dims=([ta.dim() - 1], [0 if tb.dim() < 2 else tb.dim() - 2])) if tb.dim() > 2 \
            else torch.matmul(ta, tb)
    return _mk(out, a, b)


def _axes(axis):
    return tuple(axis) if isinstance(axis, (list, tuple)) else axis


def sum(x, axis=None, keepdims=False, dtype=None):  # noqa: A001 (mirrors theano.tensor.sum)
    d = _t(x)
    return _mk(d.sum() if axis is None else d.sum(dim=_axes(axis), keepdim=keepdims), x)


def mean(x, axis=None, keepdims=False):
    d = _t(x)
    return _mk(d.mean() if axis is None else d.mean(dim=_axes(axis), keepdim=keepdims), x)


def std(x, axis=None, keepdims=False):
    d = _t(x)
    return _mk(d.std(unbiased=False) if axis is None else d.std(dim=_axes(axis), unbiased=False, keepdim=keepdims), x)


def max(x, axis=None, keepdims=False):  # noqa: A001
    d = _t(x)
    return _mk(d.max() if axis is None else d.amax(dim=_axes(axis), keepdim=keepdims), x)


def min(x, axis=None, keepdims=False):  # noqa: A001
    d = _t(x)
    return _mk(d.min() if axis is None else d.amin(dim=_axes(axis), keepdim=keepdims), x)


def argmax(x, axis=None, keepdims=False):
    d = _t(x)
    # numpy / Theano argmax: first (lowest) index among equal maxima
    if axis is None:
        d, axis = d.reshape(-1), 0
    m = d.amax(dim=axis, keepdim=True)
    n = d.shape[axis]
    ar = torch.arange(n).reshape([-1 if i == (axis % d.dim()) else 1 for i in range(d.dim())])
    first = torch.where(d == m, ar, torch.full_like(ar, n)).amin(dim=axis, keepdim=keepdims)
    return _mk(first, x)


def eq(a, b):
    return _bin(torch.eq)(a, b)


def neq(a, b):
    return _bin(torch.ne)(a, b)


def gt(a, b):
    return _bin(torch.gt)(a, b)


def lt(a, b):
    return _bin(torch.lt)(a, b)


def maximum(a, b):
    return _bin(torch.maximum)(a, b)


def minimum(a, b):
    return _bin(torch.minimum)(a, b)


def clip(x, a, b):
    return minimum(maximum(x, a), b)


def switch(cond, a, b):
    c = _t(cond)
    ta, tb = _promote(_t(a), _t(b))
    return _mk(torch.where(c != 0, ta, tb), cond, a, b)


def cast(x, dtype):
    return _mk(_t(x).to(_DTYPES[str(dtype)]), x)


def _shape(shape):
    if isinstance(shape, (Var, torch.Tensor)):
        return [int(s) for s in _t(shape).tolist()]
    if isinstance(shape, (int, numpy.integer)):
        return [int(shape)]
    return [int(s) for s in shape]


def zeros(shape, dtype=None):
    return Var(torch.zeros(_shape(shape), dtype=_DTYPES[str(dtype)] if dtype else _fx()))


def ones(shape, dtype=None):
    return Var(torch.ones(_shape(shape), dtype=_DTYPES[str(dtype)] if dtype else _fx()))


def zeros_like(x, dtype=None):
    return Var(torch.zeros_like(_t(x)))


def ones_like(x, dtype=None):
    return Var(torch.ones_like(_t(x)))


def arange(start, stop=None, step=1, dtype=None):
    if stop is None:
        start, stop = 0, start
    out = torch.arange(int(start), int(stop), int(step))
    return Var(out.to(_DTYPES[str(dtype)]) if dtype else out)


def alloc(value, *shape):
    v = _t(value)
    return _mk(v.expand(_shape(shape)).clone(), value)


def concatenate(tensors, axis=0):
    ts = [_t(t) for t in tensors]
    dt = ts[0].dtype
    for t in ts[1:]:
        dt = torch.promote_types(dt, t.dtype)
    return _mk(torch.cat([t.to(dt) for t in ts], dim=axis), *tensors)


def stack(tensors, axis=0, *more):
    if more or isinstance(tensors, Var):  # old signature stack(a, b, ...)
        tensors = [tensors, axis] + list(more)
        axis = 0
    return _mk(torch.stack([_t(t) for t in tensors], dim=axis), *tensors)


def repeat(x, repeats, axis=None):
    d = _t(x)
    r = int(repeats) if not isinstance(repeats, (Var, torch.Tensor)) or _t(repeats).dim() == 0 else _t(repeats)
    return _mk(torch.repeat_interleave(d.reshape(-1) if axis is None else d, r, dim=0 if axis is None else axis), x)


def tile(x, reps, ndim=None):
    return _mk(_t(x).repeat(*_shape(reps)), x)


def shape_padleft(x, n_ones=1):
    d = _t(x)
    return _mk(d.reshape([1] * n_ones + list(d.shape)), x)


def shape_padright(x, n_ones=1):
    d = _t(x)
    return _mk(d.reshape(list(d.shape) + [1] * n_ones), x)


def shape_padaxis(x, axis):
    return _mk(_t(x).unsqueeze(axis), x)


def patternbroadcast(x, pattern):
    return x


def unbroadcast(x, *axes):
    return x


def addbroadcast(x, *axes):
    return x


def set_subtensor(sub, y, inplace=False, tolerate_inplace_aliasing=False):
    base, idx = sub._sub
    out = base.data.clone()
    out[_idx(idx)] = _t(y).to(out.dtype)
    return _mk(out, base, y)


def inc_subtensor(sub, y, inplace=False, tolerate_inplace_aliasing=False):
    base, idx = sub._sub
    out = base.data.clone()
    out[_idx(idx)] = out[_idx(idx)] + _t(y).to(out.dtype)
    return _mk(out, base, y)


def grad(cost, wrt, **kw):
    single = isinstance(wrt, Var)
    ws = [wrt] if single else list(wrt)
    gs = torch.autograd.grad(cost.data, [w.data for w in ws], retain_graph=True, allow_unused=True)
    out = [Var(torch.zeros_like(w.data) if g is None else g) for g, w in zip(gs, ws)]
    return out[0] if single else out


def _placeholder(ndim, dtype=None):
    def make(name=None, dtype=dtype):
        raise NotImplementedError(
            "symbolic placeholders are not supported by the eager shim: call the graph-building function with "
            "concrete Vars instead of compiling it with theano.function")
    return make


scalar, vector, matrix, tensor3, tensor4 = (_placeholder(n) for n in range(5))
iscalar, ivector, imatrix, itensor3 = (_placeholder(n, 'int32') for n in range(4))
lscalar, lvector, lmatrix = (_placeholder(n, 'int64') for n in range(3))
fmatrix, dmatrix = _placeholder(2, 'float32'), _placeholder(2, 'float64')
TensorVariable = Var
TensorConstant = Var


# ------------------------------------------------------------------ theano.tensor.nnet
def sigmoid(x):
    return _mk(torch.sigmoid(_t(x)), x)


def softmax(x):
    d = _t(x)
    assert d.dim() == 2, "theano.tensor.nnet.softmax expects a matrix"
    return _mk(torch.softmax(d, dim=1), x)


def relu(x, alpha=0):
    d = _t(x)
    return _mk(torch.where(d > 0, d, alpha * d), x)


def softplus(x):
    return _mk(torch.nn.functional.softplus(_t(x)), x)


def categorical_crossentropy(coding_dist, true_dist):
    p, t = _t(coding_dist), _t(true_dist)
    if t.dim() == p.dim():
        return _mk(-(t * torch.log(p)).sum(dim=1), coding_dist, true_dist)
    rows = torch.arange(p.shape[0])
    return _mk(-torch.log(p[rows, t.long()]), coding_dist, true_dist)


def binary_crossentropy(o, t):
    to, tt = _t(o), _t(t)
    return _mk(-(tt * torch.log(to) + (1 - tt) * torch.log(1 - to)), o, t)


def images2neibs(ten4, neib_shape, neib_step=None, mode='valid'):
    d = _t(ten4)
    nr, nc = int(neib_shape[0]), int(neib_shape[1])
    sr, sc = (nr, nc) if neib_step is None else (int(neib_step[0]), int(neib_step[1]))
    assert mode == 'valid'
    # [b, c, R, C] -> patches ordered (b, c, patch row, patch col), each flattened row-major
    p = d.unfold(2, nr, sr).unfold(3, nc, sc)  # [b, c, pr, pc, nr, nc]
    return _mk(p.reshape(-1, nr * nc), ten4)


# ------------------------------------------------------------------ theano.* top level
def shared(value, name=None, strict=False, allow_downcast=None, borrow=False, broadcastable=None, **kw):
    t = _t(value).clone()
    if t.dtype.is_floating_point:
        t = t.to(_fx()).requires_grad_()
    return Var(t, name=name)


class OrderedUpdates(collections.OrderedDict):
    def __add__(self, other):
        return list(self.items()) + list(other.items() if isinstance(other, dict) else other)

    def __radd__(self, other):
        return list(other.items() if isinstance(other, dict) else other) + list(self.items())


def scan(fn, sequences=None, outputs_info=None, non_sequences=None, n_steps=None, go_backwards=False, name=None,
         truncate_gradient=-1, strict=False, **kw):
    def aslist(x):
        return [] if x is None else (list(x) if isinstance(x, (list, tuple)) else [x])
    seqs = [s if isinstance(s, Var) else as_tensor_variable(s) for s in aslist(sequences)]
    infos = aslist(outputs_info)
    infos = [i['initial'] if isinstance(i, dict) else i for i in infos]
    nonseq = aslist(non_sequences)
    n = int(n_steps) if n_steps is not None else builtins_min(len(s) for s in seqs)
    order = range(n - 1, -1, -1) if go_backwards else range(n)
    prev = [None if i is None else (i if isinstance(i, Var) else as_tensor_variable(i)) for i in infos]
    collected = None
    single = False
    for i in order:
        args = [s[i] for s in seqs] + [p for p in prev if p is not None] + nonseq
        res = fn(*args)
        if isinstance(res, tuple) and len(res) == 2 and isinstance(res[1], dict):
            res = res[0]
        single = not isinstance(res, (list, tuple))
        res = [res] if single else list(res)
        if collected is None:
            collected = [[] for _ in res]
            if not infos:
                prev = [None] * len(res)
        for j, r in enumerate(res):
            r = r if isinstance(r, Var) else as_tensor_variable(r)
            collected[j].append(r)
            if j < len(prev) and prev[j] is not None:
                prev[j] = r
    outs = [stack(c, 0) for c in collected]
    return (outs[0] if single else outs), OrderedUpdates()


def builtins_min(it):
    import builtins
    return builtins.min(it)


def ifelse(condition, then_branch, else_branch, name=None):
    """Lazy conditional.  The chosen branch's value is returned, but the node keeps BOTH branches as inputs: in a
    symbolic Theano graph both are reachable from the result whatever the condition's run-time value, which is what
    lib.get_params walks (three_tier.py:595-600: the learned h0 is a parameter of the cost for reset = 0 too)."""
    c = bool(_t(condition).item() != 0)
    chosen = then_branch if c else else_branch
    if isinstance(chosen, (list, tuple)):
        return chosen
    return _mk(_t(chosen), condition, then_branch, else_branch)


def function(*a, **k):
    raise NotImplementedError("theano.function: the eager shim evaluates graph-building code directly")


class _RandomStreams:
    """theano.sandbox.rng_mrg.MRG_RandomStreams.  The MRG31k3p stream itself is not reproduced: only the
    deterministic paths of the reference are pinned, so drawing is an error rather than a silent substitute."""

    def __init__(self, seed=None, *a, **k):
        self.seed = seed

    def _no(self, *a, **k):
        raise NotImplementedError("stochastic Theano ops are outside the pinned (deterministic) paths")

    multinomial = normal = uniform = binomial = _no


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def build_modules():
    """The module tree to install into sys.modules (see loader.install)."""
    g = globals()
    tensor_names = [
        'as_tensor_variable', 'constant', 'exp', 'log', 'log1p', 'sqrt', 'tanh', 'cos', 'sin', 'floor', 'ceil', 'sgn',
        'abs_', 'sqr', 'dot', 'sum', 'mean', 'std', 'max', 'min', 'argmax', 'eq', 'neq', 'gt', 'lt', 'maximum',
        'minimum', 'clip', 'switch', 'cast', 'zeros', 'ones', 'zeros_like', 'ones_like', 'arange', 'alloc',
        'concatenate', 'stack', 'repeat', 'tile', 'shape_padleft', 'shape_padright', 'shape_padaxis',
        'patternbroadcast', 'unbroadcast', 'addbroadcast', 'set_subtensor', 'inc_subtensor', 'grad', 'scalar',
        'vector', 'matrix', 'tensor3', 'tensor4', 'iscalar', 'ivector', 'imatrix', 'itensor3', 'lscalar', 'lvector',
        'lmatrix', 'fmatrix', 'dmatrix', 'TensorVariable', 'TensorConstant', 'Apply']
    neighbours = _module('theano.tensor.nnet.neighbours', images2neibs=images2neibs)
    bn = _module('theano.tensor.nnet.bn')
    nnet = _module('theano.tensor.nnet', sigmoid=sigmoid, softmax=softmax, relu=relu, softplus=softplus,
                   categorical_crossentropy=categorical_crossentropy, binary_crossentropy=binary_crossentropy,
                   neighbours=neighbours, bn=bn)
    extra_ops = _module('theano.tensor.extra_ops', repeat=repeat)
    tensor = _module('theano.tensor', **{n: g[n] for n in tensor_names}, nnet=nnet, extra_ops=extra_ops)
    rng_mrg = _module('theano.sandbox.rng_mrg', MRG_RandomStreams=_RandomStreams)
    sandbox = _module('theano.sandbox', rng_mrg=rng_mrg)
    ifelse_mod = _module('theano.ifelse', ifelse=ifelse)
    gof = _module('theano.gof', Apply=Apply, Variable=Var)
    compile_mod = _module('theano.compile', SharedVariable=Var)
    theano = _module('theano', config=config, tensor=tensor, shared=shared, scan=scan, function=function, grad=grad,
                     ifelse=ifelse_mod, sandbox=sandbox, gof=gof, compile=compile_mod, Variable=Var,
                     __version__='eager-shim', OrderedUpdates=OrderedUpdates)
    theano.updates = _module('theano.updates', OrderedUpdates=OrderedUpdates)
    return {m.__name__: m for m in (theano, tensor, nnet, neighbours, bn, extra_ops, sandbox, rng_mrg, ifelse_mod, gof,
                                    compile_mod, theano.updates)}
