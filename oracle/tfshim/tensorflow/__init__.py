"""Stub `tensorflow`: the calls the reference's rec.coding package makes, in numpy (see ../README.md).  Test infrastructure."""
import builtins
import types

import numpy as np

from oracle import oracle as _O

float32, float64, int32, int64 = np.float32, np.float64, np.int32, np.int64
bool = np.bool_          # noqa: A001  (tf.bool)
__version__ = "2.1.0-irec-numpy-stub"


class Shape(list):
    """TensorShape stand-in: `[n] + t.shape`, `t.shape[i]`, `len(t.shape)`, iteration, comparison."""

    def __add__(self, other):
        return Shape(list(self) + list(other))

    def __radd__(self, other):
        return Shape(list(other) + list(self))

    def as_list(self):
        return list(self)


def _np(x, like=None):
    if isinstance(x, Tensor):
        return x._a
    if isinstance(x, (list, tuple)) and any(isinstance(e, Tensor) for e in x):
        return np.stack([_np(e) for e in x])
    a = np.asarray(x)
    if like is not None and not isinstance(x, np.ndarray):        # python / numpy scalars take the tensor's dtype, as in TF
        if a.dtype.kind in "fiub" and like.dtype.kind == "f":
            a = a.astype(like.dtype)
        elif a.dtype.kind in "iub" and like.dtype.kind in "iu":
            a = a.astype(like.dtype)
    elif like is None and a.dtype == np.float64 and not isinstance(x, np.ndarray):
        a = a.astype(np.float32)                                  # tf.constant(1.5) is float32
    return a


class Tensor:
    __array_priority__ = 100

    def __init__(self, a):
        self._a = np.asarray(a)

    # ---- introspection ----
    @property
    def shape(self):
        return Shape(self._a.shape)

    @property
    def dtype(self):
        return self._a.dtype.type

    def numpy(self):
        return self._a.copy() if self._a.ndim else self._a[()]

    def __len__(self):
        return self._a.shape[0]

    def __iter__(self):
        return (Tensor(v) for v in self._a)

    def __index__(self):
        return int(self._a)

    def __int__(self):
        return int(self._a)

    def __float__(self):
        return float(self._a)

    def __bool__(self):
        return builtins.bool(self._a)

    def __repr__(self):
        return f"tfstub.Tensor({self._a!r})"

    def __format__(self, spec):
        return format(self._a[()] if self._a.ndim == 0 else str(self._a), spec)

    def __hash__(self):
        return id(self)

    # ---- indexing ----
    def __getitem__(self, key):
        def conv(k):
            return int(k._a) if isinstance(k, Tensor) and k._a.ndim == 0 else (k._a if isinstance(k, Tensor) else k)
        key = tuple(conv(k) for k in key) if isinstance(key, tuple) else conv(key)
        return Tensor(self._a[key])

    # ---- arithmetic (the tensor's dtype wins over python scalars) ----
    def _bin(self, other, fn, rev=False):
        o = _np(other, like=self._a)
        with np.errstate(all="ignore"):
            return Tensor(fn(o, self._a) if rev else fn(self._a, o))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.true_divide)
    def __rtruediv__(self, o): return self._bin(o, np.true_divide, True)
    def __floordiv__(self, o): return self._bin(o, np.floor_divide)
    def __mod__(self, o): return self._bin(o, np.mod)
    def __neg__(self): return Tensor(-self._a)
    def __lt__(self, o): return self._bin(o, np.less)
    def __le__(self, o): return self._bin(o, np.less_equal)
    def __gt__(self, o): return self._bin(o, np.greater)
    def __ge__(self, o): return self._bin(o, np.greater_equal)
    def __eq__(self, o): return self._bin(o, np.equal)
    def __ne__(self, o): return self._bin(o, np.not_equal)


class Module:
    def __init__(self, name=None, **kwargs):
        self._name = name

    @property
    def name(self):
        return self._name


class _Layer(Module):
    pass


keras = types.SimpleNamespace(layers=types.SimpleNamespace(Layer=_Layer))
TensorShape = Shape


def Variable(initial_value, **kwargs):            # only touched with extrapolate_auxiliary_ratios=False (out of scope)
    return constant(initial_value)


def constant(value, dtype=None, shape=None):
    a = _np(value)
    if dtype is not None:
        a = a.astype(dtype)
    return Tensor(a)


convert_to_tensor = constant


def cast(x, dtype):
    return Tensor(_np(x).astype(dtype))


def reshape(x, shape):
    return Tensor(_np(x).reshape([int(s) for s in shape]))


def expand_dims(x, axis):
    return Tensor(np.expand_dims(_np(x), axis))


def range(*args, dtype=None):                      # noqa: A001
    a = np.arange(*[int(v) for v in args])
    return Tensor(a.astype(dtype if dtype is not None else np.int32))


def zeros(shape, dtype=np.float32):
    return Tensor(np.zeros([int(s) for s in shape], dtype=dtype))


def zeros_like(x):
    return Tensor(np.zeros_like(_np(x)))


def ones_like(x):
    return Tensor(np.ones_like(_np(x)))


def shape(x):                                      # noqa: A001
    return Tensor(np.array(_np(x).shape, dtype=np.int32))


def rank(x):
    return _np(x).ndim


def _axes(axis):
    return None if axis is None else (tuple(int(a) for a in axis) if hasattr(axis, "__iter__") else int(axis))


def reduce_sum(x, axis=None):
    """tf.reduce_sum in the tensor's dtype.  (Eigen's order is not reproducible, SURVEY A7: numpy's pairwise float32 sum here.)"""
    a = _np(x)
    return Tensor(np.sum(a, axis=_axes(axis), dtype=a.dtype))


def reduce_any(x, axis=None):
    return Tensor(np.any(_np(x), axis=_axes(axis)))


def sqrt(x):
    return Tensor(np.sqrt(_np(x)))


def exp(x):
    return Tensor(np.exp(_np(x).astype(np.float64)).astype(_np(x).dtype))


def argsort(values, axis=-1, direction="ASCENDING"):
    """tf.argsort = top_k: ties go to the lower index; NaN after every number."""
    v = _np(values)
    assert v.ndim == 1
    key = -v if direction == "DESCENDING" else v
    return Tensor(np.argsort(key, kind="stable").astype(np.int32))


def argmax(x, axis=0):
    """Eigen ArgMaxTupleReducer: accumulator (0, lowest()), strict '>': the first maximum, never a NaN."""
    v = _np(x)
    assert v.ndim == 1
    best, bi = np.finfo(v.dtype).min, 0
    for i, val in enumerate(v):
        if val > best:
            best, bi = val, i
    return Tensor(np.int64(bi))


def stack(values, axis=0):
    return Tensor(np.stack([_np(v) for v in values], axis=axis))


def concat(values, axis):
    return Tensor(np.concatenate([_np(v) for v in values], axis=axis))


def gather_nd(params, indices):
    p, i = _np(params), _np(indices)
    return Tensor(p[tuple(np.moveaxis(i, -1, 0))])


def gather(params, indices, axis=0):
    return Tensor(np.take(_np(params), _np(indices), axis=axis))


def clip_by_value(x, lo, hi):
    return Tensor(np.clip(_np(x), lo, hi))


def _log(x):
    a = _np(x)
    with np.errstate(all="ignore"):
        return Tensor(np.log(a.astype(np.float64)).astype(a.dtype))


def _pow(x, y):
    a = _np(x)
    return Tensor(np.power(a, _np(y, like=a)))


def _floormod(x, y):
    a = _np(x)
    return Tensor(np.mod(a, _np(y, like=a)))


def _invert_permutation(x):
    a = _np(x)
    inv = np.empty_like(a)
    inv[a] = np.arange(a.size, dtype=a.dtype)
    return Tensor(inv)


math = types.SimpleNamespace(
    floormod=_floormod, pow=_pow, ceil=lambda x: Tensor(np.ceil(_np(x))), log=_log, sqrt=sqrt, exp=exp,
    squared_difference=lambda a, b: (Tensor(_np(a)) - b) * (Tensor(_np(a)) - b),
    expm1=lambda x: Tensor(np.expm1(_np(x).astype(np.float64)).astype(_np(x).dtype)),
    invert_permutation=_invert_permutation, is_inf=lambda x: builtins.bool(np.isinf(_np(x))), argmax=argmax)


# ---- random: the global seed of tf.random.set_seed + the op seed, SURVEY A1 -------------------------------------------------
_state = {"global": None}


def _set_seed(seed):
    _state["global"] = int(seed)


def _uniform(shape, minval=0, maxval=None, dtype=np.float32, seed=None):
    """tf.random.uniform(int32) after set_seed: only the form the reference uses -- global seed == op seed (beam_search_coder.py:38-43)."""
    assert dtype == np.int32 and minval == 1 and maxval == 10007, "only get_pseudo_random_sample's draw is stubbed"
    assert seed is not None and int(seed) == _state["global"], "the oracle's stream takes (global seed, op seed) = (seed, seed)"
    shp = [int(s) for s in shape]
    return Tensor(_O.uniform_int(int(seed), int(np.prod(shp))).reshape(shp))


def _shuffle(value, seed=None):
    assert seed is None, "Coder.split / merge shuffle without an op seed (coder.py:64)"
    a = _np(value)
    return Tensor(a[_O.tf_shuffle_perm(_state["global"], a.shape[0])])


def _normal(shape, mean=0.0, stddev=1.0, dtype=np.float32, seed=None):
    assert seed is None
    shp = [int(s) for s in shape]
    return Tensor(_O.tf_random_normal(_state["global"], int(np.prod(shp))).reshape(shp))


def _stateless_normal(shape, seed):
    shp = [int(s) for s in shape]
    return Tensor(_O.tf_stateless_normal(int(seed[0]), int(seed[1]), int(np.prod(shp))).reshape(shp))


random = types.SimpleNamespace(set_seed=_set_seed, uniform=_uniform, shuffle=_shuffle, normal=_normal,
                               stateless_normal=_stateless_normal)
