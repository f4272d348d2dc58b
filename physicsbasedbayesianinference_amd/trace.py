"""Python callables as potentials -- WITHOUT a C++ string and WITHOUT a CPU path.

The reference hands `HMC` / `Integrator` plain Python callables (src/HMC.py:52-60,
src/integrator.py:73): `potential(q) -> scalar` written on `jax.numpy`, `gradient = grad(potential)`
(src/tests/test_integrator_harmonic.py:22-24, src/tests/test_HMC.py:27-33,48-49).  A HIP kernel
cannot call Python, so such a callable is TRACED once: it is called on a symbolic position vector
(objects of this module that record arithmetic instead of performing it), and the recorded
expression becomes a potential descriptor whose arithmetic runs in the HIP kernels:

  * a quadratic form  c + b.q + q^T A q  (exact polynomial expansion of the trace)
        -> `Harmonic`, `GaussianDiag` or `GaussianDense`  (the MFMA kernels for a dense precision);
  * `-multivariate_normal.logpdf(q, mean, cov)` / `-log(multivariate_normal.pdf(...))` of this
    module's `scipy.stats`      -> `GaussianDense(mean, cov=cov)` exactly as the descriptor does it;
  * anything else               -> generated C++ source (potential AND its reverse-mode symbolic
    gradient, common subexpressions shared) compiled into the kernels by `custom.CustomPotential`.

Nothing is evaluated on the host: the trace only builds the expression.  A callable is traced ONCE per
(callable, D) and the descriptor cached: arrays it closes over are read at trace time (constants of the
kernel), as with `jax.jit`.  A callable that cannot be
traced (Python control flow on the VALUE of q, `float(q[0])`, a foreign array library) raises
`TypeError` -- there is no fallback.

The module doubles as the array namespace the callable is written on (the symbols the reference
touches, SURVEY 8c, and the usual elementwise set):

    from physicsbasedbayesianinference_amd import trace as jnp      # instead of: import jax.numpy as jnp
    from physicsbasedbayesianinference_amd.trace import grad        # instead of: from jax import grad
    from physicsbasedbayesianinference_amd.trace import multivariate_normal
    potential = lambda q: -multivariate_normal.logpdf(q, mean, cov=cov)
    hmc = HMC(ensemble, simulTime, stepSize, density, potential=potential)

(`dropin/jax/` re-exports it under the names `jax`, `jax.numpy`, `jax.scipy.stats`, so that the
reference's own import lines resolve.)  On NUMERIC input every function is NumPy's.
"""
import math

import numpy as np

__all__ = ["trace_potential", "plan_potential", "build_plan", "grad", "TracedGradient", "Sym", "SymArray", "is_symbolic",
           "dot", "matmul", "sum", "exp", "log", "log1p", "sqrt", "abs", "absolute", "tanh", "sin", "cos",
           "square", "power", "maximum", "minimum", "where", "logaddexp", "asarray", "array", "zeros",
           "ones", "zeros_like", "ones_like", "arange", "eye", "diag", "outer", "concatenate", "stack",
           "pi", "e", "inf", "linalg", "multivariate_normal", "norm", "float64", "float32"]

pi, e, inf = math.pi, math.e, math.inf
float64, float32 = np.float64, np.float32
_builtin_sum, _builtin_abs = sum, abs
MAX_NODES = 200000   # a trace larger than this would not compile into a kernel anyone wants to run


class TraceError(TypeError):
    pass


# ------------------------------------------------------------------------------------ expression DAG
class _Tracer:
    """Hash-consing node table of one trace: structurally equal expressions are ONE node."""

    def __init__(self, D):
        self.D = int(D)
        self.table = {}
        self.count = 0

    def node(self, op, args=(), val=None):
        key = (op, tuple(a.id for a in args), val)
        n = self.table.get(key)
        if n is None:
            self.count += 1
            if self.count > MAX_NODES:
                raise TraceError(f"the traced expression has more than {MAX_NODES} operations; state the "
                                 f"potential with the array functions of this namespace (dot, sum, matmul) or "
                                 f"as a descriptor / C++ source")
            n = Sym.__new__(Sym)
            n.tr, n.op, n.args, n.val, n.id = self, op, tuple(args), val, self.count
            self.table[key] = n
        return n

    def const(self, v):
        v = float(v)
        # (key on the bit pattern: 0.0 and -0.0, and NaNs, must not merge / miss)
        return self.node("const", (), np.float64(v).tobytes())


def _cval(n):
    return float(np.frombuffer(n.val, dtype=np.float64)[0])


_CMP = {"lt": "<", "le": "<=", "gt": ">", "ge": ">="}


class Sym:
    """One scalar of a traced expression."""
    __slots__ = ("tr", "op", "args", "val", "id")
    __array_ufunc__ = None      # ndarray <op> Sym defers to Sym.__r<op>__
    __array_priority__ = 1000

    # ---- construction helpers
    def _lift(self, x):
        if isinstance(x, Sym):
            if x.tr is not self.tr:
                raise TraceError("values of two different traces were mixed")
            return x
        if isinstance(x, (bool, int, float, np.integer, np.floating, np.bool_)):
            return self.tr.const(x)
        if isinstance(x, np.ndarray) and x.ndim == 0:
            return self.tr.const(x.item())
        return None

    def _bin(self, op, other, swap=False):
        o = self._lift(other)
        if o is None:
            if isinstance(other, (np.ndarray, list, tuple, SymArray)):   # scalar <op> array: broadcast
                me = SymArray(_obj(self))
                f = {"add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b,
                     "div": lambda a, b: a / b}[op]
                return me._b(other, (lambda a, b: f(b, a)) if swap else f)
            return NotImplemented
        a, b = (o, self) if swap else (self, o)
        return _binary(op, a, b)

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, True)
    def __neg__(self): return _unary("neg", self)
    def __pos__(self): return self
    def __abs__(self): return _unary("abs", self)

    def __pow__(self, o):
        if isinstance(o, (np.ndarray, list, tuple, SymArray)) and np.ndim(_obj(o)) > 0:
            return SymArray(_obj(self))._b(o, lambda a, b: a ** b)
        o = self._lift(o)
        if o is None:
            return NotImplemented
        return _pow(self, o)

    def __rpow__(self, o):
        o = self._lift(o)
        if o is None:
            return NotImplemented
        return _pow(o, self)

    def _cmp(self, op, other):
        o = self._lift(other)
        if o is None:
            if isinstance(other, (np.ndarray, list, tuple, SymArray)):
                return _compare(op, self, other)
            return NotImplemented
        return Cond(op, self, o)

    def __lt__(self, o): return self._cmp("lt", o)
    def __le__(self, o): return self._cmp("le", o)
    def __gt__(self, o): return self._cmp("gt", o)
    def __ge__(self, o): return self._cmp("ge", o)
    __hash__ = object.__hash__

    def __bool__(self):
        raise TraceError("the truth value of a traced quantity was requested: Python control flow on the VALUE "
                         "of q cannot be traced (use where / maximum / minimum)")

    def __float__(self):
        raise TraceError("float() of a traced quantity: the callable leaves the traceable namespace here")

    __int__ = __float__
    __index__ = __float__

    def __array__(self, *a, **k):
        raise TraceError("a traced quantity was handed to a NumPy routine that needs its VALUE; use the "
                         "functions of physicsbasedbayesianinference_amd.trace instead")

    # a scalar behaves like a 0-d array where the reference's code asks
    shape = ()
    ndim = 0
    size = 1

    @property
    def T(self): return self

    def sum(self, axis=None): return self

    def __repr__(self):
        return f"Sym<{self.op}#{self.id}>"


class Cond:
    """A comparison of traced scalars: only usable as the condition of where / maximum / minimum."""
    __slots__ = ("op", "a", "b")
    __array_ufunc__ = None

    def __init__(self, op, a, b):
        self.op, self.a, self.b = op, a, b

    def __bool__(self):
        raise TraceError("Python control flow on a comparison of traced values cannot be traced "
                         "(use where(cond, x, y))")


def _is_const(n, v=None):
    return n.op == "const" and (v is None or _cval(n) == v)


def _binary(op, a, b):
    tr = a.tr
    if a.op == "const" and b.op == "const":
        x, y = _cval(a), _cval(b)
        with np.errstate(all="ignore"):
            r = {"add": np.float64(x) + y, "sub": np.float64(x) - y, "mul": np.float64(x) * y,
                 "div": np.float64(x) / np.float64(y)}[op]
        return tr.const(r)
    # identities that are exact in floating point
    if op == "add":
        if _is_const(a, 0.0): return b
        if _is_const(b, 0.0): return a
    elif op == "sub":
        if _is_const(b, 0.0): return a
        if _is_const(a, 0.0): return _unary("neg", b)
        if a is b: return tr.const(0.0)
    elif op == "mul":
        if _is_const(a, 1.0): return b
        if _is_const(b, 1.0): return a
        if _is_const(a, -1.0): return _unary("neg", b)
        if _is_const(b, -1.0): return _unary("neg", a)
        if _is_const(a, 0.0) or _is_const(b, 0.0): return tr.const(0.0)   # (trace-time: q is finite)
        if a.id > b.id: a, b = b, a   # commutative: one node for x*y and y*x
    elif op == "div":
        if _is_const(b, 1.0): return a
        if _is_const(a, 0.0): return tr.const(0.0)
    if op == "add" and a.id > b.id:
        a, b = b, a
    return tr.node(op, (a, b))


def _unary(op, a):
    tr = a.tr
    if a.op == "const":
        x = np.float64(_cval(a))
        with np.errstate(all="ignore"):
            r = {"neg": lambda: -x, "abs": lambda: np.abs(x), "exp": lambda: np.exp(x), "log": lambda: np.log(x),
                 "log1p": lambda: np.log1p(x), "sqrt": lambda: np.sqrt(x), "tanh": lambda: np.tanh(x),
                 "sin": lambda: np.sin(x), "cos": lambda: np.cos(x), "sign": lambda: np.copysign(1.0, x)}[op]()
        return tr.const(r)
    if op == "neg" and a.op == "neg":
        return a.args[0]
    if op == "log":
        if a.op == "exp":                       # log(exp(x)) = x: -log(density) of an exp(...) density
            return a.args[0]
        if a.op == "div" and a.args[1].op == "const" and _cval(a.args[1]) > 0:   # log(x / c)
            return _binary("sub", _unary("log", a.args[0]), tr.const(np.log(_cval(a.args[1]))))
        if a.op == "mul" and a.args[0].op == "const" and _cval(a.args[0]) > 0:   # log(c * x)
            return _binary("add", tr.const(np.log(_cval(a.args[0]))), _unary("log", a.args[1]))
        if a.op == "gausspdf":                  # log(multivariate_normal.pdf) -> logpdf
            return tr.node("gausslog", (), a.val)
    if op == "exp" and a.op == "gausslog":
        return tr.node("gausspdf", (), a.val)
    if op == "abs" and a.op in ("abs", "exp", "sqrt"):
        return a
    return tr.node(op, (a,))


def _pow(a, b):
    tr = a.tr
    if b.op == "const":
        y = _cval(b)
        if y == 0.0: return tr.const(1.0)
        if y == 1.0: return a
        if a.op == "const":
            with np.errstate(all="ignore"):
                return tr.const(np.float64(_cval(a)) ** y)
        if y == 2.0:
            if a.op == "sqrt":                  # (sqrt(s))**2 = s  (norm(x)**2, src/tests/test_HMC.py:28)
                return a.args[0]
            return _binary("mul", a, a)
        if y == 0.5:
            return _unary("sqrt", a)
        if y == -1.0:
            return _binary("div", tr.const(1.0), a)
        if y == float(int(y)) and _builtin_abs(y) <= 64:
            return tr.node("powi", (a,), int(y))
        return tr.node("pow", (a, b))
    # general a**b = exp(b * log a)
    return _unary("exp", _binary("mul", b, _unary("log", a)))


def _addn(terms, empty=None):
    """Sum of many terms as ONE n-ary node (order kept: it is the summation order of the kernel)."""
    terms = [t for t in terms if not _is_const(t, 0.0)]
    if not terms:
        return empty
    if len(terms) == 1:
        return terms[0]
    if len(terms) == 2:
        return _binary("add", terms[0], terms[1])
    return terms[0].tr.node("addn", tuple(terms))


def _select(cond, x, y):
    if x is y:
        return x
    return x.tr.node("select", (cond.a, cond.b, x, y), cond.op)


# ------------------------------------------------------------------------------------ arrays
class SymArray:
    """An array of traced scalars (object ndarray inside): the array API the reference's callables use."""
    __array_ufunc__ = None
    __array_priority__ = 1000

    def __init__(self, a):
        self.a = np.asarray(a, dtype=object)

    # -- shape protocol
    @property
    def shape(self): return self.a.shape
    @property
    def ndim(self): return self.a.ndim
    @property
    def size(self): return self.a.size
    @property
    def T(self): return SymArray(self.a.T)
    def __len__(self): return len(self.a)
    def reshape(self, *s): return SymArray(self.a.reshape(*s))
    def ravel(self): return SymArray(self.a.ravel())
    flatten = ravel
    def __iter__(self): return (_wrap(x) for x in self.a)
    def __getitem__(self, k): return _wrap(self.a[k])
    def astype(self, *a, **k): return self
    def copy(self): return SymArray(self.a.copy())

    def __setitem__(self, k, v):
        self.a[k] = v.a if isinstance(v, SymArray) else v

    def __array__(self, *a, **k):
        raise TraceError("a traced array was handed to a NumPy routine that needs its VALUES; use the functions "
                         "of physicsbasedbayesianinference_amd.trace instead")

    def __bool__(self):
        raise TraceError("the truth value of a traced array was requested (Python control flow on q)")

    # -- arithmetic: NumPy broadcasts the object arrays and calls Sym's operators elementwise
    def _b(self, o, f):
        if isinstance(o, (SymArray, Sym)):
            o = _obj(o)
        elif isinstance(o, (list, tuple)):
            o = _obj(o)
        return _wrap(f(self.a, o))

    def __add__(self, o): return self._b(o, lambda a, b: a + b)
    def __radd__(self, o): return self._b(o, lambda a, b: b + a)
    def __sub__(self, o): return self._b(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._b(o, lambda a, b: b - a)
    def __mul__(self, o): return self._b(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._b(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._b(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._b(o, lambda a, b: b / a)
    def __pow__(self, o): return self._b(o, lambda a, b: a ** b)
    def __rpow__(self, o): return self._b(o, lambda a, b: b ** a)
    def __neg__(self): return _wrap(-self.a)
    def __pos__(self): return self
    def __abs__(self): return absolute(self)
    def __matmul__(self, o): return matmul(self, o)
    def __rmatmul__(self, o): return matmul(o, self)
    def __lt__(self, o): return _compare("lt", self, o)
    def __le__(self, o): return _compare("le", self, o)
    def __gt__(self, o): return _compare("gt", self, o)
    def __ge__(self, o): return _compare("ge", self, o)

    def sum(self, axis=None): return sum(self, axis=axis)
    def dot(self, o): return dot(self, o)

    def __repr__(self):
        return f"SymArray(shape={self.shape})"


def _wrap(x):
    if isinstance(x, np.ndarray):
        if x.dtype != object:
            return x
        return SymArray(x) if x.ndim else x.item()
    return x


def is_symbolic(x):
    if isinstance(x, (Sym, SymArray, Cond)):
        return True
    if isinstance(x, np.ndarray) and x.dtype == object:
        return any(isinstance(v, (Sym, Cond)) for v in x.ravel())
    if isinstance(x, (list, tuple)):
        return any(is_symbolic(v) for v in x)
    return False


def _obj(x):
    """object ndarray view of anything array-like that may hold traced scalars"""
    if isinstance(x, SymArray):
        return x.a
    if isinstance(x, Sym):
        a = np.empty((), dtype=object)
        a[()] = x
        return a
    if isinstance(x, (list, tuple)) and is_symbolic(x):
        a = np.empty(len(x), dtype=object)
        for i, v in enumerate(x):
            a[i] = v.a if isinstance(v, SymArray) else v
        if any(isinstance(v, np.ndarray) for v in a):   # nested: stack rows
            return np.stack([np.asarray(v, dtype=object) for v in a])
        return a
    return np.asarray(x)


def _tracer_of(*xs):
    for x in xs:
        for v in np.asarray(_obj(x), dtype=object).ravel():
            if isinstance(v, Sym):
                return v.tr
            if isinstance(v, Cond):
                return v.a.tr
    return None


def _lift_all(tr, a):
    """object array whose every entry is a Sym of `tr`"""
    a = np.asarray(a)
    out = np.empty(a.shape, dtype=object)
    flat_in, flat_out = a.ravel(), out.ravel()
    for i, v in enumerate(flat_in):
        flat_out[i] = v if isinstance(v, Sym) else tr.const(v)
    return out


# ------------------------------------------------------------------------------------ the namespace
def _elementwise(op, np_fn):
    def f(x, *args, **kw):
        if not is_symbolic(x):
            return np_fn(x, *args, **kw)
        if isinstance(x, Sym):
            return _unary(op, x)
        a = _obj(x)
        out = np.empty(a.shape, dtype=object)
        fo = out.ravel()
        tr = _tracer_of(x)
        for i, v in enumerate(a.ravel()):
            fo[i] = _unary(op, v if isinstance(v, Sym) else tr.const(v))
        return _wrap(out)
    f.__name__ = np_fn.__name__
    f.__doc__ = f"{np_fn.__name__}: NumPy's on numbers, recorded on traced values."
    return f


exp = _elementwise("exp", np.exp)
log = _elementwise("log", np.log)
log1p = _elementwise("log1p", np.log1p)
sqrt = _elementwise("sqrt", np.sqrt)
absolute = _elementwise("abs", np.abs)
abs = absolute   # noqa: A001  (the namespace mirrors jax.numpy)
tanh = _elementwise("tanh", np.tanh)
sin = _elementwise("sin", np.sin)
cos = _elementwise("cos", np.cos)


def square(x):
    return x * x if is_symbolic(x) else np.square(x)


def power(x, y):
    if not (is_symbolic(x) or is_symbolic(y)):
        return np.power(x, y)
    if not is_symbolic(x):
        x = np.asarray(x)
        return _wrap(x ** _obj(y)) if x.ndim else (float(x) ** y)
    return x ** (y.a if isinstance(y, SymArray) else y)


def sum(x, axis=None, **kw):   # noqa: A001
    if not is_symbolic(x):
        return np.sum(x, axis=axis, **kw)
    a = _obj(x)
    tr = _tracer_of(x)
    a = _lift_all(tr, a)
    if axis is None:
        return _addn(list(a.ravel()), tr.const(0.0))
    a = np.moveaxis(a, axis, -1)
    out = np.empty(a.shape[:-1], dtype=object)
    for idx in np.ndindex(*a.shape[:-1]):
        out[idx] = _addn(list(a[idx]), tr.const(0.0))
    return _wrap(out)


def _dot1(tr, u, v):
    """sum_k u_k v_k of two equally long vectors (numbers and/or traced), summed in index order"""
    terms = []
    for x, y in zip(u, v):
        xs, ys = isinstance(x, Sym), isinstance(y, Sym)
        if not xs and not ys:
            terms.append(tr.const(float(x) * float(y)))
        else:
            xx = x if xs else tr.const(x)
            yy = y if ys else tr.const(y)
            terms.append(_binary("mul", xx, yy))
    return _addn(terms, tr.const(0.0))


def matmul(x, y):
    if not (is_symbolic(x) or is_symbolic(y)):
        return np.matmul(x, y)
    a, b = _obj(x), _obj(y)
    tr = _tracer_of(x, y)
    if a.ndim == 0 or b.ndim == 0:
        return _wrap(a * b)
    if a.ndim == 1 and b.ndim == 1:
        if a.shape != b.shape:
            raise ValueError(f"shapes {a.shape} and {b.shape} not aligned")
        return _dot1(tr, a, b)
    if a.ndim == 2 and b.ndim == 1:
        if a.shape[1] != b.shape[0]:
            raise ValueError(f"shapes {a.shape} and {b.shape} not aligned")
        out = np.empty(a.shape[0], dtype=object)
        for i in range(a.shape[0]):
            out[i] = _dot1(tr, a[i], b)
        return SymArray(out)
    if a.ndim == 1 and b.ndim == 2:
        if a.shape[0] != b.shape[0]:
            raise ValueError(f"shapes {a.shape} and {b.shape} not aligned")
        out = np.empty(b.shape[1], dtype=object)
        for j in range(b.shape[1]):
            out[j] = _dot1(tr, a, b[:, j])
        return SymArray(out)
    if a.ndim == 2 and b.ndim == 2:
        if a.shape[1] != b.shape[0]:
            raise ValueError(f"shapes {a.shape} and {b.shape} not aligned")
        out = np.empty((a.shape[0], b.shape[1]), dtype=object)
        for i in range(a.shape[0]):
            for j in range(b.shape[1]):
                out[i, j] = _dot1(tr, a[i], b[:, j])
        return SymArray(out)
    raise TraceError("matmul / dot of traced arrays: at most 2 dimensions")


dot = matmul


def outer(x, y):
    if not (is_symbolic(x) or is_symbolic(y)):
        return np.outer(x, y)
    a, b = _obj(x).ravel(), _obj(y).ravel()
    return _wrap(a[:, None] * b[None, :])


def _cond_array(c):
    if isinstance(c, SymArray):
        return c.a
    if isinstance(c, Cond):
        a = np.empty((), dtype=object)
        a[()] = c
        return a
    return c


def where(cond, x, y):
    if not (is_symbolic(cond) or is_symbolic(x) or is_symbolic(y)):
        return np.where(cond, x, y)
    tr = _tracer_of(cond, x, y)
    c, xa, ya = np.broadcast_arrays(np.asarray(_cond_array(cond), dtype=object), _obj(x), _obj(y))
    out = np.empty(c.shape, dtype=object)
    for idx in np.ndindex(*c.shape):
        ci, xi, yi = c[idx], xa[idx], ya[idx]
        xi = xi if isinstance(xi, Sym) else tr.const(xi)
        yi = yi if isinstance(yi, Sym) else tr.const(yi)
        if isinstance(ci, Cond):
            out[idx] = _select(ci, xi, yi)
        elif isinstance(ci, (bool, np.bool_)):
            out[idx] = xi if ci else yi
        else:
            raise TraceError("where(): the condition must be a comparison")
    return _wrap(out)


def maximum(x, y):
    if not (is_symbolic(x) or is_symbolic(y)):
        return np.maximum(x, y)
    return where(_gt(x, y), x, y)


def minimum(x, y):
    if not (is_symbolic(x) or is_symbolic(y)):
        return np.minimum(x, y)
    return where(_gt(x, y), y, x)


def _compare(op, x, y):
    """elementwise comparison of traced values: an object array of Cond (a Cond for scalars)"""
    tr = _tracer_of(x, y)
    xa, ya = np.broadcast_arrays(np.asarray(_obj(x), dtype=object), np.asarray(_obj(y), dtype=object))
    out = np.empty(xa.shape, dtype=object)
    for idx in np.ndindex(*xa.shape):
        a, b = xa[idx], ya[idx]
        a = a if isinstance(a, Sym) else tr.const(a)
        b = b if isinstance(b, Sym) else tr.const(b)
        out[idx] = Cond(op, a, b)
    return SymArray(out) if out.ndim else out.item()


def _gt(x, y):
    return _compare("gt", x, y)


def logaddexp(x, y):
    if not (is_symbolic(x) or is_symbolic(y)):
        return np.logaddexp(x, y)
    return maximum(x, y) + log1p(exp(-absolute(x - y)))


def asarray(x, dtype=None, **kw):
    if is_symbolic(x):
        return x if isinstance(x, (Sym, SymArray)) else _wrap(_obj(x))
    return np.asarray(x, dtype=dtype, **kw)


array = asarray
zeros, ones, arange, eye, diag = np.zeros, np.ones, np.arange, np.eye, np.diag


def zeros_like(x, **kw):
    return np.zeros(np.shape(_obj(x))) if is_symbolic(x) else np.zeros_like(x, **kw)


def ones_like(x, **kw):
    return np.ones(np.shape(_obj(x))) if is_symbolic(x) else np.ones_like(x, **kw)


def concatenate(xs, axis=0):
    if not is_symbolic(list(xs)):
        return np.concatenate(xs, axis=axis)
    return _wrap(np.concatenate([np.atleast_1d(np.asarray(_obj(x), dtype=object)) for x in xs], axis=axis))


def stack(xs, axis=0):
    if not is_symbolic(list(xs)):
        return np.stack(xs, axis=axis)
    return _wrap(np.stack([np.asarray(_obj(x), dtype=object) for x in xs], axis=axis))


class _Linalg:
    """jnp.linalg as far as the reference goes: norm (src/tests/test_HMC.py:28); numeric det / inv / solve."""
    inv, det, slogdet, solve, cholesky = (staticmethod(f) for f in
                                          (np.linalg.inv, np.linalg.det, np.linalg.slogdet, np.linalg.solve,
                                           np.linalg.cholesky))

    @staticmethod
    def norm(x, ord=None, axis=None, **kw):
        if not is_symbolic(x):
            return np.linalg.norm(x, ord=ord, axis=axis, **kw)
        if ord not in (None, 2):
            raise TraceError("linalg.norm of a traced array: the 2-norm only")
        return sqrt(sum(x * x, axis=axis))


linalg = _Linalg()


# ---- scipy.stats as far as the reference goes (src/tests/test_HMC.py:17,48-49,124-125)
class _MultivariateNormal:
    """`multivariate_normal.pdf / logpdf(q, mean, cov)`.  On a traced q the call becomes ONE node that
    the descriptor builder maps to `GaussianDense(mean, cov=cov)` -- the MFMA kernels, not an expanded
    polynomial; on numbers it is SciPy's."""

    @staticmethod
    def _node(kind, x, mean, cov):
        tr = _tracer_of(x)
        a = _obj(x)
        if a.ndim != 1:
            raise TraceError("multivariate_normal on a traced q: q must be the (D,) position of one chain")
        for j, v in enumerate(a):
            if not (isinstance(v, Sym) and v.op == "in" and v.val == j):
                raise TraceError("multivariate_normal on a traced argument: only q itself (mean= carries the shift)")
        D = a.shape[0]
        mean = np.zeros(D) if mean is None else np.broadcast_to(np.asarray(mean, dtype=np.float64), (D,))
        cov = np.asarray(1.0 if cov is None else cov, dtype=np.float64)
        cov = np.eye(D) * cov if cov.ndim == 0 else (np.diag(cov) if cov.ndim == 1 else cov)
        if cov.shape != (D, D):
            raise ValueError(f"cov must be ({D}, {D})")
        return tr.node(kind, (), (mean.tobytes(), np.ascontiguousarray(cov).tobytes()))

    def logpdf(self, x, mean=None, cov=1, **kw):
        if not is_symbolic(x):
            from scipy.stats import multivariate_normal as mvn
            return mvn.logpdf(x, mean=mean, cov=cov, **kw)
        return self._node("gausslog", x, mean, cov)

    def pdf(self, x, mean=None, cov=1, **kw):
        if not is_symbolic(x):
            from scipy.stats import multivariate_normal as mvn
            return mvn.pdf(x, mean=mean, cov=cov, **kw)
        return self._node("gausspdf", x, mean, cov)


class _Norm:
    """`norm.logpdf / pdf(x, loc, scale)` elementwise."""

    @staticmethod
    def logpdf(x, loc=0.0, scale=1.0):
        if not (is_symbolic(x) or is_symbolic(loc) or is_symbolic(scale)):
            from scipy.stats import norm as sn
            return sn.logpdf(x, loc, scale)
        z = (x - loc) / scale
        return -0.5 * (z * z) - log(scale) - 0.5 * math.log(2.0 * math.pi)

    @classmethod
    def pdf(cls, x, loc=0.0, scale=1.0):
        if not (is_symbolic(x) or is_symbolic(loc) or is_symbolic(scale)):
            from scipy.stats import norm as sn
            return sn.pdf(x, loc, scale)
        return exp(cls.logpdf(x, loc, scale))


multivariate_normal = _MultivariateNormal()
norm = _Norm()


# ------------------------------------------------------------------------------------ grad
class TracedGradient:
    """`grad(potential)`: what `jax.grad` returns in the reference's scripts
    (src/tests/test_integrator_harmonic.py:24, src/HMC.py:60).  Handing it to `Integrator` / `HMC` as
    `gradient=` makes them trace `potential` and differentiate the trace; called on numbers it evaluates
    the gradient with the HIP kernel of the traced descriptor."""

    def __init__(self, fn):
        if not callable(fn):
            raise TypeError("grad() takes a callable")
        self.fn = fn

    def __call__(self, q):
        q = np.asarray(q, dtype=np.float64)
        return trace_potential(self.fn, D=q.shape[0]).gradient(q)


def grad(fn):
    return TracedGradient(fn)


# ------------------------------------------------------------------------------------ analysis
def _topo(roots):
    order, seen = [], set()
    stack_ = [(r, False) for r in roots]
    while stack_:
        n, done = stack_.pop()
        if done:
            order.append(n)
            continue
        if n.id in seen:
            continue
        seen.add(n.id)
        stack_.append((n, True))
        for a in n.args:
            if a.id not in seen:
                stack_.append((a, False))
    return order


def _quadratic(root, D):
    """(c, b, A) with root == c + b.q + q^T A q (A upper triangular, exact expansion of the trace's
    polynomial arithmetic), or None when the expression is not a polynomial of degree <= 2."""
    polys = {}
    for n in _topo([root]):
        op = n.op
        if op == "const":
            p = (_cval(n), {}, {})
        elif op == "in":
            p = (0.0, {n.val: 1.0}, {})
        elif op in ("add", "sub", "addn"):
            c, b, A = 0.0, {}, {}
            for k, a in enumerate(n.args):
                pa = polys[a.id]
                if pa is None:
                    c = None
                    break
                s = -1.0 if (op == "sub" and k == 1) else 1.0
                c += s * pa[0]
                for j, v in pa[1].items():
                    b[j] = b.get(j, 0.0) + s * v
                for j, v in pa[2].items():
                    A[j] = A.get(j, 0.0) + s * v
            p = None if c is None else (c, b, A)
        elif op == "neg":
            pa = polys[n.args[0].id]
            p = None if pa is None else (-pa[0], {j: -v for j, v in pa[1].items()}, {j: -v for j, v in pa[2].items()})
        elif op == "mul":
            pa, pb = polys[n.args[0].id], polys[n.args[1].id]
            if pa is None or pb is None or (pa[2] and (pb[1] or pb[2])) or (pb[2] and pa[1]):
                p = None
            else:
                c = pa[0] * pb[0]
                b, A = {}, {}
                for j, v in pa[1].items():
                    if pb[0] != 0.0: b[j] = b.get(j, 0.0) + v * pb[0]
                for j, v in pb[1].items():
                    if pa[0] != 0.0: b[j] = b.get(j, 0.0) + v * pa[0]
                for j, v in pa[2].items():
                    A[j] = A.get(j, 0.0) + v * pb[0]
                for j, v in pb[2].items():
                    A[j] = A.get(j, 0.0) + v * pa[0]
                for i, u in pa[1].items():
                    for j, v in pb[1].items():
                        k = (i, j) if i <= j else (j, i)
                        A[k] = A.get(k, 0.0) + u * v
                p = (c, b, A)
        elif op == "div":
            pa, pb = polys[n.args[0].id], polys[n.args[1].id]
            if pa is None or pb is None or pb[1] or pb[2] or pb[0] == 0.0:
                p = None
            else:
                p = (pa[0] / pb[0], {j: v / pb[0] for j, v in pa[1].items()}, {j: v / pb[0] for j, v in pa[2].items()})
        elif op == "powi" and n.val == 2:
            pa = polys[n.args[0].id]
            p = None
            if pa is not None and not pa[2]:
                A = {}
                for i, u in pa[1].items():
                    for j, v in pa[1].items():
                        k = (i, j) if i <= j else (j, i)
                        A[k] = A.get(k, 0.0) + u * v
                p = (pa[0] * pa[0], {j: 2.0 * pa[0] * v for j, v in pa[1].items()}, A)
        else:
            p = None
        polys[n.id] = p
    return polys[root.id]


def _plan_from_quadratic(poly, D):
    """Harmonic / GaussianDiag / GaussianDense parameters for U = c + b.q + q^T A q, or None when the form
    has no such descriptor (not bounded below along an axis, singular with a linear term, ...)."""
    c, b, A = poly
    P = np.zeros((D, D))
    for (i, j), v in A.items():
        if i == j:
            P[i, i] += 2.0 * v        # q^T A q = 0.5 q^T P q
        else:
            P[i, j] += v
            P[j, i] += v
    bv = np.zeros(D)
    for j, v in b.items():
        bv[j] = v
    diagonal = not np.any(P - np.diag(np.diag(P)))
    if diagonal:
        prec = np.diag(P).copy()
        if np.any(prec < 0) or np.any((prec == 0) & (bv != 0)) or not np.any(prec > 0):
            return None
        mean = np.where(prec > 0, -bv / np.where(prec > 0, prec, 1.0), 0.0) + 0.0   # (+ 0.0: no -0.0 means)
        const = c - 0.5 * float(np.sum(prec * mean * mean))
        if not np.any(mean) and const == 0.0:
            return {"kind": "harmonic", "springConsts": prec}            # src/potential.py:18-27 exactly
        return {"kind": "gauss_diag", "mean": mean, "prec": prec, "const": const}
    if np.any(bv):
        try:
            mean = -np.linalg.solve(P, bv) + 0.0
        except np.linalg.LinAlgError:
            return None
    else:
        mean = np.zeros(D)
    const = c - 0.5 * float(mean @ P @ mean)
    return {"kind": "gauss_dense", "mean": mean, "precision": P, "const": const}


# ------------------------------------------------------------------------------------ differentiation
def _gradient_nodes(root, D):
    """Reverse-mode differentiation OF THE TRACE: D expressions dU/dq_j sharing subexpressions with U."""
    tr = root.tr
    order = _topo([root])
    adj = {root.id: [tr.const(1.0)]}
    grads = [None] * D
    for n in reversed(order):
        parts = adj.pop(n.id, None)
        if not parts:
            continue
        g = _addn(parts)
        if g is None:
            continue
        op, a = n.op, n.args

        def push(node, val):
            if node.op not in ("const", "cparam"):
                adj.setdefault(node.id, []).append(val)
        if op == "in":
            grads[n.val] = g
        elif op in ("add", "addn"):
            for x in a:
                push(x, g)
        elif op == "sub":
            push(a[0], g)
            push(a[1], _unary("neg", g))
        elif op == "neg":
            push(a[0], _unary("neg", g))
        elif op == "mul":
            push(a[0], _binary("mul", g, a[1]))
            push(a[1], _binary("mul", g, a[0]))
        elif op == "div":
            push(a[0], _binary("div", g, a[1]))
            push(a[1], _unary("neg", _binary("div", _binary("mul", g, n), a[1])))
        elif op == "exp":
            push(a[0], _binary("mul", g, n))
        elif op == "log":
            push(a[0], _binary("div", g, a[0]))
        elif op == "log1p":
            push(a[0], _binary("div", g, _binary("add", tr.const(1.0), a[0])))
        elif op == "sqrt":
            push(a[0], _binary("div", g, _binary("mul", tr.const(2.0), n)))
        elif op == "abs":
            push(a[0], _binary("mul", g, _unary("sign", a[0])))
        elif op == "tanh":
            push(a[0], _binary("mul", g, _binary("sub", tr.const(1.0), _binary("mul", n, n))))
        elif op == "sin":
            push(a[0], _binary("mul", g, _unary("cos", a[0])))
        elif op == "cos":
            push(a[0], _unary("neg", _binary("mul", g, _unary("sin", a[0]))))
        elif op == "powi":
            k = n.val
            push(a[0], _binary("mul", g, _binary("mul", tr.const(float(k)), _pow(a[0], tr.const(float(k - 1))))))
        elif op == "pow":   # constant real exponent
            y = _cval(a[1])
            push(a[0], _binary("mul", g, _binary("mul", a[1], _pow(a[0], tr.const(y - 1.0)))))
        elif op == "select":
            c = Cond(n.val, a[0], a[1])
            zero = tr.const(0.0)
            push(a[2], _select(c, g, zero))
            push(a[3], _select(c, zero, g))
        elif op in ("sign", "const", "cparam"):
            pass
        elif op in ("gausslog", "gausspdf"):
            raise TraceError("multivariate_normal.pdf / logpdf may only appear as  -logpdf(q, ...)  or  "
                             "-log(pdf(q, ...))  (optionally plus a constant) in a traced potential")
        else:
            raise TraceError(f"no derivative rule for traced operation {op!r}")
    zero = tr.const(0.0)
    return [g if g is not None else zero for g in grads]


# ------------------------------------------------------------------------------------ code generation
def _literal(v):
    if v != v:
        return "T(NAN)"
    if v in (math.inf, -math.inf):
        return "T(INFINITY)" if v > 0 else "T(-INFINITY)"
    return f"T({float(v).hex()})"


def _emit(order, name_of, lines):
    for n in order:
        if n.id in name_of:
            continue
        op = n.op
        a = [name_of[x.id] for x in n.args]
        if op == "const":
            name_of[n.id] = _literal(_cval(n))
            continue
        if op == "in":
            name_of[n.id] = f"q[{n.val}]"
            continue
        if op == "cparam":   # a per-iteration constant of a rolled sum: declared at the top of its loop
            name_of[n.id] = f"c{n.val}"
            continue
        name = f"t{n.id}"
        name_of[n.id] = name
        if op == "addn":
            # the terms in order, eight per statement (a statement per term would drown the compiler)
            acc = None
            for k in range(0, len(a), 8):
                chunk = a[k:k + 8]
                expr = chunk[0] if acc is None else f"{acc} + {chunk[0]}"
                for x in chunk[1:]:
                    expr = f"({expr}) + {x}"
                acc = name if k + 8 >= len(a) else f"{name}_{k // 8}"
                lines.append(f"    const T {acc} = {expr};")
            continue
        if op in ("add", "sub", "mul", "div"):
            expr = f"{a[0]} {dict(add='+', sub='-', mul='*', div='/')[op]} {a[1]}"
        elif op == "neg":
            expr = f"-{a[0]}"
        elif op in ("exp", "log", "log1p", "sqrt", "tanh", "sin", "cos"):
            expr = f"{op}({a[0]})"
        elif op == "abs":
            expr = f"fabs({a[0]})"
        elif op == "sign":
            expr = f"copysign(T(1), {a[0]})"
        elif op == "pow":
            expr = f"pow({a[0]}, {a[1]})"
        elif op == "powi":
            k = n.val
            base, m = a[0], _builtin_abs(k)
            prod = base
            for _ in range(m - 1):
                prod = f"({prod}) * {base}"
            expr = prod if k > 0 else f"T(1) / ({prod})"
        elif op == "select":
            expr = f"({a[0]} {_CMP[n.val]} {a[1]}) ? {a[2]} : {a[3]}"
        else:
            raise TraceError(f"traced operation {op!r} cannot be emitted")
        lines.append(f"    const T {name} = {expr};")


def generate_source(root, grads):
    """The C++ source of custom.CustomPotential's contract for a traced potential and its gradient."""
    pot_lines, names = [], {}
    _emit(_topo([root]), names, pot_lines)
    src = ["// generated by physicsbasedbayesianinference_amd.trace from a Python callable",
           "template <class Q>", "PBBI_FN T potential(const Q& q, int D, const T* prm) {"]
    src += pot_lines + [f"    return {names[root.id]};", "}"]
    g_lines, names = [], {}
    _emit(_topo(grads), names, g_lines)
    src += ["template <class Q, class G>", "PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {"]
    src += g_lines + [f"    g[{j}] = {names[g.id]};" for j, g in enumerate(grads)] + ["}"]
    return "\n".join(src) + "\n"


# ------------------------------------------------------------------------------------ sums over data, rolled
# A likelihood is a sum over observations of terms of ONE shape -- sum_i softplus(x_i . w) - y_i x_i . w --
# and the trace unrolls it into M copies that differ in their constants only.  As straight-line code that is
# thousands of statements inlined into every kernel of the plugin: hipcc needs minutes for M = 64 and does not
# finish for M = 256.  So the terms of the top-level sum are grouped by SHAPE (the expression with its constants
# blanked); a group with many members becomes a loop over a table of constants (the plugin's `prm` array, read
# with wave-uniform addresses), whose body -- and the body's symbolic gradient -- is generated once.
ROLL_MIN_TERMS = 8        # fewer terms of a shape stay unrolled
ROLL_MIN_OPS = 400        # small traces stay straight-line code


def _flatten_sum(root):
    """root == sum of sign * term.  The top-level chain of binary add / sub / neg is opened; the arguments of an
    n-ary sum (`sum(array)`, `dot`) met on the way ARE the terms -- they are not opened further, so that what the
    callable wrote per observation stays one term of one shape."""
    out, stack_ = [], [(root, 1.0)]
    while stack_:
        n, sg = stack_.pop()
        if n.op == "addn":
            out.extend((sg, a) for a in n.args)
        elif n.op == "add":
            for a in reversed(n.args):
                stack_.append((a, sg))
        elif n.op == "sub":
            stack_.append((n.args[1], -sg))
            stack_.append((n.args[0], sg))
        elif n.op == "neg":
            stack_.append((n.args[0], -sg))
        else:
            out.append((sg, n))
    return out


def _shape_of(node, seen, consts):
    """Signature of one term's expression DAG with its constants blanked: a sub-expression met again is a
    back-reference (it is ONE value in the loop body too), constants are collected at their first visit."""
    idx = seen.get(node.id)
    if idx is not None:
        return ("ref", idx)
    seen[node.id] = len(seen)
    if node.op == "const":
        consts.append(_cval(node))
        return "C"
    if node.op == "in":
        return ("in", node.val)
    return (node.op, node.val if node.op in ("powi", "select") else None,
            tuple(_shape_of(a, seen, consts) for a in node.args))


def _template_of(node, memo, counter):
    """the term with its k-th constant (first-visit order of _shape_of) replaced by the leaf cparam(k)"""
    hit = memo.get(node.id)
    if hit is not None:
        return hit
    tr = node.tr
    if node.op == "const":
        r = tr.node("cparam", (), counter[0])
        counter[0] += 1
    elif node.op == "in":
        r = node
    else:
        args = tuple(_template_of(a, memo, counter) for a in node.args)
        # (no simplification: the slots must stay where they are)
        r = tr.node(node.op, args, node.val if node.op not in ("add", "sub", "mul", "div") else None)
    memo[node.id] = r
    return r


def _roll_sums(root):
    """(rest, groups): root == rest + sum over groups of sum_i sign * body(q; table[i]).  rest is a Sym or None."""
    terms = _flatten_sum(root)
    by_shape, order = {}, []
    for sg, n in terms:
        consts = []
        sig = _shape_of(n, {}, consts)
        key = None if sig == "C" else sig
        if key not in by_shape:
            by_shape[key] = []
            order.append(key)
        by_shape[key].append((sg, n, consts))
    rest_terms, groups = [], []
    for key in order:
        members = by_shape[key]
        K = len(members[0][2])
        if key is None or len(members) < ROLL_MIN_TERMS or K == 0:
            rest_terms += [(sg, n) for sg, n, _ in members]
            continue
        body = _template_of(members[0][1], {}, [0])
        # the sign rides as one more table column
        table = np.array([list(cs) + [sg] for sg, _, cs in members], dtype=np.float64)
        groups.append({"body": body, "K": K, "table": table})
    tr = root.tr
    rest = None
    if rest_terms:
        rest = _addn([n if sg > 0 else _unary("neg", n) for sg, n in rest_terms], tr.const(0.0))
    return rest, groups


def generate_source_rolled(rest, groups, D):
    """potential / gradient source with one loop per rolled group; returns (source, params)"""
    tr = groups[0]["body"].tr
    zero = tr.const(0.0)
    offsets, params, off = [], [], 0
    for g in groups:
        offsets.append(off)
        params.append(g["table"].ravel())
        off += g["table"].size
    src = ["// generated by physicsbasedbayesianinference_amd.trace from a Python callable (sums over data rolled into loops)",
           "template <class Q>", "PBBI_FN T potential(const Q& q, int D, const T* prm) {"]
    lines, names = [], {}
    if rest is not None:
        _emit(_topo([rest]), names, lines)
        src += lines + [f"    T s = {names[rest.id]};"]
    else:
        src += ["    T s = T(0);"]
    for gi, g in enumerate(groups):
        K, M = g["K"], g["table"].shape[0]
        body_lines, bn = [], {}
        _emit(_topo([g["body"]]), bn, body_lines)
        src += [f"    for (int i = 0; i < {M}; ++i) {{", f"        const T* row = prm + {offsets[gi]} + i * {K + 1};"]
        src += [f"        const T c{k} = row[{k}];" for k in range(K)]
        src += ["    " + ln for ln in body_lines] + [f"        s += row[{K}] * {bn[g['body'].id]};", "    }"]
    src += ["    return s;", "}"]
    src += ["template <class Q, class G>", "PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {"]
    lines, names = [], {}
    rest_grads = _gradient_nodes(rest, D) if rest is not None else [zero] * D
    _emit(_topo(rest_grads), names, lines)
    src += lines + [f"    T g{j} = {names[rest_grads[j].id]};" for j in range(D)]
    for gi, g in enumerate(groups):
        K, M = g["K"], g["table"].shape[0]
        bg = _gradient_nodes(g["body"], D)
        live = [j for j in range(D) if not _is_const(bg[j], 0.0)]
        body_lines, bn = [], {}
        _emit(_topo([bg[j] for j in live]), bn, body_lines)
        src += [f"    for (int i = 0; i < {M}; ++i) {{", f"        const T* row = prm + {offsets[gi]} + i * {K + 1};"]
        src += [f"        const T c{k} = row[{k}];" for k in range(K)] + [f"        const T sg = row[{K}];"]
        src += ["    " + ln for ln in body_lines] + [f"        g{j} += sg * {bn[bg[j].id]};" for j in live] + ["    }"]
    src += [f"    g[{j}] = g{j};" for j in range(D)] + ["}"]
    return "\n".join(src) + "\n", np.concatenate(params)


# ------------------------------------------------------------------------------------ entry point
def _symbolic_q(D):
    tr = _Tracer(D)
    a = np.empty(D, dtype=object)
    for j in range(D):
        a[j] = tr.node("in", (), j)
    return tr, SymArray(a)


def _as_scalar(tr, r, what):
    if isinstance(r, SymArray):
        if r.size != 1:
            raise TraceError(f"{what} must return a scalar, got shape {r.shape}")
        r = r.a.ravel()[0]
    if isinstance(r, Sym):
        return r
    if isinstance(r, (int, float, np.integer, np.floating)) or (isinstance(r, np.ndarray) and r.size == 1):
        return tr.const(float(np.asarray(r).ravel()[0]))
    raise TraceError(f"{what} returned {type(r).__name__}, not a value of the traceable namespace")


def _run(fn, q, what):
    try:
        return fn(q)
    except TraceError:
        raise
    except Exception as exc:   # the callable used something that needs the VALUE of q
        raise TraceError(f"{what} {fn!r} could not be traced ({type(exc).__name__}: {exc}).  Python callables "
                         f"are traced on the array namespace physicsbasedbayesianinference_amd.trace (import it "
                         f"in place of jax.numpy / numpy); there is no CPU fallback.  Otherwise pass a potential "
                         f"descriptor or custom.CustomPotential(D, source, params)") from exc


_CACHE = {}


def plan_potential(potential=None, D=None, gradient=None, density=None, prefer=None):
    """Host-only half of trace_potential: trace the callable(s) and decide what serves them.  Returns a
    dict with "kind" in {"harmonic", "gauss_diag", "gauss_dense", "source"} and that descriptor's
    parameters (or the generated C++ "source").  No GPU, no compiler involved."""
    if D is None:
        raise TypeError("trace_potential needs the dimension D")
    D = int(D)
    if isinstance(potential, TracedGradient):
        potential = potential.fn
    if isinstance(gradient, TracedGradient):
        if potential is None and density is None:
            potential = gradient.fn
        gradient = None
    if potential is None and density is None and gradient is None:
        raise TypeError("nothing to trace")
    tr, q = _symbolic_q(D)
    root = None
    if potential is not None:
        root = _as_scalar(tr, _run(potential, q, "potential"), "potential")
    elif density is not None:
        root = _unary("neg", _unary("log", _as_scalar(tr, _run(density, q, "density"), "density")))
    grads = None
    if gradient is not None:
        g = _run(gradient, q, "gradient")
        ga = np.asarray(_obj(g), dtype=object).ravel()
        if ga.shape != (D,):
            raise TraceError(f"gradient must return ({D},) values, got shape {np.shape(_obj(g))}")
        grads = [v if isinstance(v, Sym) else tr.const(v) for v in ga]
    plan = None
    if root is None:
        # only a gradient (Integrator(ensemble, h, T, gradient)): a LINEAR gradient with a symmetric matrix
        # has the quadratic potential 0.5 q^T A q + b.q; any other one gets a potential that reads NaN
        plan = None if prefer == "source" else _plan_from_linear_gradient(grads, D)
        if plan is None:
            root = tr.const(float("nan"))
    elif prefer != "source":
        plan = _plan_from_gauss_node(root)
        if plan is None and grads is None and not any(n.op in ("gausslog", "gausspdf") for n in _topo([root])):
            poly = _quadratic(root, D)
            if poly is not None:
                plan = _plan_from_quadratic(poly, D)
    if plan is None:
        root = _expand_gauss(root)
        ops = tr.count
        rolled = None
        if grads is None and ops >= ROLL_MIN_OPS:
            rest, groups = _roll_sums(root)
            if groups:
                rolled = generate_source_rolled(rest, groups, D)
        if rolled is not None:
            plan = {"kind": "source", "source": rolled[0], "params": rolled[1], "operations": ops,
                    "rolled_terms": int(_builtin_sum(len(g["table"]) for g in groups))}
        else:
            if grads is None:
                grads = _gradient_nodes(root, D)
            plan = {"kind": "source", "source": generate_source(root, grads), "operations": ops}
    plan["D"] = D
    return plan


def build_plan(plan, dtype="float64", device=None):
    """The potential descriptor of a plan (this is where the GPU comes in: handle creation, and hipcc for
    generated source)."""
    from .potential import GaussianDense, GaussianDiag, Harmonic
    kind = plan["kind"]
    if kind == "harmonic":
        return Harmonic(plan["springConsts"], dtype=dtype, device=device)
    if kind == "gauss_diag":
        return GaussianDiag(plan["mean"], prec=plan["prec"], const=plan["const"], dtype=dtype, device=device)
    if kind == "gauss_dense":
        mean = plan["mean"] if np.any(plan["mean"]) else None
        if "cov" in plan:   # -multivariate_normal.logpdf(q, mean, cov): as the descriptor itself builds it
            pot = GaussianDense(mean, cov=plan["cov"], dtype=dtype, device=device)
            if plan.get("const_extra"):
                const = pot.const + plan["const_extra"]
                pot.close()
                pot = GaussianDense(mean, cov=plan["cov"], const=const, dtype=dtype, device=device)
            return pot
        return GaussianDense(mean, precision=plan["precision"], const=plan["const"], dtype=dtype, device=device,
                             symmetrize=False)
    from .custom import CustomPotential
    pot = CustomPotential(plan["D"], plan["source"], plan.get("params", ()), dtype=dtype, device=device)
    pot.traced_source = plan["source"]
    return pot


def trace_potential(potential=None, D=None, gradient=None, density=None, dtype="float64", device=None,
                    prefer=None):
    """Descriptor for a Python callable `potential(q) -> scalar` (or `density(q)`: potential =
    -log density, src/HMC.py:75-84) over D dimensions; `gradient(q) -> (D,)` optional (default: the
    trace is differentiated, the reference's `grad(self.potential)`, src/HMC.py:57-60).

    prefer=None   quadratic forms and multivariate_normal map to the built-in descriptors, the rest to
                  generated source;  prefer="source"  always generates source (tests use it to push a
                  Gaussian through the generic path)."""
    key = (potential.fn if isinstance(potential, TracedGradient) else potential,
           gradient.fn if isinstance(gradient, TracedGradient) else gradient, isinstance(gradient, TracedGradient),
           density, None if D is None else int(D), str(np.dtype(dtype)), device, prefer)
    try:
        hit = _CACHE.get(key)
    except TypeError:
        key, hit = None, None
    if hit is not None and hit._handle:
        return hit
    pot = build_plan(plan_potential(potential, D, gradient, density, prefer), dtype, device)
    pot.traced_from = next(f for f in (potential, density, gradient) if f is not None)
    if key is not None:
        if len(_CACHE) > 64:
            _CACHE.clear()
        _CACHE[key] = pot
    return pot


def _gauss_params(n):
    mean = np.frombuffer(n.val[0], dtype=np.float64).copy()
    D = mean.size
    cov = np.frombuffer(n.val[1], dtype=np.float64).reshape(D, D).copy()
    return mean, cov


def _plan_from_gauss_node(root):
    """root == -logpdf(q; mean, cov) (+ const)  ->  GaussianDense(mean, cov=cov) as the descriptor builds it"""
    def match(n):   # (gausslog node, c) with n == -gausslog + c, or None
        if n.op == "neg" and n.args[0].op == "gausslog":
            return n.args[0], 0.0
        if n.op == "sub" and n.args[0].op == "const" and n.args[1].op == "gausslog":
            return n.args[1], _cval(n.args[0])
        if n.op in ("add", "sub") and n.args[1].op == "const":
            m = match(n.args[0])
            return m and (m[0], m[1] + _cval(n.args[1]) * (1.0 if n.op == "add" else -1.0))
        if n.op == "add" and n.args[0].op == "const":
            m = match(n.args[1])
            return m and (m[0], m[1] + _cval(n.args[0]))
        return None
    m = match(root)
    if m is None:
        return None
    mean, cov = _gauss_params(m[0])
    return {"kind": "gauss_dense", "mean": mean, "cov": cov, "const_extra": m[1]}


def _expand_gauss(root):
    """multivariate_normal nodes inside a LARGER expression: written out as their quadratic form"""
    nodes = [n for n in _topo([root]) if n.op in ("gausslog", "gausspdf")]
    if not nodes:
        return root
    tr = root.tr
    D = tr.D
    q = [tr.node("in", (), j) for j in range(D)]
    repl = {}
    for n in nodes:
        mean, cov = _gauss_params(n)
        P = np.linalg.inv(cov)
        P = 0.5 * (P + P.T)
        x = [_binary("sub", q[j], tr.const(mean[j])) for j in range(D)]
        Px = [_dot1(tr, P[i], x) for i in range(D)]
        quad = _dot1(tr, x, Px)
        _, logdet = np.linalg.slogdet(cov)
        lp = _binary("sub", _binary("mul", tr.const(-0.5), quad), tr.const(0.5 * (D * math.log(2 * math.pi) + logdet)))
        repl[n.id] = lp if n.op == "gausslog" else _unary("exp", lp)
    return _rebuild(root, repl)


def _rebuild(root, repl):
    new = dict(repl)
    for n in _topo([root]):
        if n.id in new:
            continue
        if not n.args:
            new[n.id] = n
            continue
        args = [new[a.id] for a in n.args]
        if all(x is y for x, y in zip(args, n.args)):
            new[n.id] = n
        elif n.op in ("add", "sub", "mul", "div"):
            new[n.id] = _binary(n.op, *args)
        elif n.op == "addn":
            new[n.id] = _addn(args)
        elif n.op == "select":
            new[n.id] = _select(Cond(n.val, args[0], args[1]), args[2], args[3])
        elif n.op in ("powi", "pow"):
            new[n.id] = n.tr.node(n.op, tuple(args), n.val)
        else:
            new[n.id] = _unary(n.op, args[0])
    return new[root.id]


def _plan_from_linear_gradient(grads, D):
    A, b = np.zeros((D, D)), np.zeros(D)
    for i, g in enumerate(grads):
        poly = _quadratic(g, D)
        if poly is None or poly[2]:
            return None
        b[i] = poly[0]
        for j, v in poly[1].items():
            A[i, j] = v
    if not np.array_equal(A, A.T):
        return None   # not the gradient of a potential
    quad_A = {}
    for i in range(D):
        for j in range(i, D):
            if A[i, j] != 0.0:
                quad_A[(i, j)] = 0.5 * A[i, i] if i == j else A[i, j]
    return _plan_from_quadratic((0.0, {j: v for j, v in enumerate(b) if v != 0.0}, quad_A), D)
