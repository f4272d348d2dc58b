"""User-defined potentials: the GPU counterpart of the reference's arbitrary callables.

The reference takes ANY Python callable `q(D,) -> scalar` as potential and differentiates it
with `jax.grad` (src/HMC.py:52-60), or takes an explicit `gradient=` callable (src/HMC.py:35-60,
src/integrator.py:36-59), and calls them once per chain per step from the Python loop.  A HIP
kernel cannot call back into Python, so here the two functions are written as a few lines of
C++ and are compiled (hipcc, gfx950) INTO the ensemble-HMC kernels of a plugin shared object --
inlined, one chain per lane, no indirect calls (csrc/pbbi_custom.h).  Everything downstream
(`HMC`, `Leapfrog`, `StormerVerlet`, `getWeights`, sharding, both RNG modes) works on a
`CustomPotential` exactly as on the built-in descriptors.

Source contract (`T` is the scalar type of the build, float64 unless dtype="float32"; `prm` are
the `params` given to the constructor -- model constants or a whole data set; q[j] / g[j] address
element j of ONE chain):

    template <class Q>
    PBBI_FN T potential(const Q& q, int D, const T* prm) {
        T s = 0;
        for (int j = 0; j < D; ++j) s += T(0.25) * q[j] * q[j] * q[j] * q[j];
        return s;
    }
    template <class Q, class G>
    PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {
        for (int j = 0; j < D; ++j) g[j] = q[j] * q[j] * q[j];     // every g[0..D) must be written
    }

Use unqualified math calls (`exp`, `log`, `log1p`, `sqrt`, `tanh`, `fma`, ...): the same source is
compiled for the host by the test oracle (oracle/oracle.py::pot_custom).

Automatic gradient (the reference's default, `grad(self.potential)`, src/HMC.py:57-60): a source
WITHOUT a `gradient` function gets one by forward-mode differentiation -- `potential` is
instantiated on dual numbers (csrc/pbbi_autodiff.h), one pass per coordinate.  Such a potential
must be written on the element type of its view, `pbbi_scalar<Q>`, instead of `T`:

    template <class Q>
    PBBI_FN pbbi_scalar<Q> potential(const Q& q, int D, const T* prm) {
        pbbi_scalar<Q> s = 0;
        for (int j = 0; j < D; ++j) s += log1p(exp(-q[j])) + T(0.5) * prm[0] * q[j] * q[j];
        return s;
    }

It costs D potential evaluations per gradient (a hand-written gradient stays the fast path and can
be checked against it: `CustomPotential.check_gradient`, or build both and compare).  No CPU
fallback: without hipcc or a GPU the constructor raises.
"""
import ctypes as C
import hashlib
import os
import re
import subprocess

import numpy as np

from . import _lib
from .potential import Potential, _dptr

__all__ = ["CustomPotential", "compile_plugin", "plugin_source", "complete_source", "has_gradient",
           "coin_toss_posterior",
           "logistic_regression_posterior", "COIN_TOSS_SOURCE", "LOGISTIC_REGRESSION_SOURCE",
           "EXAMPLE_SOURCE"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
_CACHE = os.path.join(_HERE, "_plugins")
_DEPS = ("pbbi_custom.h", "pbbi_internal.h", "pbbi_rng.h")


_AD_GRADIENT = """
// no gradient in the source: forward-mode automatic differentiation (csrc/pbbi_autodiff.h)
template <class Q, class G>
PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {
    for (int k = 0; k < D; ++k) g[k] = potential(pbbi_ad::Seeded<Q>{q, k}, D, prm).d;
}
"""


def has_gradient(source):
    return re.search(r"\bvoid\s+gradient\s*\(", source) is not None


def complete_source(source):
    """What goes inside `namespace user` of a generated translation unit (device plugin and the test
    oracle's host build alike): the dual-number support text, the user's source, and -- when the
    source states no `gradient` -- the automatic one."""
    with open(os.path.join(_CSRC, "pbbi_autodiff.h")) as f:
        ad = f.read()
    return ad + "\n" + source + ("" if has_gradient(source) else _AD_GRADIENT)


def plugin_source(source, dtype="float64", D=None):
    """The translation unit hipcc sees for one user source (and one dimension: PBBI_D lets small
    chains run from registers, csrc/pbbi_custom.h)."""
    source = complete_source(source)
    ctype = {"float64": "double", "float32": "float"}[str(np.dtype(dtype))]
    return ((f'#define PBBI_D {int(D)}\n' if D else '') +
            '#include <hip/hip_runtime.h>\n#include <type_traits>\n#include <utility>\n'
            '#include "pbbi_internal.h"\n#include "pbbi_rng.h"\n'
            f'using T = {ctype};\n#define PBBI_FN __device__ __forceinline__\n'
            'namespace user {\n' + source + '\n}  // namespace user\n'
            '#include "pbbi_custom.h"\n')


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC): user-defined potentials are compiled for "
                       "gfx950 at construction and there is no CPU fallback")


def compile_plugin(source, dtype="float64", verbose=False, D=None):
    """Build (or find in the in-tree cache) the plugin .so of a user source; needs hipcc, not a
    GPU.  The cache key covers the source, the dtype, the dimension and the kernel headers."""
    tu = plugin_source(source, dtype, D)
    h = hashlib.sha256(tu.encode())
    for dep in _DEPS:
        with open(os.path.join(_CSRC, dep), "rb") as f:
            h.update(f.read())
    with open(os.path.join(_INCLUDE, "pbbi.h"), "rb") as f:
        h.update(f.read())
    key = h.hexdigest()[:20]
    os.makedirs(_CACHE, exist_ok=True)
    so = os.path.join(_CACHE, f"pot_{key}.so")
    if os.path.exists(so):
        return so
    src = os.path.join(_CACHE, f"pot_{key}.hip")
    with open(src, "w") as f:
        f.write(tu)
    tmp = so + f".tmp{os.getpid()}"
    # (the -mllvm switches: the register kernels loop over the iterations of a fused run around a fully
    #  unrolled body; left alone hipcc hoists that body's invariants out of the loop and spills them --
    #  csrc/Makefile, FLAGS_kernels_lane2)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-mllvm", "-disable-machine-licm", "-mllvm", "-sink-insts-to-avoid-spills",
           "-mllvm", "-amdgpu-use-amdgpu-trackers=1", f"-I{_CSRC}", f"-I{_INCLUDE}", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on the user-defined potential:\n" + res.stderr[-4000:])
    os.replace(tmp, so)  # atomic: concurrent ranks may build the same plugin
    return so


class CustomPotential(Potential):
    """A potential given as C++ source (module docstring); `params` is a flat float64 array."""

    kind = "custom"

    def __init__(self, D, source, params=(), dtype="float64", device=None, verbose=False):
        super().__init__(D, dtype, device)
        self.source = source
        self.autodiff = not has_gradient(source)   # gradient by dual numbers (csrc/pbbi_autodiff.h)
        self.params = np.ascontiguousarray(params, dtype=np.float64).ravel()
        self.plugin_path = compile_plugin(source, self.dtype, verbose=verbose, D=int(D))
        _lib.call("pbbi_potential_create_custom", self.plugin_path.encode(), int(D),
                  _dptr(self.params) if self.params.size else None, int(self.params.size),
                  self._dt, self.device, C.byref(self._handle))

    def check_gradient(self, q, eps=1e-6):
        """Max relative deviation between gradient(q) and central differences of potential(q),
        both evaluated by the HIP kernels; q is (D,) or (D, N)."""
        q = np.asarray(q, dtype=np.float64)
        q2 = q.reshape(self.numDimensions, -1)
        g = self.gradient(q2)
        fd = np.empty_like(g)
        for j in range(self.numDimensions):
            dq = np.zeros_like(q2)
            dq[j] = eps
            fd[j] = (self(q2 + dq) - self(q2 - dq)) / (2 * eps)
        return float(np.max(np.abs(g - fd)) / max(1.0, float(np.max(np.abs(fd)))))


# ---------------------------------------------------------------------------------------------
# A first library of likelihoods on top of the source mechanism (the model ->
# potential boundary; the reference shows the pattern with NumPyro's log_density in
# samples/NumpyroExamples/CoinToss/CoinTossExample.py:75-107).

EXAMPLE_SOURCE = """
template <class Q>
PBBI_FN T potential(const Q& q, int D, const T* prm) {
    T s = 0;
    for (int j = 0; j < D; ++j) s += ((q[j] * q[j]) * (q[j] * q[j]));
    return (T(0.25) * prm[0]) * s;
}
template <class Q, class G>
PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {
    for (int j = 0; j < D; ++j) g[j] = prm[0] * ((q[j] * q[j]) * q[j]);
}
"""

# Coin toss: theta ~ Beta(a, b), k heads in n tosses, sampled in x = logit(theta) (unconstrained;
# the Jacobian theta (1 - theta) is included):  U(x) = A softplus(-x) + B softplus(x),
# A = k + a, B = n - k + b.  D independent coins, each with its own data: prm = [A_0, B_0, A_1, B_1, ...]
# (the reference's example tosses TWO coins c1, c2 with biases p1, p2: samples/NumpyroExamples/CoinToss).
COIN_TOSS_SOURCE = """
PBBI_FN T softplus(T z) { return (z > 0 ? z : T(0)) + log1p(exp(-fabs(z))); }
template <class Q>
PBBI_FN T potential(const Q& q, int D, const T* prm) {
    T s = 0;
    for (int j = 0; j < D; ++j) s += prm[2 * j] * softplus(-q[j]) + prm[2 * j + 1] * softplus(q[j]);
    return s;
}
template <class Q, class G>
PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {
    for (int j = 0; j < D; ++j) {
        const T sig = T(1) / (T(1) + exp(-q[j]));
        g[j] = prm[2 * j + 1] * sig - prm[2 * j] * (T(1) - sig);
    }
}
"""

# Bayesian logistic regression: prm = [M, X (M x D row-major), y in {0,1} (M), prior precision];
# U(w) = sum_i softplus(x_i.w) - y_i x_i.w + 0.5 lam |w|^2
LOGISTIC_REGRESSION_SOURCE = """
template <class Q>
PBBI_FN T potential(const Q& q, int D, const T* prm) {
    const int M = (int)prm[0];
    const T* X = prm + 1;
    const T* y = X + (long)M * D;
    const T lam = y[M];
    T s = 0;
    for (int i = 0; i < M; ++i) {
        T z = 0;
        for (int j = 0; j < D; ++j) z += X[i * D + j] * q[j];
        s += ((z > 0 ? z : T(0)) + log1p(exp(-fabs(z)))) - y[i] * z;
    }
    T r = 0;
    for (int j = 0; j < D; ++j) r += q[j] * q[j];
    return s + (T(0.5) * lam) * r;
}
template <class Q, class G>
PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {
    const int M = (int)prm[0];
    const T* X = prm + 1;
    const T* y = X + (long)M * D;
    const T lam = y[M];
    for (int j = 0; j < D; ++j) g[j] = lam * q[j];
    for (int i = 0; i < M; ++i) {
        T z = 0;
        for (int j = 0; j < D; ++j) z += X[i * D + j] * q[j];
        const T w = T(1) / (T(1) + exp(-z)) - y[i];
        for (int j = 0; j < D; ++j) g[j] += w * X[i * D + j];
    }
}
"""


def coin_toss_posterior(heads, tosses, a=1.0, b=1.0, D=None, **kw):
    """Posterior of the biases theta_j ~ Beta(a, b) of independent coins after heads[j] in tosses[j],
    as a potential in x = logit(theta); theta_j = 1/(1+exp(-x_j)) of the samples is
    Beta(heads[j]+a, tosses[j]-heads[j]+b).  `heads` / `tosses` / `a` / `b` are scalars or one value per
    coin -- the reference's example (samples/NumpyroExamples/CoinToss/CoinToss.py:18-22) is
    coin_toss_posterior([k1, k2], [n1, n2]) with its uniform prior a = b = 1; D (default: the number of
    entries) repeats scalar data over D coins."""
    k, n, a, b = (np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in (heads, tosses, a, b))
    D = int(max(k.size, n.size, a.size, b.size) if D is None else D)
    k, n, a, b = (np.broadcast_to(x, (D,)) for x in (k, n, a, b))
    if np.any(k > n) or np.any(k < 0):
        raise ValueError("need 0 <= heads <= tosses for every coin")
    prm = np.stack([k + a, n - k + b], axis=1).ravel()
    return CustomPotential(D, COIN_TOSS_SOURCE, prm, **kw)


def logistic_regression_params(X, y, prior_precision=1.0):
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).ravel()
    if X.ndim != 2 or y.size != X.shape[0]:
        raise ValueError("X must be (M, D) and y (M,)")
    return np.concatenate([[float(X.shape[0])], X.ravel(), y, [float(prior_precision)]])


def logistic_regression_posterior(X, y, prior_precision=1.0, **kw):
    """-log posterior of the weights of y_i ~ Bernoulli(sigmoid(x_i . w)), w ~ N(0, I/prior_precision);
    the data set rides in the potential's parameter array."""
    X = np.asarray(X, dtype=np.float64)
    return CustomPotential(X.shape[1], LOGISTIC_REGRESSION_SOURCE,
                           logistic_regression_params(X, y, prior_precision), **kw)
