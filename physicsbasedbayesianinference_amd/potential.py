"""Potentials -- host-side mirror of the reference's `src/potential.py` for the HMC path.

In the reference a "potential" is any Python callable `q(D,) -> scalar`, with a
gradient callable `(D,) -> (D,)` obtained from `jax.grad` (src/HMC.py:52-60,
src/integrator.py:73).  A GPU kernel cannot call back into Python, so the closed
forms the reference exercises are *descriptor objects* here: each owns an opaque
`pbbi_potential` handle (include/pbbi.h) and is also host-callable with the
reference's calling convention -- `pot(q)` / `pot.gradient(q)` accept `(D,)` or
`(D, N)` arrays -- by running the HIP evaluation kernel.  Nothing is computed
on the CPU; without the HIP library / a GPU these calls raise.

Mapping from the reference's usage:
  harmonicPotentialND(q, k)                       -> Harmonic(k)         (src/potential.py:18-27)
  -multivariate_normal.logpdf(q, mean, cov)       -> GaussianDense(mean, cov=cov)
                                                     (src/tests/test_HMC.py:49,125)
  -log(exp(-0.5*|x|^2)/sqrt(2 pi))                -> StandardGaussian(D, const=...)
                                                     (src/tests/test_HMC.py:27-33)
  Rosenbrock (BASELINE config 3; not in the reference) -> Rosenbrock(D, a, b, s)
The N-body gravity helpers (src/potential.py:30-138) are not part of the HMC
hot path and are out of scope.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._device import as_device, default_device, empty, stream_ptr, to_numpy, torch_dtype

__all__ = ["Potential", "Harmonic", "GaussianDiag", "StandardGaussian", "GaussianDense",
           "Rosenbrock", "harmonicPotentialND", "noPotential", "linear_regression_posterior"]


def _dptr(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_double)) if arr is not None else None


class Potential:
    """Base class: owns a pbbi_potential handle on one device."""

    kind = "abstract"

    def __init__(self, D, dtype="float64", device=None):
        self.numDimensions = int(D)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype("float64"), np.dtype("float32")):
            raise TypeError("dtype must be float64 or float32")
        self.device = default_device() if device is None else int(device)
        self._handle = C.c_void_p()

    # -- handle management
    @property
    def _dt(self):
        return _lib.F64 if self.dtype == np.dtype("float64") else _lib.F32

    @property
    def handle(self):
        if not self._handle:
            raise RuntimeError("potential handle was destroyed")
        return self._handle

    def close(self):
        if getattr(self, "_handle", None):
            _lib.load().pbbi_potential_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- reference calling convention: potential(q) and gradient(q)
    def _eval(self, q, want_U, want_grad):
        qh = np.asarray(q, dtype=self.dtype)
        single = qh.ndim == 1
        if single:
            qh = qh.reshape(-1, 1)
        if qh.ndim != 2 or qh.shape[0] != self.numDimensions:
            raise ValueError(f"q must be ({self.numDimensions},) or ({self.numDimensions}, N)")
        N = qh.shape[1]
        qd = as_device(qh, self.device, self.dtype)
        U = empty((N,), self.dtype, self.device) if want_U else None
        g = empty((self.numDimensions, N), self.dtype, self.device) if want_grad else None
        _lib.call("pbbi_potential_eval", self.handle, qd.data_ptr(), N, N,
                  U.data_ptr() if want_U else None, g.data_ptr() if want_grad else None,
                  stream_ptr(self.device))
        Uh = to_numpy(U) if want_U else None
        gh = to_numpy(g) if want_grad else None
        if single:
            Uh = Uh[0] if want_U else None
            gh = gh[:, 0] if want_grad else None
        return Uh, gh

    def __call__(self, q):
        return self._eval(q, True, False)[0]

    def gradient(self, q):
        return self._eval(q, False, True)[1]

    def value_and_gradient(self, q):
        return self._eval(q, True, True)

    def density(self, q):
        """exp(-U(q)): lets `HMC(ens, T, h, pot.density)` work like the reference's
        `density=` argument (src/HMC.py:75-84 takes -log of it again)."""
        return np.exp(-self(q))


class Harmonic(Potential):
    """U(q) = 0.5 * dot(springConsts, q**2)   (src/potential.py:18-27)."""

    kind = "harmonic"

    def __init__(self, springConsts, dtype="float64", device=None):
        k = np.ascontiguousarray(springConsts, dtype=np.float64).ravel()
        super().__init__(k.size, dtype, device)
        self.springConsts = k
        _lib.call("pbbi_potential_create_harmonic", k.size, _dptr(k), self._dt, self.device,
                  C.byref(self._handle))


class GaussianDiag(Potential):
    """U(q) = 0.5 * sum(prec * (q - mean)**2) + const; give `var` or `prec`."""

    kind = "gauss_diag"

    def __init__(self, mean, var=None, prec=None, const=None, dtype="float64", device=None):
        mean = np.ascontiguousarray(mean, dtype=np.float64).ravel()
        if (var is None) == (prec is None):
            raise ValueError("give exactly one of var= / prec=")
        if prec is None:
            prec = 1.0 / np.asarray(var, dtype=np.float64)
        prec = np.ascontiguousarray(np.broadcast_to(np.asarray(prec, np.float64), mean.shape))
        super().__init__(mean.size, dtype, device)
        if const is None:  # -log of the normalised density
            const = 0.5 * (mean.size * np.log(2 * np.pi) - np.sum(np.log(prec)))
        self.mean, self.prec, self.const = mean, prec, float(const)
        _lib.call("pbbi_potential_create_gauss_diag", mean.size, _dptr(mean), _dptr(prec),
                  self.const, self._dt, self.device, C.byref(self._handle))


class StandardGaussian(GaussianDiag):
    """U(q) = 0.5*|q|^2 + const (BASELINE config 1: the 1-D standard Gaussian)."""

    def __init__(self, D, const=0.0, dtype="float64", device=None):
        super().__init__(np.zeros(int(D)), prec=np.ones(int(D)), const=const, dtype=dtype,
                         device=device)


class GaussianDense(Potential):
    """U(q) = 0.5 (q-mean)^T P (q-mean) + const, grad = P (q-mean).

    Give `precision` (P) or `cov` (P = inv(cov)); P is symmetrised as
    0.5*(P + P^T).  With `const=None` the constant is 0.5*log det(2 pi cov), i.e.
    U = -multivariate_normal.logpdf(q, mean, cov) as in src/tests/test_HMC.py:49,125.
    D <= 128 in float64 runs on the register-resident MFMA kernels; larger D (and float32)
    on the streaming MFMA-GEMM path (one fused GEMM per integrator step).
    """

    kind = "gauss_dense"

    def __init__(self, mean, precision=None, cov=None, const=None, dtype="float64", device=None,
                 symmetrize=True):
        if (precision is None) == (cov is None):
            raise ValueError("give exactly one of precision= / cov=")
        if precision is None:
            precision = np.linalg.inv(np.asarray(cov, dtype=np.float64))
        P = np.array(precision, dtype=np.float64)
        if P.ndim != 2 or P.shape[0] != P.shape[1]:
            raise ValueError("precision must be square")
        if symmetrize:
            P = 0.5 * (P + P.T)
        P = np.ascontiguousarray(P)
        D = P.shape[0]
        mean = np.zeros(D) if mean is None else np.ascontiguousarray(mean, dtype=np.float64).ravel()
        if mean.size != D:
            raise ValueError("mean and precision sizes differ")
        super().__init__(D, dtype, device)
        if const is None:
            sign, logdet = np.linalg.slogdet(P)
            const = 0.5 * (D * np.log(2 * np.pi) - logdet)
        self.mean, self.precision, self.const = mean, P, float(const)
        _lib.call("pbbi_potential_create_gauss_dense", D, _dptr(mean), _dptr(P), self.const,
                  self._dt, self.device, C.byref(self._handle))


def linear_regression_posterior(X, y, noise_var, prior_precision=0.0, prior_mean=None, **kw):
    """Model -> potential for the conjugate Bayesian linear model (the
    free-fall regression of the reference's notes): y = X w + eps, eps ~ N(0, noise_var),
    w ~ N(prior_mean, prior_precision^-1).  The posterior is Gaussian, so its potential
    -log p(w | X, y) is a GaussianDense descriptor with
        P = X^T X / noise_var + Lambda,   mean = P^-1 (X^T y / noise_var + Lambda prior_mean)
    (constant dropped).  Parameter set-up on the host; sampling runs on the dense kernels."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).ravel()
    M, D = X.shape
    if y.size != M:
        raise ValueError("X has %d rows, y has %d entries" % (M, y.size))
    Lam = np.asarray(prior_precision, dtype=np.float64)
    Lam = np.eye(D) * Lam if Lam.ndim == 0 else (np.diag(Lam) if Lam.ndim == 1 else Lam)
    m0 = np.zeros(D) if prior_mean is None else np.asarray(prior_mean, dtype=np.float64).ravel()
    P = X.T @ X / noise_var + Lam
    mean = np.linalg.solve(P, X.T @ y / noise_var + Lam @ m0)
    return GaussianDense(mean, precision=P, const=0.0, **kw)


class Rosenbrock(Potential):
    """U(q) = sum_{i<D-1} [b (q_{i+1} - q_i^2)^2 + (a - q_i)^2] * (1/s).

    Defined by this build (BASELINE config 3): a=1, b=100, s=20.
    """

    kind = "rosenbrock"

    def __init__(self, D, a=1.0, b=100.0, s=20.0, dtype="float64", device=None):
        super().__init__(D, dtype, device)
        self.a, self.b, self.s = float(a), float(b), float(s)
        _lib.call("pbbi_potential_create_rosenbrock", int(D), self.a, self.b, self.s, self._dt,
                  self.device, C.byref(self._handle))


_HARMONIC_CACHE = {}


def harmonicPotentialND(q, springConsts):
    """Drop-in for src/potential.py:18-27: 0.5 * dot(springConsts, q**2) for q of shape
    (D,) or (D, N), evaluated by the HIP kernel.  On a TRACED q (a lambda around this function handed to
    HMC / Integrator / grad, src/tests/test_integrator_harmonic.py:23-24) it records the same expression."""
    from . import trace
    if trace.is_symbolic(q):
        return 0.5 * trace.dot(np.asarray(springConsts, dtype=np.float64), q ** 2)
    k = np.ascontiguousarray(springConsts, dtype=np.float64).ravel()
    key = (k.tobytes(), default_device())
    pot = _HARMONIC_CACHE.get(key)
    if pot is None:
        if len(_HARMONIC_CACHE) > 64:
            _HARMONIC_CACHE.clear()
        pot = _HARMONIC_CACHE[key] = Harmonic(k)
    return pot(q)


def noPotential(q):
    """src/potential.py:141-142."""
    return 0
