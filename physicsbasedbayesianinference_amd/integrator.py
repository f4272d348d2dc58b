"""Integrators -- drop-in for the reference's `src/integrator.py` on the gradient path.

`Integrator(ensemble, stepSize, finalTime, gradient)`, `Leapfrog.integrate()` and
`StormerVerlet.integrate()` keep the reference's names, argument order, attributes
(`q, p, v, mass, numParticles, stepSize, finalTime, numSteps, gradient`) and in-place
aliasing semantics (`integrator.q is ensemble.q`; `integrate()` mutates those arrays
and returns the same objects, src/integrator.py:40-42,123).  The per-particle Python
loops of src/integrator.py:105-120 and :142-163 are replaced by one fused HIP kernel
launch over the whole ensemble (`pbbi_leapfrog` / `pbbi_stormer_verlet`).

`gradient` comes from a potential descriptor (`pot.gradient`, or the descriptor itself, including
custom.CustomPotential for user-written potentials) or is a Python callable written on the traceable
array namespace (`trace.py`: `grad(potential)` as in src/tests/test_integrator_harmonic.py:24, or a
`(D,) -> (D,)` function): the callable is traced once and its arithmetic compiled into the kernels.
A callable that cannot be traced is rejected with a TypeError -- there is no host fallback.  `gradient=None` selects the reference's
N-body gravity mode (src/integrator.py:57-59), which is not an HMC path and is out of
scope: NotImplementedError.
"""
import numpy as np

from . import _lib
from ._device import as_device, empty, stream_ptr, to_numpy
from .potential import Potential

__all__ = ["Integrator", "Leapfrog", "StormerVerlet", "resolve_potential"]


def resolve_potential(fn, what="gradient", D=None, potential=None):
    """Descriptor behind a `potential=` / `gradient=` / `density=` argument.  Descriptors and their bound
    methods are returned as they are; any other callable is traced on a symbolic (D,) position
    (trace.trace_potential: src/HMC.py:52-60's lambdas without a C++ string and without a CPU path) --
    `potential` optionally names the potential callable a `gradient=` callable belongs to."""
    if isinstance(fn, Potential):
        return fn
    owner = getattr(fn, "__self__", None)
    if isinstance(owner, Potential):
        return owner
    if callable(fn) and D is not None:
        from . import trace
        kw = {"potential": fn} if what == "potential" else {"density": fn} if what == "density" else \
            {"gradient": fn, "potential": potential}
        return trace.trace_potential(D=D, **kw)   # raises TypeError when the callable cannot be traced
    raise TypeError(
        f"{what} must be a potential descriptor of physicsbasedbayesianinference_amd.potential "
        f"(e.g. GaussianDense(mean, cov=cov), Harmonic(k), Rosenbrock(D)) or one of its bound "
        f"methods; got {fn!r}.  Arbitrary Python callables cannot execute inside the HIP "
        f"kernels and this package has no CPU fallback: state the function as C++ source with "
        f"custom.CustomPotential(D, source, params) instead.")


def mass_or_none(mass, N, dtype, device):
    """Device mass array, or None when every mass is exactly 1 (division-free fast path)."""
    m = np.asarray(mass, dtype=np.float64)
    if m.shape != (N,):
        raise ValueError(f"ensemble.mass must have shape ({N},)")
    if np.all(m == 1.0):
        return None
    return as_device(m, device, dtype)


class Integrator:
    method_id = None

    def __init__(self, ensemble, stepSize, finalTime, gradient):
        self.ensemble = ensemble
        # aliases, not copies (src/integrator.py:40-42)
        self.q = ensemble.q
        self.p = ensemble.p
        self.mass = ensemble.mass
        self.v = self.p / self.mass  # Integrator.v starts as the initial velocities (:45)
        self.numParticles = ensemble.numParticles
        self.stepSize = stepSize
        self.finalTime = finalTime
        self.numSteps = int(self.finalTime / self.stepSize)  # truncation as in :51
        self.gradient = gradient
        if not gradient:
            raise NotImplementedError(
                "gradient=None selects the reference's N-body gravity simulation "
                "(src/integrator.py:57-59); that is not part of the ensemble-HMC hot path "
                "and is not provided by this build")
        self.potential = resolve_potential(gradient, D=ensemble.numDimensions)

    def getAccel(self, i):
        """-gradient(q[:, i]) / mass[i]  (src/integrator.py:61-73), via the HIP eval kernel."""
        return -self.potential.gradient(self.q[:, i]) / self.mass[i]

    def integrate(self):
        raise NotImplementedError("Integrator superclass doesn't specify integration method")

    def _integrate_on_device(self):
        pot = self.potential
        D, N = self.q.shape
        if D != pot.numDimensions:
            raise ValueError(f"potential has D={pot.numDimensions}, ensemble has D={D}")
        qd = as_device(self.q, pot.device, pot.dtype)
        pd = as_device(self.p, pot.device, pot.dtype)
        vd = empty((D, N), pot.dtype, pot.device)
        md = mass_or_none(self.mass, N, pot.dtype, pot.device)
        _lib.call("pbbi_integrate", pot.handle, self.method_id, qd.data_ptr(), pd.data_ptr(),
                  md.data_ptr() if md is not None else None, vd.data_ptr(), N, N,
                  float(self.stepSize), int(self.numSteps), stream_ptr(pot.device))
        # in place: the arrays stay aliased with the ensemble's
        self.q[...] = to_numpy(qd)
        self.p[...] = to_numpy(pd)
        if self.v.shape != self.q.shape:
            self.v = np.empty_like(self.q)
        self.v[...] = to_numpy(vd)
        return (self.q, self.p)


class Leapfrog(Integrator):
    """Velocity-Verlet form of leapfrog, L+1 gradient evaluations (src/integrator.py:95-123)."""

    method_id = _lib.LEAPFROG

    def integrate(self):
        return self._integrate_on_device()


class StormerVerlet(Integrator):
    """Two-step position Verlet: L+1 position steps and a backward-difference velocity,
    exactly as src/integrator.py:127-165."""

    method_id = _lib.STORMER_VERLET

    def integrate(self):
        return self._integrate_on_device()
