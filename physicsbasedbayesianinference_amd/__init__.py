"""MI355X-native ensemble HMC: the hot path of Anton-Le/PhysicsBasedBayesianInference
(leapfrog / Stormer-Verlet integration + potential gradient + Metropolis accept) as
hand-written HIP kernels for gfx950 behind the reference's own Python class API.

    from physicsbasedbayesianinference_amd import Ensemble, HMC, GaussianDense
    ens = Ensemble(numDimensions, numParticles)
    pot = GaussianDense(mean, cov=cov)
    samples, momenta = HMC(ens, 1.0, 0.1, None, potential=pot).getSamples(100, 1/kB, 1.0)

Modules mirror the reference's `src/` files: ensemble, integrator, potential, HMC
(the `dropin/` directory at the repo root re-exports them under those bare names).
"""
from .ensemble import Ensemble
from .potential import (GaussianDense, GaussianDiag, Harmonic, Potential, Rosenbrock,
                        StandardGaussian, harmonicPotentialND, linear_regression_posterior)
from .integrator import Integrator, Leapfrog, StormerVerlet
from .HMC import HMC
from .custom import CustomPotential
from . import trace
from .trace import grad, trace_potential

__all__ = ["Ensemble", "HMC", "Integrator", "Leapfrog", "StormerVerlet", "Potential",
           "Harmonic", "GaussianDiag", "StandardGaussian", "GaussianDense", "Rosenbrock",
           "harmonicPotentialND", "linear_regression_posterior", "CustomPotential", "trace", "grad",
           "trace_potential"]
__version__ = "0.1.0"
