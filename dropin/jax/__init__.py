"""Bare-name shim for the reference's `jax` imports (`from jax import grad`, `import jax.numpy as jnp`,
`from jax.scipy.stats import multivariate_normal`, `jax.config.update(...)`; src/HMC.py:10-15,
src/potential.py:12, src/tests/test_HMC.py:13-22): with `dropin/` on sys.path they resolve to the
traceable array namespace of this build (physicsbasedbayesianinference_amd/trace.py), so the
reference's lambdas are traced into HIP kernels instead of being called per chain.  This is NOT JAX: it
holds the symbols the reference touches, nothing else."""
from physicsbasedbayesianinference_amd.trace import grad  # noqa: F401
from . import numpy, scipy  # noqa: F401


class _Config:
    def update(self, *a, **k):
        """jax.config.update("jax_enable_x64", True): float64 is this build's default already."""


config = _Config()
