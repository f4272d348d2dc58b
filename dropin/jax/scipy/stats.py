"""`jax.scipy.stats` of the dropin shim: multivariate_normal / norm of the traceable namespace."""
from physicsbasedbayesianinference_amd.trace import multivariate_normal, norm  # noqa: F401
