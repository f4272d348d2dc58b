"""`jax.numpy` of the dropin shim: the traceable array namespace (physicsbasedbayesianinference_amd/trace.py)."""
from physicsbasedbayesianinference_amd.trace import *  # noqa: F401,F403
from physicsbasedbayesianinference_amd.trace import linalg, pi, e, inf  # noqa: F401
