"""Bare-name shim: put this directory on sys.path and the reference's own import lines
(`from potential import ...`, cf. its src/tests/*.py) resolve to the MI355X implementation."""
from physicsbasedbayesianinference_amd.potential import *  # noqa: F401,F403
from physicsbasedbayesianinference_amd.potential import __all__  # noqa: F401
from physicsbasedbayesianinference_amd.custom import CustomPotential  # noqa: F401  user-written potentials
