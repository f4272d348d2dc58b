"""Bare-name shim: put this directory on sys.path and the reference's own import lines
(`from integrator import ...`, cf. its src/tests/*.py) resolve to the MI355X implementation."""
from physicsbasedbayesianinference_amd.integrator import *  # noqa: F401,F403
from physicsbasedbayesianinference_amd.integrator import __all__  # noqa: F401
