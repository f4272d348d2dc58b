"""Bare-name shim: put this directory on sys.path and the reference's own import lines
(`from ensemble import ...`, cf. its src/tests/*.py) resolve to the MI355X implementation."""
from physicsbasedbayesianinference_amd.ensemble import *  # noqa: F401,F403
from physicsbasedbayesianinference_amd.ensemble import __all__  # noqa: F401
