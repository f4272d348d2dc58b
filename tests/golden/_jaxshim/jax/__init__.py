"""Stand-in for the third-party `jax` package (absent offline).

Build-container tooling only: it lets `tests/golden/gen_golden.py` import the
reference's *unmodified* `src/*.py` (which `import jax` at module top) so that
golden vectors can be produced from the reference's own arithmetic.  It holds
no reference code.  `jax.numpy` re-exports NumPy; `jax.grad` refuses to run so
that every gradient used for a fixture is an explicit NumPy callable.
Never imported by the product, the tests or the GPU box.
"""
import sys as _sys
import types as _types

import numpy as _np

numpy = _types.ModuleType("jax.numpy")
for _k in dir(_np):
    if not _k.startswith("__"):
        setattr(numpy, _k, getattr(_np, _k))
_sys.modules["jax.numpy"] = numpy


class _Cfg:
    def update(self, *a, **k):
        pass


config = _Cfg()


def grad(f):
    def _g(*a, **k):
        raise NotImplementedError(
            "autodiff unavailable offline; pass gradient= explicitly"
        )

    return _g
