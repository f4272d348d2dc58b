"""User-defined potentials: the GPU counterpart of the reference's arbitrary callables.

The reference takes ANY Python callable `q(D,) -> scalar` as potential and differentiates it
with `jax.grad` (src/HMC.py:52-60), or takes an explicit `gradient=` callable (src/HMC.py:35-60,
src/integrator.py:36-59), and calls them once per chain per step from the Python loop.  A HIP
kernel cannot call back into Python, so here the two functions are written as a few lines of
C++ and are compiled (hipcc, gfx950) INTO the ensemble-HMC kernels of a plugin shared object --
inlined, one chain per lane, no indirect calls (csrc/pbbi_custom.h).  Everything downstream
(`HMC`, `Leapfrog`, `StormerVerlet`, `getWeights`, sharding, both RNG modes) works on a
`CustomPotential` exactly as on the built-in descriptors.

Source contract (`T` is the scalar type of the build, float64 unless dtype="float32"; `prm` are
the `params` given to the constructor -- model constants or a whole data set; q[j] / g[j] address
element j of ONE chain):

    template <class Q>
    PBBI_FN T potential(const Q& q, int D, const T* prm) {
        T s = 0;
        for (int j = 0; j < D; ++j) s += T(0.25) * q[j] * q[j] * q[j] * q[j];
        return s;
    }
    template <class Q, class G>
    PBBI_FN void gradient(const Q& q, G& g, int D, const T* prm) {
        for (int j = 0; j < D; ++j) g[j] = q[j] * q[j] * q[j];     // every g[0..D) must be written
    }

Use unqualified math calls (`exp`, `log`, `log1p`, `sqrt`, `tanh`, `fma`, ...): the same source is
compiled for the host by the test oracle (oracle/oracle.py::pot_custom).  There is no autodiff:
the gradient is the user's (check it with `CustomPotential.check_gradient`).  No CPU fallback:
without hipcc or a GPU the constructor raises.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

from . import _lib
from .potential import Potential, _dptr

__all__ = ["CustomPotential", "compile_plugin", "plugin_source"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
_CACHE = os.path.join(_HERE, "_plugins")
_DEPS = ("pbbi_custom.h", "pbbi_internal.h", "pbbi_rng.h")


def plugin_source(source, dtype="float64"):
    """The translation unit hipcc sees for one user source."""
    ctype = {"float64": "double", "float32": "float"}[str(np.dtype(dtype))]
    return ('#include <hip/hip_runtime.h>\n#include <type_traits>\n'
            '#include "pbbi_internal.h"\n#include "pbbi_rng.h"\n'
            f'using T = {ctype};\n#define PBBI_FN __device__ __forceinline__\n'
            'namespace user {\n' + source + '\n}  // namespace user\n'
            '#include "pbbi_custom.h"\n')


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC): user-defined potentials are compiled for "
                       "gfx950 at construction and there is no CPU fallback")


def compile_plugin(source, dtype="float64", verbose=False):
    """Build (or find in the in-tree cache) the plugin .so of a user source; needs hipcc, not a
    GPU.  The cache key covers the source, the dtype and the kernel headers."""
    tu = plugin_source(source, dtype)
    h = hashlib.sha256(tu.encode())
    for dep in _DEPS:
        with open(os.path.join(_CSRC, dep), "rb") as f:
            h.update(f.read())
    with open(os.path.join(_INCLUDE, "pbbi.h"), "rb") as f:
        h.update(f.read())
    key = h.hexdigest()[:20]
    os.makedirs(_CACHE, exist_ok=True)
    so = os.path.join(_CACHE, f"pot_{key}.so")
    if os.path.exists(so):
        return so
    src = os.path.join(_CACHE, f"pot_{key}.hip")
    with open(src, "w") as f:
        f.write(tu)
    tmp = so + f".tmp{os.getpid()}"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", f"-I{_CSRC}", f"-I{_INCLUDE}", src, "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed on the user-defined potential:\n" + res.stderr[-4000:])
    os.replace(tmp, so)  # atomic: concurrent ranks may build the same plugin
    return so


class CustomPotential(Potential):
    """A potential given as C++ source (module docstring); `params` is a flat float64 array."""

    kind = "custom"

    def __init__(self, D, source, params=(), dtype="float64", device=None, verbose=False):
        super().__init__(D, dtype, device)
        self.source = source
        self.params = np.ascontiguousarray(params, dtype=np.float64).ravel()
        self.plugin_path = compile_plugin(source, self.dtype, verbose=verbose)
        _lib.call("pbbi_potential_create_custom", self.plugin_path.encode(), int(D),
                  _dptr(self.params) if self.params.size else None, int(self.params.size),
                  self._dt, self.device, C.byref(self._handle))

    def check_gradient(self, q, eps=1e-6):
        """Max relative deviation between gradient(q) and central differences of potential(q),
        both evaluated by the HIP kernels; q is (D,) or (D, N)."""
        q = np.asarray(q, dtype=np.float64)
        q2 = q.reshape(self.numDimensions, -1)
        g = self.gradient(q2)
        fd = np.empty_like(g)
        for j in range(self.numDimensions):
            dq = np.zeros_like(q2)
            dq[j] = eps
            fd[j] = (self(q2 + dq) - self(q2 - dq)) / (2 * eps)
        return float(np.max(np.abs(g - fd)) / max(1.0, float(np.max(np.abs(fd)))))
