"""Ensemble -- drop-in for the reference's `src/ensemble.py` (state container + RNG).

Same public surface: `Ensemble(numDimensions, numParticles)` with PUBLIC, mutable
NumPy attributes `q, p` of shape (numDimensions, numParticles) float64 C-order
(chain index contiguous: already the ensemble-major layout the kernels want),
`mass, weights` of shape (numParticles,); `setPosition(qStd)`, `setMomentum(T)`,
`particle(i)` (src/ensemble.py:25-43, 63-114).

RNG parity: `setPosition` / `setMomentum` draw from NumPy's *global* legacy
RandomState exactly as the reference's `scipy.stats.norm.rvs(scale, size)` does:
`standard_normal((D, N)) * scale`, element [d, n] = draw number d*N + n
(verified bit-exact against the reference in tests/golden).  This
host stream is what "identical seeds" means for the reference; the device-side
Philox stream (`HMC.getSamples(..., rng="philox")`) is the throughput mode.
Large draws go through `_hoststream` (csrc/hoststream.c): the same global stream and the same
state afterwards, bit for bit, with the Box-Muller transform spread over the host's cores.
"""
import numpy as np
from scipy.constants import k as boltzmannConst

from . import _hoststream

__all__ = ["Ensemble", "boltzmannConst"]


class Ensemble:
    def __init__(self, numDimensions, numParticles):
        D, N = int(numDimensions), int(numParticles)
        self.numDimensions, self.numParticles = D, N
        self.q, self.p = np.zeros((D, N)), np.zeros((D, N))  # (D, N): chain index fastest
        self.mass, self.weights = np.ones(N), np.zeros(N)

    # The reference's __iter__ (src/ensemble.py:45-50) returns a tuple and reads a
    # non-existent attribute; it cannot work and is deliberately not mirrored.

    def setPosition(self, qStd):
        """src/ensemble.py:63-76.  Rebinds and returns self.q (a NEW array)."""
        self.q = _hoststream.standard_normal((self.numDimensions, self.numParticles)) * qStd
        return self.q

    def setMomentum(self, temperature):
        """src/ensemble.py:78-93: p ~ N(0, mass*kB*T) per particle.  Rebinds self.p."""
        pStd = np.sqrt(self.mass * boltzmannConst * temperature)
        self.p = _hoststream.standard_normal((self.numDimensions, self.numParticles)) * pStd
        return self.p

    def particle(self, particleNum):
        """(q, p, mass, weight) of one ensemble member; IndexError outside
        [0, numParticles) like src/ensemble.py:95-114."""
        n = self.numParticles
        if particleNum < 0 or particleNum >= n:
            raise IndexError(f"Index {particleNum} out of bounds. numParticles={n}")
        col = particleNum
        return self.q[:, col], self.p[:, col], self.mass[col], self.weights[col]
