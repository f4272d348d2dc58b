"""Tempering that USES the temperature: a ladder of sub-ensembles with replica exchange.

The reference draws momenta at kB*T (src/ensemble.py:88), keeps a commented-out `Ensemble.setWeights`
(src/ensemble.py:52-61) and plans "canonical <-> micro-canonical" ensembles
(references/PhysicsBasedHMC_SoHPC2022_WeekPlan.md:25-27); nothing in it consumes a temperature beyond the
momentum draw.  Here the ensemble's chains form R rungs at temperatures kT_0 = 1 < kT_1 < ... < kT_{R-1}:

  * every rung samples exp(-U / kT_r) with the fused HMC kernels (`pbbi_hmc_run` on the rung's block of the
    state, `PBBI_BETA_ACCEPT` with the rung's kT: the accept test at the temperature of the momentum draw);
  * between sampling stretches the rungs exchange positions on the device (`pbbi_replica_exchange`: the energy
    evaluation and ONE swap launch, no host round trip), alternating even and odd pairs.

Hot rungs cross barriers a kT = 1 chain never would; swaps carry the crossings down, so rung 0 samples the
target with all its modes.  Chains of one rung share nothing with each other, so every rung is itself an
ensemble of `chainsPerRung` chains -- the data-parallel shape the kernels want.
"""
import numpy as np

from . import _lib
from ._device import as_device, empty, stream_ptr, synchronize, to_numpy
from .integrator import resolve_potential

__all__ = ["TemperingLadder", "geometric_ladder"]


def geometric_ladder(kT_max, rungs):
    """kT_r = kT_max ** (r / (rungs - 1)): the usual geometric spacing from 1 to kT_max."""
    if rungs < 2:
        return np.ones(1)
    return float(kT_max) ** (np.arange(rungs) / (rungs - 1.0))


class TemperingLadder:
    """R sub-ensembles of `chainsPerRung` chains at temperatures `kTs` (kTs[0] is the target's, normally 1).

        ladder = TemperingLadder(potential, D, 4096, geometric_ladder(40.0, 6), simulTime=1.0, stepSize=0.1)
        samples = ladder.run(numSamples=200, qStd=1.0, burn_in=200)      # (D, chainsPerRung, numSamples), rung 0

    `potential` is a descriptor or a traceable callable (trace.py).  `stepSizes` (one per rung; default
    stepSize * sqrt(kT_r): a hotter rung moves on the scale sqrt(kT) wider) keeps the acceptance up the ladder.
    After run(): `acceptRates` (R), `swapRates` (R-1), `state` the (D, R*chainsPerRung) device state."""

    def __init__(self, potential, numDimensions, chainsPerRung, kTs, simulTime, stepSize, stepSizes=None, seed=0,
                 kdk_fma=True, draw_f64=False, mass=None):
        self.D, self.Nr = int(numDimensions), int(chainsPerRung)
        self.kTs = np.ascontiguousarray(kTs, dtype=np.float64)
        if self.kTs.ndim != 1 or self.kTs.size < 1 or np.any(self.kTs <= 0):
            raise ValueError("kTs must be a non-empty list of positive temperatures")
        self.R = self.kTs.size
        self.pot = resolve_potential(potential, "potential", self.D)
        if self.pot.numDimensions != self.D:
            raise ValueError(f"potential has D={self.pot.numDimensions}, ladder has D={self.D}")
        self.simulTime, self.stepSize, self.seed = float(simulTime), float(stepSize), int(seed)
        hs = self.stepSize * np.sqrt(self.kTs / self.kTs[0]) if stepSizes is None else np.asarray(stepSizes, float)
        if hs.shape != (self.R,):
            raise ValueError("stepSizes: one step size per rung")
        self.stepSizes = hs
        self.numSteps = [max(1, int(self.simulTime / h)) for h in hs]   # src/integrator.py:51 per rung
        self.flags = _lib.BETA_ACCEPT | (_lib.KDK_FMA if kdk_fma else 0) | (_lib.DRAW_F64 if draw_f64 else 0)
        if mass is not None:
            raise NotImplementedError("TemperingLadder: unit masses (the reference's default, src/ensemble.py:42)")
        self.acceptRates = self.swapRates = self.state = None

    def run(self, numSamples, qStd, swap_every=1, burn_in=0, record_rungs=(0,), device_output=False):
        """`burn_in` unrecorded then `numSamples` recorded iterations; after every `swap_every` HMC iterations one
        exchange step (even pairs, then odd pairs, alternating).  Returns rung 0's samples (D, chainsPerRung,
        numSamples) -- or a dict {rung: samples} when record_rungs names several."""
        import torch
        pot, D, Nr, R = self.pot, self.D, self.Nr, self.R
        N, dev, dt = R * Nr, pot.device, pot.dtype
        st = stream_ptr(dev)
        es = np.dtype(dt).itemsize
        S, B, k = int(numSamples), int(burn_in), max(1, int(swap_every))
        q = empty((D, N), dt, dev)
        _lib.call("pbbi_philox_normal", self.seed, _lib.STREAM_POSITION, 0, 0, D, N, N, float(qStd), None, pot._dt,
                  dev, q.data_ptr(), st)
        betas = as_device(1.0 / self.kTs, dev, np.float64)
        rec = {int(r): empty((max(S, 1), D, Nr), dt, dev) for r in record_rungs}
        scratch = empty((k, D, Nr), dt, dev)          # slabs of an unrecorded stretch
        reject = empty((k, Nr), np.uint8, dev)
        swapped = torch.zeros((max(R - 1, 1), Nr), dtype=torch.uint8, device=q.device)
        n_rej = torch.zeros(R, dtype=torch.float64, device=q.device)
        n_swap = torch.zeros(max(R - 1, 1), dtype=torch.float64, device=q.device)
        n_swap_tries = np.zeros(max(R - 1, 1))
        it, done, sweep = 0, 0, 0
        total = B + S
        while it < total:
            c = min(k, total - it, (B - it) if it < B else total)   # a stretch never straddles the burn-in boundary
            recording = it >= B
            for r in range(R):
                out = rec[r][done:done + c] if (recording and r in rec) else scratch[:c]
                _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr() + r * Nr * es, None, out.data_ptr(),
                          None, reject.data_ptr(), None, Nr, N, float(self.stepSizes[r]), int(self.numSteps[r]), c,
                          self.flags, self.seed, it, r * Nr, float(self.kTs[r]), st)
                n_rej[r] += reject[:c].sum()
            it += c
            if recording:
                done += c
            if R > 1:
                parity = sweep & 1
                swapped.zero_()
                _lib.call("pbbi_replica_exchange", pot.handle, q.data_ptr(), Nr, R, N, betas.data_ptr(), parity,
                          self.seed, sweep, 0, swapped.data_ptr(), st)
                n_swap += swapped.sum(dim=1)
                n_swap_tries[parity:R - 1:2] += Nr
                sweep += 1
        synchronize(dev)
        self.state = q
        self.acceptRates = 1.0 - to_numpy(n_rej) / max(total * Nr, 1)
        self.swapRates = to_numpy(n_swap)[:max(R - 1, 0)] / np.maximum(n_swap_tries[:max(R - 1, 0)], 1)
        outs = {r: (v[:S].permute(1, 2, 0) if device_output else np.ascontiguousarray(to_numpy(v[:S]).transpose(1, 2, 0)))
                for r, v in rec.items()}
        return outs[0] if list(outs) == [0] else outs
