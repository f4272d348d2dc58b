"""Ensemble sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference has no multi-device path; its chains are mathematically independent
("ensemble of non-interacting particles": the gradient uses only q[:, i],
src/integrator.py:73; energies and the accept test are per chain, src/HMC.py:109-115,
168-176).  So the hot path shards embarrassingly over contiguous blocks of the chain
axis, with NO collective inside the sampling loop.  The single exchange is one
all-gather of the per-rank (S, D, N_local) sample slabs at collection time -- RCCL over
xGMI with backend "nccl" on GPUs, gloo in the CPU tests.

Reproducibility across shardings:
  rng="philox": counters carry the GLOBAL chain index (chain0 + n), so any sharding gives
      bit-identical chains.
  rng="numpy":  every rank replays the same global NumPy RandomState stream and keeps its
      own columns (HostStream), so the gathered result equals the single-process run.
"""
import numpy as np
from scipy.constants import k as boltzmannConst

from . import _hoststream

__all__ = ["ensemble_weights", "shard_bounds", "HostStream", "gather_samples", "get_samples_sharded"]


def shard_bounds(numParticles, rank, world):
    """Contiguous block [lo, hi) of the chain axis owned by `rank`; the remainder of an
    uneven split goes to the lowest ranks."""
    N, r, w = int(numParticles), int(rank), int(world)
    if not 0 <= r < w:
        raise ValueError("rank out of range")
    base, rem = divmod(N, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


class HostStream:
    """The reference's RNG consumption order on NumPy's global legacy RandomState,
    restricted to columns [lo, hi) of an ensemble of N_total chains:
      positions:  standard_normal((D, N_total)) * qStd             (src/ensemble.py:72-74)
      momenta:    standard_normal((D, N_total)) * sqrt(mass*kB*T)  (src/ensemble.py:88-91)
      uniforms:   uniform(size=N_total)                            (src/HMC.py:168)
    Every rank must have seeded the global stream identically (np.random.seed(s))."""

    def __init__(self, numDimensions, numParticlesTotal, lo=0, hi=None):
        self.D, self.N = int(numDimensions), int(numParticlesTotal)
        self.lo, self.hi = int(lo), self.N if hi is None else int(hi)

    def positions(self, qStd):
        return np.ascontiguousarray(
            (_hoststream.standard_normal((self.D, self.N)) * qStd)[:, self.lo:self.hi])

    def momenta(self, mass_local, temperature):
        z = _hoststream.standard_normal((self.D, self.N))[:, self.lo:self.hi]
        return np.ascontiguousarray(z * np.sqrt(np.asarray(mass_local) * boltzmannConst * temperature))

    def uniforms(self):
        return np.ascontiguousarray(_hoststream.uniform(self.N)[self.lo:self.hi])


def _dist():
    import torch.distributed as dist
    return dist


def gather_samples(local_sdn, group=None, _force_collective=False):
    """All-gather per-rank (S, D, N_local) slabs along the chain axis -> (S, D, N_total) on
    every rank (rank r's chains at columns shard_bounds(N_total, r, world)).  One collective:
    `all_gather_into_tensor` when the shards are equal, a padded one otherwise.  Works on CUDA
    tensors over RCCL and on CPU tensors over gloo."""
    import torch
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return local_sdn
    world = dist.get_world_size(group)
    if world == 1 and not _force_collective:  # (_force_collective: the one-rank RCCL test on a one-GPU box)
        return local_sdn
    S, D, Nl = local_sdn.shape
    sizes = torch.tensor([Nl], dtype=torch.int64, device=local_sdn.device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = [int(s.item()) for s in all_sizes]
    Nmax = max(all_sizes)
    send = local_sdn.contiguous()
    if Nl != Nmax:  # uneven split: pad to the largest shard, trim after the gather
        pad = torch.zeros((S, D, Nmax), dtype=send.dtype, device=send.device)
        pad[:, :, :Nl] = send
        send = pad
    flat = torch.empty((world * S, D, Nmax), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(flat, send, group=group)  # rank r's slab at rows [r*S, (r+1)*S)
    recv = flat.view(world, S, D, Nmax)
    if all(n == Nmax for n in all_sizes):
        return recv.permute(1, 2, 0, 3).reshape(S, D, world * Nmax)
    return torch.cat([recv[r, :, :, :all_sizes[r]] for r in range(world)], dim=2)


def ensemble_weights(H_local, beta=1.0, group=None):
    """Normalised canonical weights of a sharded ensemble from the per-chain Hamiltonians
    (pbbi_energy / HMC.getWeights give H and exp(-H); src/HMC.py:86-104):
        w_n = exp(-beta (H_n - H_min)) / Z,   Z = sum over ALL ranks' chains
    with the shift by the global minimum so that exp never underflows.  The first cross-chain
    reductions of the path: one all-reduce(MIN) and one all-reduce(SUM) of a scalar each (RCCL on
    CUDA tensors, gloo on CPU tensors); without a process group the sums are local.  A CUDA tensor
    is reduced by the library's own kernels, a CPU tensor (the gloo tests) by torch on the host.
    Returns (w_local, log_Z) with log_Z = log sum_n exp(-beta H_n)."""
    import torch
    dist = _dist()
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if torch.is_tensor(H_local) and H_local.is_cuda:
        # device tensors: the library's reduction kernels (include/pbbi.h, "ensemble weights"); only
        # the two all-reduced scalars are torch's business
        from . import _lib
        from ._device import stream_ptr
        H = H_local.contiguous()
        if H.dtype not in (torch.float64, torch.float32):
            H = H.double()
        code = _lib.F64 if H.dtype == torch.float64 else _lib.F32
        dev, n, st = H.device.index, H.numel(), stream_ptr(H.device.index)
        hmin = torch.empty(1, dtype=torch.float64, device=H.device)
        z = torch.empty(1, dtype=torch.float64, device=H.device)
        w = torch.empty_like(H)
        _lib.call("pbbi_reduce_min", H.data_ptr(), n, code, dev, hmin.data_ptr(), st)
        if sharded:
            dist.all_reduce(hmin, op=dist.ReduceOp.MIN, group=group)
        _lib.call("pbbi_canonical_weights", H.data_ptr(), n, float(beta), hmin.data_ptr(), code, dev,
                  w.data_ptr(), z.data_ptr(), st)
        if sharded:
            dist.all_reduce(z, op=dist.ReduceOp.SUM, group=group)
        _lib.call("pbbi_scale_inverse", w.data_ptr(), n, z.data_ptr(), code, dev, st)
        return w, float(torch.log(z) - beta * hmin)
    H = torch.as_tensor(H_local, dtype=torch.float64)
    hmin = torch.min(H).reshape(1) if H.numel() else torch.full((1,), float("inf"), dtype=torch.float64,
                                                                  device=H.device)
    if sharded:
        dist.all_reduce(hmin, op=dist.ReduceOp.MIN, group=group)
    e = torch.exp(-beta * (H - hmin))
    z = e.sum().reshape(1)
    if sharded:
        dist.all_reduce(z, op=dist.ReduceOp.SUM, group=group)
    return e / z, float(torch.log(z) - beta * hmin)


def get_samples_sharded(potential, numDimensions, numParticles, simulTime, stepSize, numSamples,
                        temperature, qStd, method="Leapfrog", rng="philox", seed=0, mass=None,
                        compat=True, group=None, gather=True, verbose=False, kdk_fma=None,
                        jitter=0.0, burn_in=0):
    """HMC.getSamples over an ensemble of `numParticles` chains sharded across the process
    group (one rank per GPU).  Returns torch tensors (samples, momenta) shaped (D, N, S):
    the gathered ensemble when gather=True (one RCCL all-gather each), else this rank's
    block.  `potential` must live on this rank's device.  kdk_fma / jitter / burn_in as in
    HMC / HMC.getSamples (jitter's step counts come from a seed-derived host stream: the same on
    every rank)."""
    from .ensemble import Ensemble
    from .HMC import HMC
    dist = _dist()
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    lo, hi = shard_bounds(numParticles, rank, world)
    ens = Ensemble(numDimensions, hi - lo)
    if mass is not None:
        ens.mass = np.asarray(mass, dtype=np.float64)[lo:hi].copy()
    hmc = HMC(ens, simulTime, stepSize, None, potential=potential, method=method, compat=compat,
              rng=rng, seed=seed, verbose=verbose, kdk_fma=kdk_fma)
    s, m = hmc.getSamples(numSamples, temperature, qStd, device_output=True, chain0=lo,
                          host_stream=HostStream(numDimensions, numParticles, lo, hi)
                          if rng == "numpy" else None, jitter=jitter, burn_in=burn_in)
    s_sdn, m_sdn = s.permute(2, 0, 1), m.permute(2, 0, 1)  # back to the (S, D, N_local) slabs
    if gather and world > 1:
        s_sdn, m_sdn = gather_samples(s_sdn, group), gather_samples(m_sdn, group)
    return s_sdn.permute(1, 2, 0), m_sdn.permute(1, 2, 0), hmc
