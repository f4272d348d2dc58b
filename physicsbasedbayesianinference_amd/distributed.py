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

__all__ = ["ensemble_weights", "shard_bounds", "HostStream", "gather_samples", "gather_blocks", "GatheredBlocks",
           "OverlappedGather", "get_samples_sharded", "sample_chunks_sharded"]


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


def _shard_sizes(n_total, world):
    return [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]


def _exchange_sizes(Nl, device, world, group):
    """the ranks' shard sizes when the caller did not say how the ensemble is split (one small all_gather)"""
    import torch
    if world == 1:
        return [int(Nl)]
    mine = torch.tensor([Nl], dtype=torch.int64, device=device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    _dist().all_gather(every, mine, group=group)
    return [int(x.item()) for x in every]


class GatheredBlocks:
    """What an all-gather of (S, D, N_local) slabs lands in: ONE buffer `blocks` of shape
    (world, S, D, N_max) -- rank r's slabs at blocks[r] (columns past sizes[r] are padding when the split is
    uneven) -- received IN PLACE by `all_gather_into_tensor`, nothing copied afterwards.  Views:
      view4()   (S, D, world, N_max): permuted view, chain (r, n) = global chain offsets[r] + n
      rank(r)   (S, D, sizes[r]) view of rank r's block
      to_sdn()  (S, D, N_total) contiguous COPY (the chain axis is not contiguous in `blocks`)
    A reduction over all draws of all chains (pbbi_sample_moments & co.) can read `blocks` itself as
    (world*S, D, N_max) slabs when the split is even."""

    def __init__(self, blocks, sizes):
        self.blocks, self.sizes = blocks, list(sizes)
        self.offsets = [0]
        for n in self.sizes[:-1]:
            self.offsets.append(self.offsets[-1] + n)
        self.n_total = sum(self.sizes)

    @property
    def even(self):
        return all(n == self.sizes[0] for n in self.sizes)

    def view4(self):
        return self.blocks.permute(1, 2, 0, 3)

    def rank(self, r):
        return self.blocks[r, :, :, :self.sizes[r]]

    def to_sdn(self, out=None):
        import torch
        world, S, D, _ = self.blocks.shape
        if out is None:
            out = torch.empty((S, D, self.n_total), dtype=self.blocks.dtype, device=self.blocks.device)
        for r in range(world):
            out[:, :, self.offsets[r]:self.offsets[r] + self.sizes[r]] = self.rank(r)
        return out


def gather_blocks(local_sdn, n_total=None, group=None, out=None, async_op=False, _force_collective=False,
                  sizes=None):
    """ONE all-gather of per-rank (S, D, N_local) slabs, received in place into a (world, S, D, N_max)
    buffer (`out` reuses a caller-owned one) -> GatheredBlocks (no copy, no host synchronisation).
    `n_total`: chains of the whole ensemble, sharded by `shard_bounds` -- the shard sizes then follow
    without an exchange; None asks the ranks (one small all_gather and a host sync).
    async_op=True returns (GatheredBlocks, work): the collective is in flight until work.wait()."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    S, D, Nl = local_sdn.shape
    if sizes is not None:       # (the caller knows every rank's shard size already)
        sizes = list(sizes)
    elif n_total is not None:
        sizes = _shard_sizes(int(n_total), world)
        rank = dist.get_rank(group) if world > 1 else 0
        if sizes[rank] != Nl:
            raise ValueError(f"rank {rank} holds {Nl} chains, shard_bounds({n_total}, {rank}, {world}) says {sizes[rank]}")
    else:
        sizes = _exchange_sizes(Nl, local_sdn.device, world, group)
    Nmax = max(sizes)
    send = local_sdn.contiguous()
    if Nl != Nmax:  # uneven split: the send buffer is padded to the largest shard
        pad = torch.zeros((S, D, Nmax), dtype=send.dtype, device=send.device)
        pad[:, :, :Nl] = send
        send = pad
    if out is None:
        out = torch.empty((world, S, D, Nmax), dtype=send.dtype, device=send.device)
    elif tuple(out.shape) != (world, S, D, Nmax) or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous ({world}, {S}, {D}, {Nmax}) buffer")
    work = None
    if world == 1 and not _force_collective:
        out[0].copy_(send)
    else:
        work = dist.all_gather_into_tensor(out.view(world * S, D, Nmax), send, group=group, async_op=async_op)
    res = GatheredBlocks(out, sizes)
    return (res, work) if async_op else res


def gather_samples(local_sdn, group=None, _force_collective=False, n_total=None, max_chunk_bytes=1 << 28):
    """All-gather per-rank (S, D, N_local) slabs along the chain axis -> a contiguous (S, D, N_total) tensor on
    every rank (rank r's chains at columns shard_bounds(N_total, r, world)).  The chain axis of the result
    is not contiguous in what a collective delivers (rank-major blocks), so the blocks are re-laid: to keep
    the peak at ~1x the result the slabs travel in groups of <= max_chunk_bytes per rank, each received in
    place and copied into its columns (`gather_blocks` returns the received blocks themselves, with no
    copy at all).  Works on CUDA tensors over RCCL and on CPU tensors over gloo."""
    import torch
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return local_sdn
    world = dist.get_world_size(group)
    if world == 1 and not _force_collective:  # (_force_collective: the one-rank RCCL test on a one-GPU box)
        return local_sdn
    S, D, Nl = local_sdn.shape
    sizes = _exchange_sizes(Nl, local_sdn.device, world, group) if n_total is None else _shard_sizes(int(n_total), world)
    N = sum(sizes)
    out = torch.empty((S, D, N), dtype=local_sdn.dtype, device=local_sdn.device)
    slab_bytes = max(1, D * max(sizes) * local_sdn.element_size())
    step = max(1, min(S, max_chunk_bytes // slab_bytes)) if S else 1
    buf = None
    for s0 in range(0, S, step):
        c = min(step, S - s0)
        if buf is None or buf.shape[1] != c:
            buf = torch.empty((world, c, D, max(sizes)), dtype=local_sdn.dtype, device=local_sdn.device)
        blk = gather_blocks(local_sdn[s0:s0 + c], None, group, out=buf, _force_collective=_force_collective, sizes=sizes)
        blk.to_sdn(out[s0:s0 + c])
    return out


class OverlappedGather:
    """Collection of a long sharded run off the sampling critical path ("per large chunk,
    double-buffered"): chunk k's all-gather is in flight on a side stream while chunk k+1 samples.

        og = OverlappedGather((c, D, N_local), dtype, device, n_total)
        for k in range(n_chunks):
            buf = og.local(k)              # this rank's (c, D, N_local) slab buffer for chunk k (two alternate)
            <enqueue the sampling of chunk k into buf[:rows]>
            done = og.submit(k, rows)      # starts chunk k's gather; returns chunk k-1's GatheredBlocks (or None)
            if done: consume(done)         # valid until the submit after next
        consume(og.finish())               # the last chunk

    Two send and two receive buffers.  On CUDA tensors the gather of chunk k is ordered after the sampling of
    chunk k by an event and runs beside chunk k+1's kernels; the sampling stream waits for chunk k-1's gather
    only when it hands that chunk out.  On CPU tensors (gloo tests) it is the same protocol with async_op."""

    def __init__(self, shape_local, dtype, device, n_total, group=None):
        import torch
        dist = _dist()
        self.t, self.group = torch, group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        c, D, Nl = shape_local
        self.sizes = _shard_sizes(int(n_total), self.world)
        if self.sizes[self.rank] != Nl:
            raise ValueError("shape_local does not match shard_bounds(n_total, rank, world)")
        Nmax = max(self.sizes)
        dev = torch.device(device)
        self.cuda = dev.type == "cuda"
        # an uneven split sends padded slabs: the local buffers are allocated at N_max and handed out as
        # [:, :, :N_local] views -- but a kernel wants dense (D, N_local) slabs, so the padded send buffer is
        # separate in that case (one extra copy per chunk on the ranks with the short shard only)
        self.pad = Nl != Nmax
        self.send = [torch.empty((c, D, Nl), dtype=dtype, device=dev) for _ in range(2)]
        self.sendpad = [torch.zeros((c, D, Nmax), dtype=dtype, device=dev) for _ in range(2)] if self.pad else None
        self.recv = [torch.empty((self.world, c, D, Nmax), dtype=dtype, device=dev) for _ in range(2)]
        self.work = [None, None]
        self.rows = [0, 0]
        self.pending = None
        if self.cuda:
            self.side = torch.cuda.Stream(device=dev)
            self.sampled = [torch.cuda.Event() for _ in range(2)]
            self.gathered = [torch.cuda.Event() for _ in range(2)]
            self.consumed = [None, None]

    def local(self, k):
        """this rank's slab buffer for chunk k; safe to overwrite once submit(k - 1) has returned"""
        return self.send[k & 1]

    def _start(self, k, rows):
        t, b = self.t, k & 1
        dist = _dist()
        src = self.send[b]
        recv = self.recv[b][:, :rows] if rows == self.recv[b].shape[1] else None
        if self.cuda:
            main = t.cuda.current_stream(src.device)
            self.sampled[b].record(main)                 # chunk k's kernels are enqueued up to here
            if self.consumed[b] is not None:             # whoever read recv[b] (two chunks ago) is done by then
                self.side.wait_event(self.consumed[b])
            self.side.wait_event(self.sampled[b])
            ctx = t.cuda.stream(self.side)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if self.pad:
                self.sendpad[b][:rows, :, :src.shape[2]].copy_(src[:rows])
                src = self.sendpad[b]
            if recv is None:   # a short last chunk: gather the rows that exist into a dense sub-buffer
                recv = self.recv[b].view(-1)[:self.world * rows * src.shape[1] * src.shape[2]].view(
                    self.world, rows, src.shape[1], src.shape[2])
            if self.world > 1:
                self.work[b] = dist.all_gather_into_tensor(recv.view(self.world * rows, *recv.shape[2:]),
                                                           src[:rows].contiguous(), group=self.group, async_op=True)
            else:
                recv[0].copy_(src[:rows])
            if self.cuda:
                if self.work[b] is not None:
                    self.work[b].wait()                  # orders the SIDE stream after the collective
                self.gathered[b].record(self.side)
        self.rows[b] = rows
        self._views = getattr(self, "_views", [None, None])
        self._views[b] = recv

    def _hand_out(self, k):
        b = k & 1
        if self.cuda:
            main = self.t.cuda.current_stream(self.recv[b].device)
            main.wait_event(self.gathered[b])            # the consumer's work is enqueued behind the gather
            ev = self.t.cuda.Event()
            self.consumed[b] = ev                        # recorded by the NEXT submit / finish, see _mark
            self._to_mark = (b, ev)
        elif self.work[b] is not None:
            self.work[b].wait()
        self.work[b] = None
        return GatheredBlocks(self._views[b], self.sizes)

    def _mark(self):
        # everything the consumer enqueued on the sampling stream since the last hand-out is recorded now
        tm = getattr(self, "_to_mark", None)
        if tm is not None and self.cuda:
            b, ev = tm
            ev.record(self.t.cuda.current_stream(self.recv[b].device))
        self._to_mark = None

    def submit(self, k, rows=None):
        self._mark()
        rows = self.send[k & 1].shape[0] if rows is None else int(rows)
        self._start(k, rows)
        done = self._hand_out(self.pending) if self.pending is not None else None
        self.pending = k
        return done

    def finish(self):
        self._mark()
        if self.pending is None:
            return None
        done = self._hand_out(self.pending)
        self.pending = None
        return done


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
        s_sdn = gather_samples(s_sdn, group, n_total=numParticles)
        m_sdn = gather_samples(m_sdn, group, n_total=numParticles)
    return s_sdn.permute(1, 2, 0), m_sdn.permute(1, 2, 0), hmc


def sample_chunks_sharded(potential, numDimensions, numParticles, simulTime, stepSize, numSamples, chunk,
                          temperature, qStd, method="Leapfrog", rng="philox", seed=0, mass=None, compat=True,
                          group=None, verbose=False, kdk_fma=None, draw_f64=False, momenta=False):
    """A long sharded run collected WHILE it samples: the ensemble of `numParticles` chains is
    sharded over the process group, sampled in chunks of `chunk` iterations, and chunk k's all-gather (RCCL
    over xGMI) runs on a side stream while chunk k+1's kernels run -- two send and two receive buffers,
    received in place, nothing re-laid.  Generator of (GatheredBlocks samples, GatheredBlocks momenta | None,
    hmc) per chunk: `.blocks` is (world, c, D, N_max), `.view4()` the (c, D, world, N_max) view, `.to_sdn()`
    a contiguous copy; a chunk's buffers are overwritten two chunks later.  The concatenated chunks equal
    ONE get_samples_sharded / single-process run bit for bit (both RNG modes, any split)."""
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
    hmc = HMC(ens, simulTime, stepSize, None, potential=potential, method=method, compat=compat, rng=rng,
              seed=seed, verbose=verbose, kdk_fma=kdk_fma, draw_f64=draw_f64)
    hs = HostStream(numDimensions, numParticles, lo, hi) if rng == "numpy" else None
    yield from hmc.sampleChunksGathered(numSamples, chunk, temperature, qStd, n_total=numParticles, chain0=lo,
                                        host_stream=hs, momenta=momenta, group=group)
