"""HMC -- drop-in for the reference's `src/HMC.py` (ensemble Hamiltonian Monte Carlo).

`HMC(ensemble, simulTime, stepSize, density, potential=None, gradient=None,
method="Leapfrog")` and `getSamples(numSamples, temperature, qStd)` keep the
reference's signature, return shapes `(D, N, S)` float64 and semantics
(src/HMC.py:26-71, 123-183).  The body of the reference's sampling loop --
integrate, energies, ratio, accept/reject, record (src/HMC.py:154-179) -- is ONE
fused HIP kernel launch per iteration over the whole ensemble.

Two RNG modes:
  rng="numpy"  (default, reference parity): positions, momenta and uniforms come
      from NumPy's global legacy RandomState in the reference's order (D*N normals
      per iteration, then N uniforms; src/ensemble.py:72-74,88-91, src/HMC.py:168)
      and are uploaded; `np.random.seed(s)` therefore reproduces the reference run.
  rng="philox" (throughput): everything is drawn inside the kernels from the
      counter-based stream of include/pbbi.h; the chain state never leaves HBM.

`compat=True` (default) reproduces the reference's quirks:
rejected chains record their OLD POSITION as momentum (src/HMC.py:176); momenta are
stored un-negated (:164,:179); a NaN ratio is accepted (:168-173); the accept test
uses beta = 1 whatever `temperature` is (:115).  `compat=False` only changes the
first: a rejected chain then records the momentum it drew.

`kdk_fma` (Leapfrog only; default: True for rng="philox", False for rng="numpy") passes
PBBI_KDK_FMA (include/pbbi.h): kernels that honour it
integrate in kick-drift-kick form with fused multiply-adds -- the same integrator algebraically,
trajectories within ~1e-13 of the reference's operation order instead of bit-identical, accept
masks as before -- which turns the elementwise-potential kernels from instruction-bound into
HBM-bound.  kdk_fma=False keeps the reference's operation order everywhere.
"""
import numpy as np
from scipy.constants import Boltzmann as boltzmannConst

from . import _hoststream, _lib
from ._device import as_device, empty, stream_ptr, synchronize, to_numpy
from .integrator import Leapfrog, StormerVerlet, mass_or_none, resolve_potential
from .potential import Potential

__all__ = ["HMC"]


class _HostDrawPipeline:
    """rng="numpy": the reference's stream (src/ensemble.py:88-91, src/HMC.py:168) is NumPy's global
    legacy RandomState -- sequential by construction, drawn on the host.  A producer thread draws
    iteration i+1 (NumPy releases the GIL while it generates), stages it in pinned memory and starts
    the H2D copy on a side stream while the main thread launches iteration i; three slots.  The
    consumption ORDER of the global stream is exactly the reference's: only the producer draws."""

    class Slot:
        pass

    def __init__(self, draw, S, D, N, device, np_dtype, depth=3, cache=None):
        import queue
        import threading
        from ._device import dev, torch, torch_dtype
        t = torch()
        self.t, self.S, self.draw = t, int(S), draw
        self.draw_seconds = 0.0
        self.ready = queue.Queue()
        self.direct = np.dtype(np_dtype) == np.dtype(np.float64)
        self.error = None
        self.thread = None
        self.stop = False
        self.finished = False
        # the staging buffers outlive the call in `cache` (a dict owned by the HMC object: HMC.releaseHostBuffers
        # empties it); the producer draws ahead of the launches, so the NumPy state BEFORE each iteration's draw
        # is kept: a run that dies half way hands back the state that follows its last LAUNCHED iteration
        self.cache = cache if cache is not None else {}
        self.states = {}
        self.launched = 0
        if self.S == 0:
            return
        self.side = t.cuda.Stream(device=dev(device))
        self.main = t.cuda.current_stream(dev(device))
        td = torch_dtype(np_dtype)
        self.free = queue.Queue()   # slots the consumer has handed back (initially: all of them)
        # the staging buffers of the previous call with the same shape are kept (pinning 3 x D*N doubles takes
        # longer than a short run's draws); one shape at a time
        self.key = (int(D), int(N), str(td), str(dev(device)))
        cached = self.cache.pop(self.key, [])
        self.cache.clear()
        self.slots = []
        for _ in range(min(depth, self.S)):
            if cached:
                s = cached.pop()
            else:
                s = self.Slot()
                s.pin_p = t.empty((D, N), dtype=td, pin_memory=True)
                s.pin_u = t.empty((N,), dtype=td, pin_memory=True)
                s.p = t.empty((D, N), dtype=td, device=dev(device))
                s.u = t.empty((N,), dtype=td, device=dev(device))
                s.uploaded = t.cuda.Event()
            s.consumed = None
            self.slots.append(s)
            self.free.put(s)
        # One producer thread for the process, not one per call: the host stream's OpenMP team belongs to the
        # thread that calls it, and a fresh thread would start a fresh team (and first-touch its buffers) in
        # every getSamples call -- 3 ms per iteration of a 30-iteration run.
        self.done = threading.Event()
        self.thread = _producer()
        self.thread.jobs.put(self._run)

    def _run(self):
        try:
            self._produce()
        finally:
            self.done.set()

    def _produce(self):
        import time
        t = self.t
        try:
            for i in range(self.S):
                if self.stop:  # the consumer gave up (an error in a launch): leave the rest of the stream undrawn
                    return
                # a slot is reused only after the consumer has LAUNCHED the kernel that reads it (it comes
                # back through `free`) and that kernel has finished (its event)
                s = self.free.get()
                if s is None or self.stop:
                    return
                if s.consumed is not None:
                    s.consumed.synchronize()
                self.states[i] = np.random.get_state()   # (2.5 KB; iterations older than the slots are dropped)
                self.states.pop(i - len(self.slots) - 1, None)
                t0 = time.perf_counter()
                if self.direct:   # float64 handle: the draws are written into the pinned buffers themselves
                    self.draw(i, s.pin_p.numpy(), s.pin_u.numpy())
                    self.draw_seconds += time.perf_counter() - t0
                else:
                    p, u = self.draw(i)
                    self.draw_seconds += time.perf_counter() - t0
                    s.pin_p.copy_(t.from_numpy(np.ascontiguousarray(p)))   # converts float64 -> the handle's dtype
                    s.pin_u.copy_(t.from_numpy(np.ascontiguousarray(u)))
                with t.cuda.stream(self.side):
                    s.p.copy_(s.pin_p, non_blocking=True)
                    s.u.copy_(s.pin_u, non_blocking=True)
                    s.uploaded.record(self.side)
                self.ready.put(s)
        except BaseException as e:  # surfaces in acquire()
            self.error = e
            self.ready.put(None)

    def acquire(self):
        s = self.ready.get()
        if s is None:
            raise self.error
        self.main.wait_event(s.uploaded)
        return s

    def release(self, s):
        s.consumed = self.t.cuda.Event()
        s.consumed.record(self.main)
        self.launched += 1
        self.free.put(s)

    def close(self):
        self.stop = True
        if self.thread is not None:
            self.free.put(None)   # wakes a producer that waits for a slot
            self.done.wait()
            # a run that went through keeps its staging buffers for the next call (getSamples synchronises the
            # device before it returns, so nothing reads them any more)
            if self.finished and self.error is None:
                self.cache[self.key] = list(self.slots)
            elif self.launched in self.states:
                # the consumer gave up after `launched` iterations but the producer had drawn further: the global
                # stream goes back to where the reference would have left it at that point
                np.random.set_state(self.states[self.launched])

class _Producer:
    """The process's host-draw thread: runs the pipelines' producer loops one after the other."""

    def __init__(self):
        import queue
        import threading
        self.jobs = queue.Queue()
        self.thread = threading.Thread(target=self._loop, daemon=True, name="pbbi-host-draws")
        self.thread.start()

    def _loop(self):
        while True:
            self.jobs.get()()


_producer_singleton = None


def _producer():
    global _producer_singleton
    if _producer_singleton is None or not _producer_singleton.thread.is_alive():
        _producer_singleton = _Producer()
    return _producer_singleton


# adaptStepSize's Philox key = seed ^ this (the sampling run keeps `seed`)
WARMUP_SEED_MASK = 0xA5A55A5ADA7A0001


class HMC:
    def __init__(self, ensemble, simulTime, stepSize, density, potential=None, gradient=None,
                 method="Leapfrog", compat=True, rng="numpy", seed=0, verbose=True, kdk_fma=None,
                 beta_accept=False, draw_f64=False):
        self.ensemble = ensemble
        self.simulTime = simulTime
        self.stepSize = stepSize
        self.density = density
        self.compat = bool(compat)
        # PBBI_BETA_ACCEPT (include/pbbi.h): accept with exp((oldH - newH) / (kB*T)), the test that
        # matches the momentum draw at kB*T; default off = the reference's exp(oldH - newH)
        self.beta_accept = bool(beta_accept)
        # PBBI_DRAW_F64 (rng="philox"): positions and momenta drawn in double precision, the counterpart of
        # the reference's float64 normals (src/ensemble.py:72-74,88-91); default: the single-precision draw
        self.draw_f64 = bool(draw_f64)
        # PBBI_KDK_FMA, the throughput form of Leapfrog (include/pbbi.h): default on in the
        # throughput RNG mode, off in the reference-parity mode
        self.kdk_fma = (rng == "philox") if kdk_fma is None else bool(kdk_fma)
        self.rng = rng
        self.seed = int(seed)
        self.verbose = verbose

        # potential: given descriptor, else the descriptor behind `density` (-log density); plain Python
        # callables (the reference's lambdas, :52-60) are traced into one (trace.py)
        D = ensemble.numDimensions
        traced_gradient = gradient is not None and callable(gradient) and \
            not isinstance(getattr(gradient, "__self__", gradient), Potential)
        if potential:
            if traced_gradient and not isinstance(getattr(potential, "__self__", potential), Potential):
                # potential AND gradient as callables: one descriptor from both (a hand-written gradient
                # is used as written; grad(potential) means "differentiate the trace")
                self._pot = resolve_potential(gradient, "gradient", D, potential=potential)
            else:
                self._pot = resolve_potential(potential, "potential", D)
                if traced_gradient:
                    raise ValueError("gradient= is a Python callable but potential= is a descriptor, which "
                                     "brings its own gradient")
            self.potential = potential
        else:
            self._pot = resolve_potential(density, "density", D)
            self.potential = self.potentialFunc
        if gradient:
            if not traced_gradient and resolve_potential(gradient, "gradient") is not self._pot:
                raise ValueError("gradient= must belong to the same descriptor as potential=")
            self.gradient = gradient
        else:
            self.gradient = self._pot.gradient  # stands in for jax.grad(self.potential) (:60)
        # the integrator gets the DESCRIPTOR's gradient: whatever `gradient` was, it is part of self._pot now
        integ_gradient = self._pot.gradient

        if method == "Leapfrog":
            self.integrator = Leapfrog(ensemble, stepSize, simulTime, integ_gradient)
        elif method == "Stormer-Verlet":
            self.integrator = StormerVerlet(ensemble, stepSize, simulTime, integ_gradient)
        else:
            raise ValueError("Invalid integration method selected.")
        self.method = method
        # diagnostics of the last getSamples call
        self.reject_masks = None   # (S, N) bool
        self.ratios = None         # (S, N) exp(oldH - newH)
        self.acceptRate = None
        self._host_cache = {}      # rng="numpy": pinned staging buffers kept between getSamples calls

    def describeRun(self, numSamples=2):
        """What getSamples(numSamples, rng="philox") would launch, in words (pbbi_describe_run): the kernel
        family, whether the gradient is carried between iterations (and why not), iterations per launch."""
        import ctypes
        buf = ctypes.create_string_buffer(1024)
        N = self.ensemble.numParticles
        _lib.call("pbbi_describe_run", self._pot.handle, self.integrator.method_id, N, N,
                  int(self.integrator.numSteps), int(numSamples), self._flags(), buf, len(buf))
        return buf.value.decode()

    def releaseHostBuffers(self):
        """Drop the pinned host / device staging buffers the rng="numpy" pipeline keeps between calls (three
        (D, N) + (N,) pairs: ~200 MB pinned and ~200 MB of HBM at config C2); the next call pins new ones."""
        self._host_cache.clear()

    # ------------------------------------------------------------------ helpers
    def potentialFunc(self, q):
        """U(q) = -log(density(q))  (src/HMC.py:75-84); for a traced density: the descriptor's U(q),
        evaluated by the HIP kernel."""
        if not isinstance(getattr(self.density, "__self__", self.density), Potential):
            return self._pot(q)
        return -np.log(self.density(q))

    def _upload_state(self, *arrays):
        pot = self._pot
        D, N = self.ensemble.numDimensions, self.ensemble.numParticles
        out = []
        for a in arrays:
            a = np.asarray(a)
            if a.shape != (D, N):
                raise ValueError(f"expected a ({D}, {N}) array, got {a.shape}")
            out.append(as_device(a, pot.device, pot.dtype))
        return out

    def _flags(self):
        return ((_lib.COMPAT_P_FROM_OLDQ if self.compat else 0) | (_lib.KDK_FMA if self.kdk_fma else 0) |
                (_lib.BETA_ACCEPT if self.beta_accept else 0) | (_lib.DRAW_F64 if self.draw_f64 else 0))

    def _position_stream(self):
        return _lib.STREAM_POSITION | (_lib.STREAM_DRAW_F64 if self.draw_f64 else 0)

    def ensembleWeights(self, q, p, temperature=None):
        """Normalised canonical weights of the ensemble, w_n = exp(-beta H_n) / sum_m exp(-beta H_m)
        with H = 0.5 p.p/mass + potential(q) (HMC.getWeights, src/HMC.py:86-104, normalised; the
        reference's commented-out Ensemble.setWeights, src/ensemble.py:52-61) and beta = 1/(kB*T)
        (beta = 1 when temperature is None).  Energies, the min / sum reductions and the scaling all
        run on the GPU (pbbi_energy, pbbi_reduce_min, pbbi_canonical_weights, pbbi_scale_inverse);
        with an initialised torch.distributed group the two scalars are all-reduced, so every rank
        holds its shard of the weights of the WHOLE ensemble.  Also stored in ensemble.weights."""
        from . import distributed
        pot = self._pot
        N = self.ensemble.numParticles
        qd, pd = self._upload_state(q, p)
        md = self._mass()
        H = empty((N,), pot.dtype, pot.device)
        _lib.call("pbbi_energy", pot.handle, qd.data_ptr(), pd.data_ptr(),
                  md.data_ptr() if md is not None else None, N, N, H.data_ptr(), None, stream_ptr(pot.device))
        beta = 1.0 if temperature is None else 1.0 / float(boltzmannConst * temperature)
        w = to_numpy(distributed.ensemble_weights(H, beta)[0]).astype(np.float64)
        self.ensemble.weights = w
        return w

    def _mass(self):
        pot = self._pot
        return mass_or_none(self.ensemble.mass, self.ensemble.numParticles, pot.dtype, pot.device)

    def getWeights(self, q, p):
        """exp(-H) per chain, H = 0.5*dot(p,p)/mass + potential(q)   (src/HMC.py:86-104)."""
        pot = self._pot
        N = self.ensemble.numParticles
        qd, pd = self._upload_state(q, p)
        md = self._mass()
        w = empty((N,), pot.dtype, pot.device)
        _lib.call("pbbi_energy", pot.handle, qd.data_ptr(), pd.data_ptr(),
                  md.data_ptr() if md is not None else None, N, N, None, w.data_ptr(),
                  stream_ptr(pot.device))
        return to_numpy(w)

    def getWeightsRatio(self, newQ, newP, oldQ, oldP):
        """exp(oldH - newH) per chain   (src/HMC.py:106-116)."""
        pot = self._pot
        N = self.ensemble.numParticles
        nq, np_, oq, op = self._upload_state(newQ, newP, oldQ, oldP)
        md = self._mass()
        r = empty((N,), pot.dtype, pot.device)
        _lib.call("pbbi_weights_ratio", pot.handle, nq.data_ptr(), np_.data_ptr(), oq.data_ptr(),
                  op.data_ptr(), md.data_ptr() if md is not None else None, N, N, r.data_ptr(),
                  stream_ptr(pot.device))
        return to_numpy(r)

    def print_information(self):
        print("integrator: ", self.integrator)
        print("final integration time: ", self.simulTime)
        print("time step: ", self.stepSize)

    # ------------------------------------------------------------------ sampling
    def _numpy_stream_run(self, S, temperature, qStd, host_stream=None):
        """rng="numpy" plumbing shared by getSamples and sampleChunksGathered: identical RNG consumption to the
        reference -- q0 (src/HMC.py:148), then per iteration p (src/ensemble.py:88-91, src/HMC.py:154) and u
        (src/HMC.py:168) -- drawn by the host pipeline (one producer thread, pinned staging, side-stream
        upload) and consumed by one pbbi_hmc_iter_kt launch per iteration."""
        pot, ens = self._pot, self.ensemble
        D, N = ens.numDimensions, ens.numParticles
        dev, dt = pot.device, pot.dtype
        if host_stream is None:
            self.integrator.q = ens.setPosition(qStd)                   # :148
        else:
            self.integrator.q = ens.q = host_stream.positions(qStd)
        hmc = self

        class Run:
            pass
        run = Run()
        run.q0 = as_device(self.integrator.q, dev, dt)
        kT_host = float(boltzmannConst * temperature) if self.beta_accept else 1.0

        def draw(i, pin_p=None, pin_u=None):
            """Iteration i's draws in the reference's order (p, then u).  With float64 upload buffers
            (pin_p, pin_u: NumPy views of pinned memory) they are written there directly."""
            if hmc.verbose and i % 100 == 0:
                print("HMC iteration ", i + 1)                           # :151-152
            if host_stream is None and pin_p is not None:
                pStd = np.sqrt(ens.mass * boltzmannConst * temperature)  # src/ensemble.py:88
                p = ens.p = _hoststream.scaled_normal_into(pin_p, pStd)  # :89-91, :154
                u = _hoststream.uniform_into(pin_u)                      # :168
            elif host_stream is None:
                p = ens.setMomentum(temperature)                         # :154
                u = _hoststream.uniform(N)                               # :168 (np.random.uniform(size=N))
            else:
                p = ens.p = host_stream.momenta(ens.mass, temperature)
                u = host_stream.uniforms()
                if pin_p is not None:
                    pin_p[...], pin_u[...] = p, u
            hmc.integrator.p = p
            return p, u
        # The NumPy legacy stream can only be drawn in order, on the host (~15 ns per normal):
        # a producer thread draws iteration i+1 into pinned memory and starts its upload on a
        # side stream while iteration i's kernel runs (_HostDrawPipeline above).
        run.pipe = _HostDrawPipeline(draw, S, D, N, dev, dt, cache=self._host_cache)
        md = self._mass()
        mptr = md.data_ptr() if md is not None else None
        L, h, flags, stream = self.integrator.numSteps, float(self.stepSize), self._flags(), stream_ptr(dev)

        def step(i, q_in, q_out, p_out, ratio_out, reject_out):
            slot = run.pipe.acquire()
            _lib.call("pbbi_hmc_iter_kt", pot.handle, hmc.integrator.method_id, q_in.data_ptr(),
                      slot.p.data_ptr(), slot.u.data_ptr(), mptr, q_out.data_ptr(),
                      p_out.data_ptr() if p_out is not None else None,
                      ratio_out.data_ptr() if ratio_out is not None else None,
                      reject_out.data_ptr() if reject_out is not None else None, N, N, h, L, flags, kT_host,
                      stream)
            run.pipe.release(slot)
        run.step, run.keep = step, md
        return run

    def getSamples(self, numSamples, temperature, qStd, rng=None, seed=None, device_output=False,
                   chain0=0, iter0=0, host_stream=None, jitter=0.0, burn_in=0, per_chain_steps=False):
        """HMC.getSamples (src/HMC.py:123-183): returns (samples_hmc, momentum_hmc), each
        (D, N, numSamples) with the sample index fastest.

        Extra keyword-only behaviour (not in the reference): rng / seed override the
        constructor's; device_output=True returns torch tensors that are (D, N, S)
        *views* of the (S, D, N) device slabs (no transpose pass, no D2H copy).
        chain0 / iter0 offset the Philox counters (ensemble sharding / resuming);
        host_stream (distributed.HostStream) makes rng="numpy" keep only this shard's
        columns of the global NumPy stream.  jitter in (0, 1) (rng="philox" only) draws the number
        of leapfrog steps of every ITERATION uniformly from [L(1-jitter), L(1+jitter)] (one value
        for the whole ensemble, from a host stream seeded by `seed`, identical on every shard): a
        fixed trajectory length leaves modes with omega*T near k*pi unmixed; the mixture of
        trajectory lengths is still a valid HMC kernel.  burn_in (rng="philox" only) runs that many
        unrecorded iterations first (draw indices iter0 .. iter0+burn_in-1; the recorded ones
        follow), without sample or momentum slabs: getSamples(S, burn_in=B) equals the last S
        draws of getSamples(B + S).  per_chain_steps=True (rng="philox", Leapfrog; PBBI_PER_CHAIN_STEPS,
        include/pbbi.h) gives every CHAIN its own number of leapfrog steps in every iteration, uniform
        on [1, numSteps] from the Philox stream: randomised-length HMC per chain, inside the fused
        kernels (finished chains are masked out); self.steps holds the (S, N) counts.
        """
        pot = self._pot
        ens = self.ensemble
        D, N, S = ens.numDimensions, ens.numParticles, int(numSamples)
        if D != pot.numDimensions:
            raise ValueError(f"potential has D={pot.numDimensions}, ensemble has D={D}")
        rng = self.rng if rng is None else rng
        seed = self.seed if seed is None else int(seed)
        L = self.integrator.numSteps
        h = float(self.stepSize)
        flags = self._flags()
        dev, dt = pot.device, pot.dtype
        stream = stream_ptr(dev)

        samples = empty((S, D, N), dt, dev)
        momenta = empty((S, D, N), dt, dev)
        reject = empty((S, N), np.uint8, dev)
        ratio = empty((S, N), dt, dev)
        md = self._mass()
        mptr = md.data_ptr() if md is not None else None

        if self.verbose:
            self.print_information()
        if rng == "numpy" and (burn_in or jitter or per_chain_steps):
            raise ValueError("burn_in / jitter / per_chain_steps need rng='philox' (the parity mode replays "
                             "the reference's stream)")
        self.steps = None
        if rng == "numpy":
            run = self._numpy_stream_run(S, temperature, qStd, host_stream)
            q_prev = run.q0
            try:
                for i in range(S):
                    run.step(i, q_prev, samples[i], momenta[i], ratio[i], reject[i])
                    q_prev = samples[i]
                run.pipe.finished = True  # every iteration was launched: the staging buffers may be kept
            finally:
                run.pipe.close()
                if not run.pipe.finished and run.pipe.direct and S > 0:
                    # a run that died leaves no alias of a pinned staging buffer behind
                    self.integrator.p = ens.p = np.array(ens.p, copy=True)
            pipe = run.pipe
            self.host_rng_ms = pipe.draw_seconds * 1e3 / max(S, 1)
        elif rng == "philox":
            kT = float(boltzmannConst * temperature)                         # src/ensemble.py:88
            q_state = empty((D, N), dt, dev)
            _lib.call("pbbi_philox_normal", seed, self._position_stream(), int(iter0), int(chain0), D,
                      N, N, float(qStd), None, pot._dt, dev, q_state.data_ptr(), stream)
            if burn_in:
                if jitter:
                    raise ValueError("burn_in and jitter cannot be combined in one call")
                _lib.call("pbbi_hmc_run", pot.handle, self.integrator.method_id, q_state.data_ptr(),
                          mptr, None, None, None, None, N, N, h, L, int(burn_in), flags, seed,
                          int(iter0), int(chain0), kT, stream)
                iter0 = int(iter0) + int(burn_in)
            if jitter and per_chain_steps:
                raise ValueError("jitter (one length per iteration) and per_chain_steps (one per chain) "
                                 "are alternatives")
            if jitter and S > 0:
                if not 0.0 < jitter < 1.0:
                    raise ValueError("jitter must be in (0, 1)")
                u = np.random.RandomState((seed + 0x5EED) % (2 ** 32)).uniform(size=int(iter0) + S)
                steps = np.maximum(1, np.rint(L * (1.0 + jitter * (2.0 * u - 1.0)))).astype(int)
                self.numSteps_used = steps[int(iter0):]
                for i in range(S):
                    _lib.call("pbbi_hmc_run", pot.handle, self.integrator.method_id, q_state.data_ptr(),
                              mptr, samples[i].data_ptr(), momenta[i].data_ptr(), reject[i].data_ptr(),
                              ratio[i].data_ptr(), N, N, h, int(steps[int(iter0) + i]), 1, flags, seed,
                              int(iter0) + i, int(chain0), kT, stream)
            elif per_chain_steps:
                steps = empty((S, N), np.int32, dev)
                _lib.call("pbbi_hmc_run_dyn", pot.handle, self.integrator.method_id, q_state.data_ptr(),
                          mptr, samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(),
                          ratio.data_ptr(), steps.data_ptr(), N, N, h, L, S, flags | _lib.PER_CHAIN_STEPS,
                          seed, int(iter0), int(chain0), kT, stream)
                self.steps = to_numpy(steps)
            else:
                _lib.call("pbbi_hmc_run", pot.handle, self.integrator.method_id, q_state.data_ptr(),
                          mptr, samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(),
                          ratio.data_ptr(), N, N, h, L, S, flags, seed, int(iter0), int(chain0), kT,
                          stream)
        else:
            raise ValueError("rng must be 'numpy' or 'philox'")

        synchronize(dev)
        self.reject_masks = to_numpy(reject).astype(bool)
        self.ratios = to_numpy(ratio)
        self.acceptRate = 1.0 - float(self.reject_masks.mean()) if S > 0 and N > 0 else None
        if S > 0:
            # leave the ensemble where the reference leaves it: q, p alias the final state
            if rng != "numpy":
                self.integrator.q = ens.q = np.empty((D, N))
            if rng != "numpy" or pipe.direct:   # (the last draw lives in a pinned upload buffer)
                self.integrator.p = ens.p = np.empty((D, N))
            self.integrator.q[...] = to_numpy(samples[S - 1])
            self.integrator.p[...] = to_numpy(momenta[S - 1])
        if device_output:
            return samples.permute(1, 2, 0), momenta.permute(1, 2, 0)
        return self._to_dns(samples), self._to_dns(momenta)

    def getSamplesGIST(self, numSamples, temperature, qStd, max_steps=None, seed=None, device_output=False, chain0=0,
                       iter0=0):
        """getSamples with a REVERSIBLE per-chain dynamic trajectory length: the self-tuning no-U-turn sampler
        (GIST; pbbi_hmc_run_gist, include/pbbi.h) -- the "no u-turn sampling" the reference plans
        (references/PhysicsBasedHMC_SoHPC2022_WeekPlan.md:16-17).  Every iteration every chain integrates forward
        until its own U-turn ((q_j - q_0) . p_j < 0, at most max_steps: default 8 x numSteps), draws its
        trajectory length uniformly below that, and accepts with min(1, e^{-dH} tau_f / tau_b [L <= tau_b]), where
        tau_b is the U-turn count seen from the proposal backwards: lengths adapt to the local geometry chain by
        chain and the target stays invariant.  In-kernel draws (rng="philox" counters; draw_f64 honoured);
        Leapfrog; potentials the per-chain-length kernels serve (elementwise D <= 32, dense fp64 D <= 256).  Returns
        (samples, momenta) like getSamples; self.gist_tau holds the (S, 3, N) counts tau_f, L, tau_b,
        self.ratios the full acceptance ratios."""
        pot, ens = self._pot, self.ensemble
        D, N, S = ens.numDimensions, ens.numParticles, int(numSamples)
        if self.integrator.method_id != _lib.LEAPFROG:
            raise ValueError("GIST sampling integrates with Leapfrog")
        seed = self.seed if seed is None else int(seed)
        dev, dt = pot.device, pot.dtype
        stream = stream_ptr(dev)
        kT = float(boltzmannConst * temperature)
        Lmax = int(max_steps) if max_steps else max(8, 8 * int(self.integrator.numSteps))
        samples, momenta = empty((S, D, N), dt, dev), empty((S, D, N), dt, dev)
        reject, ratio = empty((S, N), np.uint8, dev), empty((S, N), dt, dev)
        tau = empty((S, 3, N), np.int32, dev)
        md = self._mass()
        q_state = empty((D, N), dt, dev)
        _lib.call("pbbi_philox_normal", seed, self._position_stream(), int(iter0), int(chain0), D, N, N, float(qStd),
                  None, pot._dt, dev, q_state.data_ptr(), stream)
        flags = self._flags() & (_lib.COMPAT_P_FROM_OLDQ | _lib.BETA_ACCEPT | _lib.DRAW_F64)
        _lib.call("pbbi_hmc_run_gist", pot.handle, q_state.data_ptr(), md.data_ptr() if md is not None else None,
                  samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(), ratio.data_ptr(), tau.data_ptr(), N, N,
                  float(self.stepSize), Lmax, S, flags, seed, int(iter0), int(chain0), kT, stream)
        synchronize(dev)
        self.reject_masks = to_numpy(reject).astype(bool)
        self.ratios = to_numpy(ratio)
        self.gist_tau = to_numpy(tau)
        self.acceptRate = 1.0 - float(self.reject_masks.mean()) if S > 0 and N > 0 else None
        if device_output:
            return samples.permute(1, 2, 0), momenta.permute(1, 2, 0)
        return self._to_dns(samples), self._to_dns(momenta)

    def sampleChunks(self, numSamples, chunk, temperature, qStd, seed=None, chain0=0, iter0=0,
                     spill_dir=None, momenta=False):
        """Generator over a long in-kernel-draw run in chunks of `chunk` iterations (SURVEY 8f row 4:
        at C2 scale 1000 draws are 64 GB, more than one wants resident or returned at once).  The
        chain state stays on the GPU between chunks and the Philox iteration counter continues,
        so the concatenated chunks are bit-identical to ONE getSamples(numSamples, rng="philox")
        call.  Yields (samples, momenta_or_None) per chunk as (D, N, c) device views that are
        OVERWRITTEN by the next chunk -- reduce them (sampleMoments) or copy them before
        advancing.  With spill_dir every chunk is also written as samples_00000.npy, ... in the
        reference's (D, N, c) layout (momenta_XXXXX.npy when momenta=True)."""
        import os
        pot, ens = self._pot, self.ensemble
        D, N = ens.numDimensions, ens.numParticles
        S, chunk = int(numSamples), max(1, int(chunk))
        seed = self.seed if seed is None else int(seed)
        dev, dt = pot.device, pot.dtype
        stream = stream_ptr(dev)
        kT = float(boltzmannConst * temperature)
        flags = self._flags()
        md = self._mass()
        mptr = md.data_ptr() if md is not None else None
        q_state = empty((D, N), dt, dev)
        _lib.call("pbbi_philox_normal", seed, self._position_stream(), int(iter0), int(chain0), D, N, N,
                  float(qStd), None, pot._dt, dev, q_state.data_ptr(), stream)
        c_alloc = min(chunk, max(S, 1))
        samples = empty((c_alloc, D, N), dt, dev)
        mom = empty((c_alloc, D, N), dt, dev) if momenta else None
        reject = empty((c_alloc, N), np.uint8, dev)
        if spill_dir is not None:
            os.makedirs(spill_dir, exist_ok=True)
        done, k, n_rej = 0, 0, 0.0
        while done < S:
            c = min(chunk, S - done)
            _lib.call("pbbi_hmc_run", pot.handle, self.integrator.method_id, q_state.data_ptr(), mptr,
                      samples.data_ptr(), mom.data_ptr() if momenta else None, reject.data_ptr(), None,
                      N, N, float(self.stepSize), self.integrator.numSteps, c, flags, seed,
                      int(iter0) + done, int(chain0), kT, stream)
            n_rej += float(reject[:c].float().sum().item()) if N else 0.0
            s_view = samples[:c].permute(1, 2, 0)
            m_view = mom[:c].permute(1, 2, 0) if momenta else None
            if spill_dir is not None:
                np.save(os.path.join(spill_dir, f"samples_{k:05d}.npy"), self._to_dns(samples[:c]))
                if momenta:
                    np.save(os.path.join(spill_dir, f"momenta_{k:05d}.npy"), self._to_dns(mom[:c]))
            done += c
            k += 1
            self.acceptRate = 1.0 - n_rej / (done * N) if N else None
            yield s_view, m_view

    def sampleChunksGathered(self, numSamples, chunk, temperature, qStd, n_total=None, seed=None, chain0=0, iter0=0,
                             host_stream=None, momenta=False, group=None, rng=None):
        """sampleChunks for a SHARDED ensemble with the collection overlapped (driven by
        distributed.sample_chunks_sharded): this rank's chains are sampled in chunks of `chunk` iterations and
        chunk k's all-gather runs on a side stream while chunk k+1 samples (distributed.OverlappedGather: two
        send and two receive buffers, received in place).  Yields (samples GatheredBlocks, momenta
        GatheredBlocks | None, self) per chunk; a chunk's buffers are overwritten two chunks later.  Both RNG
        modes: "philox" (one pbbi_hmc_run per chunk, global chain index chain0 + n in the counters) and "numpy"
        (the global NumPy stream replayed, this shard's columns kept: host_stream)."""
        from .distributed import OverlappedGather
        from ._device import torch_dtype
        pot, ens = self._pot, self.ensemble
        D, N = ens.numDimensions, ens.numParticles
        S, chunk = int(numSamples), max(1, int(chunk))
        rng = self.rng if rng is None else rng
        seed = self.seed if seed is None else int(seed)
        dev, dt = pot.device, pot.dtype
        stream = stream_ptr(dev)
        n_total = N if n_total is None else int(n_total)
        c_alloc = min(chunk, max(S, 1))
        tdev = f"cuda:{dev}"
        og_s = OverlappedGather((c_alloc, D, N), torch_dtype(dt), tdev, n_total, group)
        og_m = OverlappedGather((c_alloc, D, N), torch_dtype(dt), tdev, n_total, group) if momenta else None
        reject = empty((c_alloc, N), np.uint8, dev)
        md = self._mass()
        mptr = md.data_ptr() if md is not None else None
        if rng == "philox":
            kT = float(boltzmannConst * temperature)
            q_state = empty((D, N), dt, dev)
            _lib.call("pbbi_philox_normal", seed, self._position_stream(), int(iter0), int(chain0), D, N, N,
                      float(qStd), None, pot._dt, dev, q_state.data_ptr(), stream)
            run = None
        elif rng == "numpy":
            run = self._numpy_stream_run(S, temperature, qStd, host_stream)
            q_prev = run.q0
        else:
            raise ValueError("rng must be 'numpy' or 'philox'")
        done, k, n_rej = 0, 0, 0.0
        try:
            while done < S:
                c = min(chunk, S - done)
                sbuf = og_s.local(k)
                mbuf = og_m.local(k) if momenta else None
                if run is None:
                    _lib.call("pbbi_hmc_run", pot.handle, self.integrator.method_id, q_state.data_ptr(), mptr,
                              sbuf.data_ptr(), mbuf.data_ptr() if momenta else None, reject.data_ptr(), None, N, N,
                              float(self.stepSize), self.integrator.numSteps, c, self._flags(), seed,
                              int(iter0) + done, int(chain0), kT, stream)
                else:
                    for i in range(c):
                        run.step(done + i, q_prev, sbuf[i], mbuf[i] if momenta else None, None, reject[i])
                        q_prev = sbuf[i]
                n_rej += float(reject[:c].float().sum().item()) if N else 0.0
                got_s = og_s.submit(k, c)
                got_m = og_m.submit(k, c) if momenta else None
                done += c
                k += 1
                self.acceptRate = 1.0 - n_rej / (done * N) if N else None
                if got_s is not None:
                    yield got_s, got_m, self
            if run is not None:
                run.pipe.finished = True
        finally:
            if run is not None:
                run.pipe.close()
        if S > 0:
            yield og_s.finish(), (og_m.finish() if momenta else None), self

    def adaptStepSize(self, temperature, qStd, target=0.8, iterations=60, seed=None, chain0=0,
                      gamma=0.05, t0=10.0, kappa=0.75):
        """Step-size adaptation by dual averaging (Hoffman & Gelman 2014, Alg. 5) on the
        ENSEMBLE-mean acceptance probability -- SURVEY 8f row 2 (planned in the reference's
        WeekPlan.md:16-17; described in NotesOnParticleBasedHMC.pdf 0.1.2), not a reference
        feature.  Runs `iterations` warm-up iterations with in-kernel draws (one fused launch each;
        only the scalar mean(min(1, ratio)) leaves the GPU, all-reduced over the process group when
        the ensemble is sharded), keeps simulTime fixed (numSteps = int(simulTime/stepSize) follows
        the step), then sets self.stepSize / the integrator's to the averaged step and returns it.
        The warm-up state is discarded."""
        import torch
        pot, ens = self._pot, self.ensemble
        D, N = ens.numDimensions, ens.numParticles
        if N == 0:
            return float(self.stepSize)
        seed = self.seed if seed is None else int(seed)
        # The warm-up draws from its own Philox KEY: the counter holds only 32 iteration bits
        # (include/pbbi.h), so no iteration offset can keep its draws apart from getSamples'.
        seed = (seed ^ WARMUP_SEED_MASK) & 0xFFFFFFFFFFFFFFFF
        dev, dt = pot.device, pot.dtype
        stream = stream_ptr(dev)
        kT = float(boltzmannConst * temperature)
        q_state = empty((D, N), dt, dev)
        ratio = empty((1, N), dt, dev)  # burn-in form of pbbi_hmc_run: no sample slab
        md = self._mass()
        mptr = md.data_ptr() if md is not None else None
        _lib.call("pbbi_philox_normal", seed, self._position_stream(), 0, int(chain0), D, N, N,
                  float(qStd), None, pot._dt, dev, q_state.data_ptr(), stream)
        sharded = torch.distributed.is_available() and torch.distributed.is_initialized()
        h = float(self.stepSize)
        run_flags = self._flags() & ~_lib.COMPAT_P_FROM_OLDQ  # the same kernels and accept test as the sampling run
        mu, hbar, log_hbar = np.log(10.0 * h), 0.0, 0.0
        for m in range(1, int(iterations) + 1):
            L = max(1, int(self.simulTime / h))
            _lib.call("pbbi_hmc_run", pot.handle, self.integrator.method_id, q_state.data_ptr(), mptr,
                      None, None, None, ratio.data_ptr(), N, N, h, L, 1, run_flags, seed,
                      m, int(chain0), kT, stream)
            acc = torch.nan_to_num(torch.clamp(ratio[0].double(), max=1.0), nan=0.0).sum()
            cnt = torch.tensor(float(N), dtype=torch.float64, device=acc.device)
            if sharded:
                both = torch.stack([acc, cnt])
                torch.distributed.all_reduce(both)
                acc, cnt = both[0], both[1]
            alpha = float(acc / cnt)
            hbar = (1.0 - 1.0 / (m + t0)) * hbar + (target - alpha) / (m + t0)
            log_h = mu - np.sqrt(m) / gamma * hbar
            eta = m ** (-kappa)
            log_hbar = eta * log_h + (1.0 - eta) * log_hbar
            h = float(np.exp(log_h))
        h = float(np.exp(log_hbar))
        self.stepSize = self.integrator.stepSize = h
        self.integrator.numSteps = max(1, int(self.simulTime / h))  # what the warm-up itself ran
        return h

    def adaptTrajectoryLength(self, temperature, qStd, iterations=12, max_steps=None, quantile=0.5,
                              seed=None, chain0=0):
        """Trajectory length from the no-U-turn criterion, measured on the ENSEMBLE (the
        reference's WeekPlan.md:16-17 plans "no u-turn sampling"): `iterations` warm-up iterations run with
        PBBI_UTURN_STOP -- every chain integrates until (q - q0).p < 0, its lane masked out from then on,
        at most max_steps (default 8 x the current numSteps) -- and the `quantile` of the chains' U-turn
        step counts over the second half of the warm-up becomes numSteps (simulTime = numSteps*stepSize).
        Stopping at a U-turn is not a reversible move, so these iterations only MEASURE: the recorded run
        afterwards uses the fixed (or jittered / per-chain random) length, which is.  Elementwise
        potentials with D <= 32 and dense Gaussians with D <= 128 (include/pbbi.h); the warm-up state is
        discarded.  Returns simulTime."""
        pot, ens = self._pot, self.ensemble
        D, N = ens.numDimensions, ens.numParticles
        if N == 0:
            return float(self.simulTime)
        seed = ((self.seed if seed is None else int(seed)) ^ WARMUP_SEED_MASK) & 0xFFFFFFFFFFFFFFFF
        dev, dt = pot.device, pot.dtype
        stream = stream_ptr(dev)
        kT = float(boltzmannConst * temperature)
        Lmax = int(max_steps) if max_steps else max(8, 8 * int(self.integrator.numSteps))
        q_state = empty((D, N), dt, dev)
        steps = empty((int(iterations), N), np.int32, dev)
        md = self._mass()
        _lib.call("pbbi_philox_normal", seed, self._position_stream(), 0, int(chain0), D, N, N, float(qStd), None,
                  pot._dt, dev, q_state.data_ptr(), stream)
        _lib.call("pbbi_hmc_run_dyn", pot.handle, self.integrator.method_id, q_state.data_ptr(),
                  md.data_ptr() if md is not None else None, None, None, None, None, steps.data_ptr(), N, N,
                  float(self.stepSize), Lmax, int(iterations),
                  (self._flags() & ~_lib.COMPAT_P_FROM_OLDQ) | _lib.UTURN_STOP, seed, 0, int(chain0), kT, stream)
        st = to_numpy(steps)[int(iterations) // 2:]
        self.uturn_steps = st
        L = int(max(1, round(float(np.quantile(st, quantile)))))
        self.integrator.numSteps = L
        self.simulTime = self.integrator.finalTime = L * float(self.stepSize)
        return self.simulTime

    def sampleMoments(self, samples_dns):
        """Per-dimension (mean, variance) over every draw of every chain, computed on the GPU from
        the (D, N, S) device view that getSamples(device_output=True) returns -- the sample sink
        for runs whose (D, N, S) array is too large to pull to the host."""
        pot = self._pot
        sdn = samples_dns.permute(2, 0, 1)  # back to the (S, D, N) slabs (a view)
        if not sdn.is_contiguous():
            sdn = sdn.contiguous()
        S, D, N = sdn.shape
        mean = empty((D,), pot.dtype, pot.device)
        var = empty((D,), pot.dtype, pot.device)
        _lib.call("pbbi_sample_moments", sdn.data_ptr(), S, D, N, pot._dt, pot.device,
                  mean.data_ptr(), var.data_ptr(), stream_ptr(pot.device))
        return to_numpy(mean).astype(np.float64), to_numpy(var).astype(np.float64)

    def rhat(self, samples_dns):
        """Gelman-Rubin potential scale reduction per dimension, across the ensemble's N chains,
        from the (D, N, S) device view of getSamples(device_output=True) / sampleChunks -- computed
        on the GPU (per-chain Welford moments, then the ensemble moments of those): with
        W = mean_n var_s, B/S = var_n mean_s,  R = sqrt(((S-1)/S W + B/S) / W).  Values near 1 say
        the chains agree; the ensemble design makes this the natural convergence check
        (SURVEY 8f row 4).  Returns a float64 array of D values."""
        pot = self._pot
        sdn = samples_dns.permute(2, 0, 1)
        if not sdn.is_contiguous():
            sdn = sdn.contiguous()
        S, D, N = sdn.shape
        if S < 2 or N < 2:
            raise ValueError("rhat needs at least 2 draws and 2 chains")
        cm, cv = empty((1, D, N), pot.dtype, pot.device), empty((1, D, N), pot.dtype, pot.device)
        stream = stream_ptr(pot.device)
        _lib.call("pbbi_chain_moments", sdn.data_ptr(), S, D, N, pot._dt, pot.device, cm.data_ptr(),
                  cv.data_ptr(), stream)
        w = empty((D,), pot.dtype, pot.device)
        bvar = empty((D,), pot.dtype, pot.device)
        _lib.call("pbbi_sample_moments", cv.data_ptr(), 1, D, N, pot._dt, pot.device, w.data_ptr(), None,
                  stream)                                     # W = mean over chains of the chain variances
        _lib.call("pbbi_sample_moments", cm.data_ptr(), 1, D, N, pot._dt, pot.device, None,
                  bvar.data_ptr(), stream)                    # biased variance over chains of the chain means
        W = to_numpy(w).astype(np.float64)
        B_over_S = to_numpy(bvar).astype(np.float64) * N / (N - 1.0)
        return np.sqrt(((S - 1.0) / S * W + B_over_S) / W)

    def sampleCovariance(self, samples_dns):
        """(mean (D,), covariance (D, D)) over every draw of every chain, computed on the GPU from the
        (D, N, S) device view of getSamples(device_output=True) / sampleChunks (the
        posterior covariance without moving 6.7 GB per 100 draws of config C2 across PCIe)."""
        import torch
        pot = self._pot
        sdn = samples_dns.permute(2, 0, 1)
        if not sdn.is_contiguous():
            sdn = sdn.contiguous()
        S, D, N = sdn.shape
        code, npdt = self._slab_dtype(sdn)
        stream = stream_ptr(pot.device)
        mean = empty((D,), npdt, pot.device)
        _lib.call("pbbi_sample_moments", sdn.data_ptr(), S, D, N, code, pot.device, mean.data_ptr(), None,
                  stream)
        mean64 = mean.double()
        cov = torch.empty((D, D), dtype=torch.float64, device=sdn.device)
        _lib.call("pbbi_sample_covariance", sdn.data_ptr(), S, D, N, code, pot.device, mean64.data_ptr(),
                  cov.data_ptr(), stream)
        return to_numpy(mean64), to_numpy(cov)

    def ess(self, samples_dns, max_lag=32):
        """Effective sample size per dimension of the ensemble's N chains x S draws,
        from autocovariances computed on the GPU (pbbi_chain_moments, pbbi_chain_autocov): the
        multi-chain estimator of Gelman et al. (BDA3 11.5; Stan's ess) -- rho_t = 1 - (W - mean_n
        gamma_t,n) / var+, summed in pairs up to the first negative pair (Geyer), lags <= max_lag <= 32.
        When the sum is cut by max_lag instead (slowly mixing chains) the value is an upper bound;
        self.ess_truncated marks those dimensions.  Returns a float64 array of D values (<= N*S; > N*S
        is reported as is for antithetic chains)."""
        import torch
        pot = self._pot
        sdn = samples_dns.permute(2, 0, 1)
        if not sdn.is_contiguous():
            sdn = sdn.contiguous()
        S, D, N = sdn.shape
        if S < 4 or N < 2:
            raise ValueError("ess needs at least 4 draws and 2 chains")
        T = int(min(max_lag, 32, S - 2))
        code, npdt = self._slab_dtype(sdn)
        stream = stream_ptr(pot.device)
        cm, cv = empty((1, D, N), npdt, pot.device), empty((1, D, N), npdt, pot.device)
        _lib.call("pbbi_chain_moments", sdn.data_ptr(), S, D, N, code, pot.device, cm.data_ptr(),
                  cv.data_ptr(), stream)
        acov = torch.empty((T + 1, D), dtype=torch.float64, device=sdn.device)
        _lib.call("pbbi_chain_autocov", sdn.data_ptr(), cm.data_ptr(), S, D, N, T, code, pot.device,
                  acov.data_ptr(), stream)
        bvar = empty((D,), npdt, pot.device)
        _lib.call("pbbi_sample_moments", cm.data_ptr(), 1, D, N, code, pot.device, None, bvar.data_ptr(),
                  stream)                                      # biased variance over chains of the chain means
        g = to_numpy(acov)                                     # (T+1, D): mean_n gamma_t,n
        W = g[0] * S / (S - 1.0)
        var_plus = W * (S - 1.0) / S + to_numpy(bvar).astype(np.float64) * N / (N - 1.0)
        rho = 1.0 - (W[None, :] - g) / var_plus[None, :]
        rho[0] = 1.0
        ess = np.empty(D)
        self.ess_truncated = np.zeros(D, dtype=bool)
        for d in range(D):
            tau, prev, cut = -1.0, np.inf, False
            for t in range(0, T, 2):
                pair = rho[t, d] + rho[t + 1, d]
                if pair < 0.0:
                    cut = True
                    break
                pair = min(pair, prev)                         # Geyer's initial monotone sequence
                tau += 2.0 * pair
                prev = pair
            self.ess_truncated[d] = not cut
            ess[d] = N * S / max(tau, 1.0 / np.log10(max(N * S, 10)))
        return ess

    @staticmethod
    def _slab_dtype(sdn):
        """(C-ABI dtype code, NumPy dtype) of a device slab tensor."""
        import torch
        if sdn.dtype == torch.float64:
            return _lib.F64, np.float64
        if sdn.dtype == torch.float32:
            return _lib.F32, np.float32
        raise TypeError("sample slabs must be float64 or float32")

    def _to_dns(self, sdn):
        """(S, D, N) device slabs -> host (D, N, S) array via the LDS-tiled transpose kernel."""
        pot = self._pot
        S, D, N = sdn.shape
        if S == 0:
            return np.zeros((D, N, 0))
        out = empty((D, N, S), pot.dtype, pot.device)
        _lib.call("pbbi_transpose_sdn_to_dns", sdn.data_ptr(), out.data_ptr(), S, D, N, pot._dt,
                  pot.device, stream_ptr(pot.device))
        return to_numpy(out).astype(np.float64, copy=False)
