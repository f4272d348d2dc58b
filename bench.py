#!/usr/bin/env python3
"""bench.py -- leapfrog-steps x chains / second of the fused ensemble-HMC hot path.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher in the environment (WORLD_SIZE unset) this script starts the N ranks
itself -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
... bench.py <same arguments>` as a child process, before anything touches the GPU -- and exits
with the child's code; launched under torchrun it is one of those ranks.

Headline workload (BASELINE.json configs[1], "C2"; per GPU): d = 128 correlated Gaussian with a
dense precision matrix (Sigma = A A^T/D + I, A = RandomState(0) normals),
ensemble = 65 536 chains, fp64, stepSize 0.1, simulTime 1.0 -> L = 10 leapfrog steps.
A "step" is ONE full HMC iteration over the whole ensemble, exactly the body of the
reference's getSamples loop (src/HMC.py:154-179): momentum draw (in-kernel Philox),
L leapfrog steps with L+1 gradient evaluations, both Hamiltonians, Metropolis
accept/reject, and the store of the position AND momentum sample slabs.  Inputs are
resident in HBM when the timed region starts.  With N > 1 every rank owns its own 65 536
chains (weak scaling, C4 = 8 x 65 536); the only collective is the RCCL all-gather of sample
slabs, measured AFTER the timed region: blocking (allgather_ms, bytes, GB/s) and overlapped with the
sampling of the next chunk (collection.overlapped: the exposed time).

value = K * L * N_total_chains / t, t = max over ranks of the barrier-bracketed wall time of
the K steps that follow EXACTLY W warm-up steps.  `value_steady` is the same measurement repeated
once the chip has run >= SETTLE iterations in all (its clock needs ~30 ms of this work to settle;
DESIGN.md section 5); the two agree when W >= SETTLE.

At N = 1 the same JSON line also carries, under "other_workloads", the lines of BASELINE configs C3
(Rosenbrock d=32, 262 144 chains; PBBI_KDK_FMA and reference operation order), C5 (d=4096 dense,
fp32, 8 192 chains), of a dense Gaussian at d=256 (fp64, the streamed-P kernel) and of C2 through the class API's default rng="numpy" mode, measured after the headline (`--no-extras` skips them; `--workload c3|c5|
stream|parity` prints one of them as its own line instead).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

D = 128
N_PER_GPU = 65536
SETTLE = 100       # iterations after which the C2 clock has settled (35 ms)
SETTLE_C3 = 600    # C3's launches are ~45 us
STEP = 0.1
SIMUL = 1.0
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix = fp64 vector peak
FP32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
UNIT = "leapfrog-steps*chains/s"


def precision_matrix(d):
    A = np.random.RandomState(0).standard_normal((d, d))
    Pm = np.linalg.inv(A @ A.T / d + np.eye(d))
    return 0.5 * (Pm + Pm.T)


def flops_per_step_chain(d, L):
    # gradient mat-vecs (U reuses them) + update + energies
    return 2.0 * d * d * (L + 1) / L + 11.0 * d + 8.0 * d / L


def bytes_per_step_chain(d, L, w=8):
    # read q, read/draw p, write q sample, write p sample, u, accept byte; per step
    return (4.0 * d * w + w + 1) / L


def profile_json(pattern, key):
    """Newest committed profiles/<pattern> whose "derived" block has `key` (HBM bytes per launch
    from separate rocprofv3 --pmc passes, tools/run_profiles.sh)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(f)).get("derived", {})
            if key in d:
                return d[key], os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, None


def roofline(bound, kernel, peak, unit, t_iter, t_iter_steady, executed, algorithmic, what, **extra):
    """One roofline block.  `frac` / `achieved` count the work the launch EXECUTES (`executed`, per
    iteration: what the hardware did, so the fraction is a utilisation and cannot pass 1);
    `frac_algorithmic` / `achieved_algorithmic` count SURVEY 8d's per-iteration figure (`algorithmic`),
    which includes work a fused / carried launch no longer does -- it is printed as a fraction only while
    it stays <= 1 (beyond that it no longer describes a utilisation; the rate is still printed)."""
    scale = 1e12 if unit == "TFLOP/s" else 1e9
    ach, ach_s = executed / t_iter / scale, executed / t_iter_steady / scale
    alg, alg_s = algorithmic / t_iter / scale, algorithmic / t_iter_steady / scale
    key = "flops" if unit == "TFLOP/s" else "bytes"
    out = {"bound": bound, "kernel": kernel, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
           "frac_steady": ach_s / peak, "frac_basis": "executed: " + what,
           "frac_executed": ach / peak,  # (same number; key kept from round 2)
           "achieved_algorithmic": alg, "frac_algorithmic": alg / peak if alg <= peak else None,
           "frac_algorithmic_steady": alg_s / peak if alg_s <= peak else None,
           f"required_{key}_per_iteration": executed, f"algorithmic_{key}_per_iteration": algorithmic,
           "iteration_ms": t_iter * 1e3, "iteration_ms_steady": t_iter_steady * 1e3}
    out.update(extra)
    return out


# ------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(Pm, L, seconds_target=4.0):
    """The oracle (CPU restatement of the reference loop; 'port') on a bounded sample of the same
    workload (same D, L, potential; fewer chains): all host threads, and ONE thread in the
    reference's per-chain order."""
    from oracle import oracle as orc
    pot = orc.pot_gauss_dense(np.zeros(D), Pm)

    def one(threads, seconds):
        orc.set_threads(threads)
        n_probe = 64 * threads
        q = orc.philox_normal(1, orc.STREAM_POSITION, 0, 0, D, n_probe)
        t0 = time.perf_counter()
        orc.hmc_run_philox(pot, "Leapfrog", q, None, STEP, L, 1, seed=1, want_momenta=True)
        t_probe = time.perf_counter() - t0
        n = int(min(4 * N_PER_GPU, max(n_probe, n_probe * seconds / max(t_probe, 1e-3))))
        n -= n % threads
        q = orc.philox_normal(1, orc.STREAM_POSITION, 0, 0, D, n)
        t0 = time.perf_counter()
        orc.hmc_run_philox(pot, "Leapfrog", q, None, STEP, L, 1, seed=1, want_momenta=True)
        dt = time.perf_counter() - t0
        how = (f"OpenMP over chains on {threads} threads" if threads > 1
               else "one thread, per-chain loop order of src/integrator.py:105-120")
        return {"value": n * L / dt, "unit": UNIT, "cores": threads, "kind": "port",
                "sample": f"oracle/pbbi_oracle.c hmc_run_philox, 1 HMC iteration, D={D}, {n} chains, "
                          f"L={L}, {how}, {dt:.1f} s wall = {dt * threads:.0f} core-seconds"}
    # the box's CPU share for one GPU is 16 cores (affinity may show the whole host)
    threads = max(1, min(16, len(os.sched_getaffinity(0)), orc.max_threads()))
    out = one(threads, seconds_target)
    out["single_thread"] = one(1, seconds_target)
    orc.set_threads(threads)
    return out


# ------------------------------------------------------------------------------ timing helper
RETRY_STALLED = False   # set for the embedded extra lines (main): see timed_runs


def timed_runs(run, K, W, settle, barrier=lambda: None, reduce_max=lambda t: t):
    """W warm-up steps, then K timed steps (`first`); then further untimed steps until `settle`
    have run in all, then K timed steps again (`steady`).  run(S, iter0) enqueues S iterations.
    Each timing is (wall seconds max over ranks, HIP-event milliseconds on the launch stream).
    The supplementary lines measured later in the same process (RETRY_STALLED) repeat a timing ONCE when its wall
    time is more than 1.5 x its device time: a second workload in one process meets a single ~65 ms device-wide
    stall outside the kernels' event bracket in one of its first two synchronisations
    (tools/embedded_probe2.py); the headline line never uses this."""
    import torch

    def once(it0):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        run(K, it0)
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        return reduce_max(t), ev0.elapsed_time(ev1)
    done = 0
    if W > 0:
        run(W, 0)
        done = W
    first = once(done)
    done += K
    if RETRY_STALLED and first[0] * 1e3 > 1.5 * first[1] + 1.0:
        first = once(done)
        done += K
    if done < settle:
        run(settle - done, done)
        done = settle
    steady = once(done)
    if RETRY_STALLED and steady[0] * 1e3 > 1.5 * steady[1] + 1.0:
        steady = once(done + K)
    return first, steady


# ------------------------------------------------------------------------------ workloads
def bench_c3(args, exact_order):
    """BASELINE config 3: Rosenbrock (a=1, b=100, s=20), d=32, 262 144 chains, fp64, h=0.01,
    L=10.  HBM-bound: 103 B and ~870 flop per step*chain."""
    import torch
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    d, N, h, L = 32, args.chains_c3, 0.01, 10
    K, W = args.steps, args.warmup
    # default: kick-drift-kick with FMAs (PBBI_KDK_FMA, ~1e-13 from the reference's operation order);
    # exact_order times the bit-exact velocity-Verlet kernel instead
    flags = _lib.COMPAT_P_FROM_OLDQ | (0 if exact_order else _lib.KDK_FMA)
    pot = P.Rosenbrock(d)
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.empty((d, N), dtype=torch.float64, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 0.1, None, _lib.F64, 0,
              q.data_ptr(), stream)
    q += 1.0  # q0 ~ N(1, 0.1^2): trajectories stay finite
    S_alloc = max(K, 1)
    samples = torch.empty((S_alloc, d, N), dtype=torch.float64, device="cuda")
    momenta = torch.empty((S_alloc, d, N), dtype=torch.float64, device="cuda")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device="cuda")

    def run(S, it0):
        while S > 0:  # the slabs hold K iterations; longer (untimed) stretches reuse them
            s = min(S, S_alloc)
            _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                      momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, s, flags, 7, it0, 0, 1.0,
                      stream)
            S, it0 = S - s, it0 + s
    (t, ev), (ts, evs) = timed_runs(run, K, W, SETTLE_C3)
    # pbbi_hmc_run fuses up to 16 consecutive iterations into ONE launch of k_ros2_hmc (the chain stays
    # in registers); ks / kss are per ITERATION, a launch covers K / n_launches of them
    fuse = int(os.environ.get("PBBI_FUSE_ITERS", "16"))
    n_launches = -(-K // max(fuse, 1))
    ks, kss = ev * 1e-3 / K, evs * 1e-3 / K
    bytes_alg = bytes_per_step_chain(d, L) * L * N
    ipl = K / n_launches
    # what a fused launch moves per iteration: both sample slabs and the decision bytes are written, the
    # position is read once per LAUNCH (the chain stays in registers), the momentum is drawn in the kernel
    bytes_req = (2.0 * d * 8 + 1) * N + d * 8.0 * N / ipl
    traffic, src = profile_json("r*_pmc_c3_exact.json" if exact_order else "r*_pmc_c3.json", "hbm_bytes_per_launch")
    instr, _ = profile_json("r*_pmc_c3_exact.json" if exact_order else "r*_pmc_c3.json",
                            "valu_instructions_per_wave_iteration")
    if N != 262144:
        traffic = src = instr = None
    rl = roofline("hbm", "k_ros2_hmc<unit mass, D=32> (two lanes per chain)", HBM_PEAK_GBS, "GB/s", ks, kss,
                  bytes_req, bytes_alg,
                  "q and p sample slabs + decisions written every iteration, q read once per fused launch; "
                  "frac_algorithmic = SURVEY 8d's 4*D*w + w + 1 bytes per chain and iteration",
                  traffic=traffic, traffic_source=src, iterations_per_launch=ipl,
                  launch_ms=ks * 1e3 * ipl, launch_ms_steady=kss * 1e3 * ipl,
                  note="one k_ros2_hmc launch = iterations_per_launch fused HMC iterations; every figure is per "
                       "iteration.  The kernel is bound by fp64 vector-instruction ISSUE, not by HBM (see "
                       "`issue`): the HBM fraction is a proxy")
    if instr:
        # one fp64 (or 64-bit integer multiply) instruction occupies its SIMD for ~4.5 cycles with >= 2 waves
        # resident (tools/ubench/valu_f64_clock.hip); 1024 SIMDs, 2.4 GHz nominal
        waves = 2 * N / 64
        floor_s = instr * waves * 4.5 / (1024 * 2.4e9)
        rl["issue"] = {"valu_instructions_per_wave_iteration": instr, "issue_floor_ms": floor_s * 1e3,
                       "frac_of_issue_floor": floor_s / ks, "frac_of_issue_floor_steady": floor_s / kss,
                       "note": "instructions per wave and iteration from the committed --pmc pass x 4.5 cycles "
                               "per instruction and SIMD at 2.4 GHz"}
    return {
        "metric": "leapfrog-steps*chains/sec; Rosenbrock d=32, ensemble=262144 (config C3)",
        "value": K * L * N / t, "value_steady": K * L * N / ts, "unit": UNIT, "n_gpus": 1, "steps": K,
        "warmup": W, "ms_per_step": t * 1e3 / K, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C3: Rosenbrock d=32, {N} chains, L=10, h=0.01",
                   "integrator_form": "velocity-Verlet, reference operation order (bit-exact)"
                   if exact_order else "kick-drift-kick with FMA (PBBI_KDK_FMA)",
                   "accept_rate": 1.0 - float(reject[:K].float().mean().item())},
        "roofline": rl}


def bench_stream(args):
    """Extra: elementwise potentials at any D / dtype (separable, multi-lane Rosenbrock and the
    workspace-streaming kernels).  Reports the rate, the algorithmic-bytes roofline fraction
    (samples in/out only, as for C3) and the rate of the workspace kernels' design traffic."""
    import torch
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    d, N, h, L = args.dim, args.chains, 0.01, 10
    f32 = args.dtype == "f32"
    tdt, w, code = (torch.float32, 4, _lib.F32) if f32 else (torch.float64, 8, _lib.F64)
    K, W = args.steps, args.warmup
    if args.potential == "diag":
        rs = np.random.RandomState(0)
        pot = P.GaussianDiag(rs.standard_normal(d), prec=rs.uniform(0.5, 2.0, d), const=0.0,
                             dtype="float32" if f32 else "float64")
        h = 0.1
    else:
        pot = P.Rosenbrock(d, dtype="float32" if f32 else "float64")
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.empty((d, N), dtype=tdt, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 0.1, None, code, 0,
              q.data_ptr(), stream)
    q += 1.0
    S_alloc = max(K, 1)
    samples = torch.empty((S_alloc, d, N), dtype=tdt, device="cuda")
    momenta = torch.empty((S_alloc, d, N), dtype=tdt, device="cuda")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device="cuda")
    flags = _lib.COMPAT_P_FROM_OLDQ | (0 if args.exact_order else _lib.KDK_FMA)

    def run(S, it0):
        while S > 0:
            s = min(S, S_alloc)
            _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                      momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, s, flags, 7, it0, 0, 1.0,
                      stream)
            S, it0 = S - s, it0 + s
    (t, ev), (ts, evs) = timed_runs(run, K, W, max(W + K, SETTLE))
    ks, kss = ev * 1e-3 / K, evs * 1e-3 / K
    bytes_alg = bytes_per_step_chain(d, L, w) * L * N
    # the least any kernel must move per iteration of a run with in-kernel draws: both sample slabs and the
    # decision bytes written (fused kernels move exactly this plus one position read per launch; unfused
    # ones also read the position every iteration)
    bytes_req = (2.0 * d * w + 1) * N
    design = (6 * L + 12) * w * d * N  # workspace kernels: 6 accesses per element-step + init/energy/output sweeps
    return {
        "metric": f"leapfrog-steps*chains/sec; {args.potential} d={d}, ensemble={N}, {args.dtype}",
        "value": K * L * N / t, "value_steady": K * L * N / ts, "unit": UNIT, "n_gpus": 1, "steps": K,
        "warmup": W, "ms_per_step": t * 1e3 / K, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.potential} d={d}, {N} chains, L=10, h={h}",
                   "integrator_form": "reference operation order" if args.exact_order
                   else "PBBI_KDK_FMA where a kernel honours it",
                   "accept_rate": 1.0 - float(reject[:K].float().mean().item())},
        "roofline": roofline("hbm", "k_lane_hmc / k_ros2_hmc / k_sepn / k_rosg / k_stream_hmc", HBM_PEAK_GBS,
                             "GB/s", ks, kss, bytes_req, bytes_alg,
                             "the q and p sample slabs and the decision bytes an iteration must write (lower "
                             "bound of any kernel's traffic); frac_algorithmic = SURVEY 8d's 4*D*w + w + 1",
                             traffic=None, design_traffic_GBs=design / ks / 1e9, launch_ms=ks * 1e3)}


def bench_dense(args):
    """Extra: a dense-precision Gaussian at any D (fp64): the register-resident MFMA kernel up to D = 128, the same
    kernel with P streamed through LDS up to D = 256, the GEMM path beyond.  The executed mat-vec count comes from
    pbbi_describe_run (carried gradient or not)."""
    import ctypes
    import torch
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    d, N, h, L = args.dim, args.chains, STEP, int(SIMUL / STEP)
    K, W = args.steps, args.warmup
    pot = P.GaussianDense(None, precision=precision_matrix(d), const=0.0)
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.empty((d, N), dtype=torch.float64, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 1.0, None, _lib.F64, 0, q.data_ptr(), stream)
    S_alloc = max(K, 1)
    samples = torch.empty((S_alloc, d, N), dtype=torch.float64, device="cuda")
    momenta = torch.empty((S_alloc, d, N), dtype=torch.float64, device="cuda")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device="cuda")

    def run(S, it0):
        while S > 0:
            s = min(S, S_alloc)
            _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                      momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, s, 1, 7, it0, 0, 1.0, stream)
            S, it0 = S - s, it0 + s
    (t, ev), (ts, evs) = timed_runs(run, K, W, max(W + K, SETTLE))
    buf = ctypes.create_string_buffer(1024)
    _lib.call("pbbi_describe_run", pot.handle, _lib.LEAPFROG, N, N, L, K, 1, buf, len(buf))
    route = buf.value.decode()
    carried = "carried between iterations: yes" in route
    ks, kss = ev * 1e-3 / K, evs * 1e-3 / K
    flops_alg = flops_per_step_chain(d, L) * L * N
    flops_exec = (2.0 * d * d * ((L + 1.0 / K) if carried else L + 1) + 11.0 * d * L + 8.0 * d) * N
    traffic = src = None
    if d == 256 and N == 65536 and carried:  # the committed --pmc passes of tools/run_profiles_dstream.sh
        traffic, src = profile_json("r*_pmc_dense_d256.json", "hbm_bytes_per_iteration")
    return {
        "metric": f"leapfrog-steps*chains/sec; dense Gaussian d={d}, ensemble={N}, f64",
        "value": K * L * N / t, "value_steady": K * L * N / ts, "unit": UNIT, "n_gpus": 1, "steps": K,
        "warmup": W, "ms_per_step": t * 1e3 / K, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"dense precision d={d}, {N} chains, L={L}, h={h}", "route": route,
                   "accept_rate": 1.0 - float(reject[:K].float().mean().item())},
        "roofline": roofline("mfma", route.split(";")[0], FP64_MFMA_PEAK_TFLOPS, "TFLOP/s", ks, kss, flops_exec,
                             flops_alg, "the gradient mat-vecs performed (see config.route) + updates + energies, "
                             "counted at the true D (rows padded to the kernel's tile are not work)", traffic=traffic,
                             **({"traffic_source": src, "traffic_note": "HBM bytes per iteration (FETCH_SIZE + WRITE_SIZE, "
                                 "separate --pmc passes); P itself is served by L2"} if src else {}))}


def bench_gist(args):
    """Extra: the self-tuning no-U-turn sampler (pbbi_hmc_run_gist) on the C2 target.  One iteration integrates
    tau_f + L + tau_b leapfrog steps per chain (forward search, proposal, backward search) in six launches; the line
    counts the steps the chains' OWN lengths call for (the masked kernels run each 16-chain tile to its longest
    chain) and reports the fixed-length headline kernel beside it."""
    import torch
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    d, N, h = args.dim, args.chains, STEP
    K, W, Lmax = min(args.steps, 20), min(args.warmup, 5), 64
    if args.potential == "diag":   # elementwise potential, D <= 32: one fused launch per iteration (k_lane_gist_hmc)
        rs = np.random.RandomState(0)
        pot = P.GaussianDiag(rs.standard_normal(d), prec=rs.uniform(0.5, 2.0, d), const=0.0)
    else:
        pot = P.GaussianDense(None, precision=precision_matrix(d), const=0.0)
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.empty((d, N), dtype=torch.float64, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 1.0, None, _lib.F64, 0, q.data_ptr(), stream)
    samples = torch.empty((K, d, N), dtype=torch.float64, device="cuda")
    momenta = torch.empty((K, d, N), dtype=torch.float64, device="cuda")
    reject = torch.empty((K, N), dtype=torch.uint8, device="cuda")
    tau = torch.empty((K, 3, N), dtype=torch.int32, device="cuda")

    def run(S, it0):
        while S > 0:
            s = min(S, K)
            _lib.call("pbbi_hmc_run_gist", pot.handle, q.data_ptr(), None, samples.data_ptr(), momenta.data_ptr(),
                      reject.data_ptr(), None, tau.data_ptr(), N, N, h, Lmax, s, 1, 7, it0, 0, 1.0, stream)
            S, it0 = S - s, it0 + s
    (t, ev), (ts, evs) = timed_runs(run, K, W, W + K)
    steps = tau.double().sum(dim=1).mean().item()          # tau_f + L + tau_b per chain and iteration
    tf, Ld, tb = (tau[:, i].double().mean().item() for i in range(3))
    return {
        "metric": f"leapfrog-steps*chains/sec; GIST (self-tuning no-U-turn) on a dense Gaussian d={d}, ensemble={N}",
        "value": K * steps * N / t, "value_steady": K * steps * N / ts, "unit": UNIT, "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": t * 1e3 / K, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"GIST, {'diagonal' if args.potential == 'diag' else 'dense precision'} Gaussian d={d}, {N} chains, "
                               f"h={h}, L_max={Lmax}",
                   "form": "composed (PBBI_GIST_COMPOSED)" if os.environ.get("PBBI_GIST_COMPOSED") else
                           ("one fused launch per iteration" if args.potential == "diag" and d <= 32 else
                            "three masked launches + small kernels per iteration"),
                   "mean_tau_forward": tf, "mean_length": Ld, "mean_tau_backward": tb,
                   "leapfrog_steps_per_chain_and_iteration": steps,
                   "accept_rate": 1.0 - float(reject.float().mean().item()),
                   "note": "value counts the steps of the three masked trajectories of an iteration; independent "
                           "draws per unit time are what the sampler is for (ESS), not this rate"}}


def bench_c5(args):
    """BASELINE config 5: d=4096 dense-precision Gaussian, 8192 chains, fp32, h=0.05, L=10:
    L+1 fused MFMA GEMMs per HMC iteration (kernels_big.hip).  MFMA-bound."""
    import torch
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    d, N, h, L = 4096, args.chains_c5, 0.05, 10
    K, W = min(args.steps, 20), min(args.warmup, 5)
    pot = P.GaussianDense(None, precision=precision_matrix(d), const=0.0, dtype="float32")
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.empty((d, N), dtype=torch.float32, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 1.0, None, _lib.F32, 0,
              q.data_ptr(), stream)
    S_alloc = max(K, 1)
    samples = torch.empty((S_alloc, d, N), dtype=torch.float32, device="cuda")
    momenta = torch.empty((S_alloc, d, N), dtype=torch.float32, device="cuda")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device="cuda")

    def run(S, it0):
        while S > 0:
            s = min(S, S_alloc)
            _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                      momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, s, 1, 7, it0, 0, 1.0, stream)
            S, it0 = S - s, it0 + s
    (t, ev), (ts, evs) = timed_runs(run, K, W, W + K)  # 25 ms iterations: settled after two
    it_s, it_ss = ev * 1e-3 / K, evs * 1e-3 / K
    flops_it = 2.0 * d * d * (L + 1) * N
    # inside pbbi_hmc_run the first GEMM of an iteration is replaced by an elementwise pass over the gradient
    # the previous iteration kept: L GEMMs are executed where SURVEY 8d's figure counts L + 1
    carried = os.environ.get("PBBI_NO_CARRY") is None
    flops_exec = 2.0 * d * d * (L if carried else L + 1) * N
    traffic = src = None
    if N == 8192:
        rd, src = profile_json("r*_pmc_c5.json", "fetch_bytes_per_launch_x2")
        wr, _ = profile_json("r*_pmc_c5.json", "write_bytes_per_launch")
        if rd is not None and wr is not None:
            traffic = (rd + wr) * (L if carried else L + 1)  # counters are per GEMM launch
    return {
        "metric": "leapfrog-steps*chains/sec; d=4096 dense Gaussian fp32, ensemble=8192 (config C5)",
        "value": K * L * N / t, "value_steady": K * L * N / ts, "unit": UNIT, "n_gpus": 1, "steps": K,
        "warmup": W, "ms_per_step": t * 1e3 / K, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"C5: d=4096 dense precision, {N} chains, fp32, L=10, h=0.05",
                   "accept_rate": 1.0 - float(reject[:K].float().mean().item())},
        "roofline": roofline("mfma", "k_big_gemm_wide<256x128x16, KDK> x L per iteration (+ k_big_first_kick on "
                             "the carried gradient)", FP32_MFMA_PEAK_TFLOPS, "TFLOP/s", it_s, it_ss, flops_exec,
                             flops_it,
                             "the L gradient GEMMs an iteration performs (the first of SURVEY 8d's L + 1 is "
                             "replaced by a pass over the gradient the previous iteration kept; bit-identical "
                             "samples); frac_algorithmic counts L + 1" if carried else "L + 1 gradient GEMMs",
                             traffic=traffic, traffic_source=src,
                             traffic_note="HBM bytes per iteration = (FETCH_SIZE x 2 KiB + WRITE_SIZE x 1 KiB) of "
                                          "one k_big_gemm_wide launch x GEMMs per iteration (separate --pmc passes)")}


def bench_parity(args):
    """The drop-in's DEFAULT mode (rng="numpy"): the reference's NumPy RandomState stream is drawn on
    the host (D*N normals + N uniforms per iteration, src/ensemble.py:88-91, src/HMC.py:168),
    uploaded, and consumed by pbbi_hmc_iter -- through the class API, HMC.getSamples, wall time of
    the whole call (host draws, PCIe and D2H of the (D,N,S) result included).  Not the headline:
    it measures what a reference user gets without changing a line."""
    import torch
    from scipy.constants import k as kB
    import physicsbasedbayesianinference_amd as P
    N, K = args.chains, args.steps
    L = int(SIMUL / STEP)
    pot = P.GaussianDense(None, precision=precision_matrix(D), const=0.0)
    np.random.seed(42)
    hmc = P.HMC(P.Ensemble(D, N), SIMUL, STEP, None, potential=pot, verbose=False)
    hmc.getSamples(min(args.warmup, 3), 1 / kB, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hmc.getSamples(K, 1 / kB, 1.0, device_output=True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    return {
        "metric": "leapfrog-steps*chains/sec; d=128 Gaussian, class API, rng='numpy' (reference parity mode)",
        "value": K * L * N / t, "unit": UNIT, "n_gpus": 1, "steps": K, "warmup": min(args.warmup, 3),
        "ms_per_step": t * 1e3 / K, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"C2 through HMC.getSamples(rng='numpy'): D={D}, {N} chains, L={L}",
                   "accept_rate": hmc.acceptRate,
                   "host_rng_ms_per_step": getattr(hmc, "host_rng_ms", None),
                   "note": "wall time of the whole getSamples call: NumPy legacy-stream draws on the host "
                           "(libpbbi_host.so: the same stream bit for bit, transform spread over the "
                           "cores) + H2D + kernels; bounded by the host generator, not the GPU"}}


def bench_c2(args, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib

    _lib.load()  # raises if the HIP extension is missing: no fallback
    dev = local_rank
    N, K, W = args.chains, args.steps, args.warmup
    L = int(SIMUL / STEP)
    Pm = precision_matrix(D)
    pot = P.GaussianDense(None, precision=Pm, const=0.0, device=dev)
    chain0 = rank * N
    seed = 42
    stream = torch.cuda.current_stream().cuda_stream
    # --draw f64: momenta from the double-precision draw of the RNG contract (PBBI_DRAW_F64, include/pbbi.h),
    # the counterpart of the reference's float64 normals; default: the single-precision draw
    f64 = getattr(args, "draw", "f32") == "f64"
    run_flags = _lib.COMPAT_P_FROM_OLDQ | (_lib.DRAW_F64 if f64 else 0)

    q_state = torch.empty((D, N), dtype=torch.float64, device=f"cuda:{dev}")
    _lib.call("pbbi_philox_normal", seed, _lib.STREAM_POSITION, 0, chain0, D, N, N, 1.0, None,
              _lib.F64, dev, q_state.data_ptr(), stream)
    S_alloc = max(K, 1)
    samples = torch.empty((S_alloc, D, N), dtype=torch.float64, device=f"cuda:{dev}")
    momenta = torch.empty((S_alloc, D, N), dtype=torch.float64, device=f"cuda:{dev}")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device=f"cuda:{dev}")

    def run(S, iter0):
        while S > 0:
            s = min(S, S_alloc)
            _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q_state.data_ptr(), None,
                      samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(), None, N, N, STEP, L,
                      s, run_flags, seed, iter0, chain0, 1.0, stream)
            S, iter0 = S - s, iter0 + s

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_max(t):
        if world > 1:
            tt = torch.tensor([t], dtype=torch.float64, device=f"cuda:{dev}")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return t
    (t, dev_ms), (ts, dev_ms_s) = timed_runs(run, K, W, SETTLE, barrier, reduce_max)
    accept = 1.0 - float(reject[:K].float().mean().item())

    # sample collection (outside the timed loop): (1) ONE blocking all-gather of a chunk of slabs over
    # RCCL/xGMI, received in place; (2) the same chunks collected WHILE the next chunk samples
    # (distributed.OverlappedGather: side stream, two send / two receive buffers) against the same sampling
    # without any collection -- the difference is what the collection costs when it is overlapped
    collect = None
    if world > 1:
        from physicsbasedbayesianinference_amd.distributed import OverlappedGather, gather_blocks
        c = max(1, min(K, 8))                       # slabs per chunk: 8 x 64 MiB per rank at C2
        n_chunks = 4
        torch.cuda.synchronize()
        barrier()
        g0 = time.perf_counter()
        blk = gather_blocks(samples[:c], n_total=N * world)
        torch.cuda.synchronize()
        allgather_s = time.perf_counter() - g0
        assert blk.blocks.shape == (world, c, D, N)
        gathered_bytes = blk.blocks.numel() * 8
        del blk

        og = OverlappedGather((c, D, N), torch.float64, f"cuda:{dev}", N * world)

        def chunks(gather):
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for k in range(n_chunks):
                buf = og.local(k)                    # the same two slab buffers with and without the collection
                _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q_state.data_ptr(), None, buf.data_ptr(),
                          None, reject.data_ptr(), None, N, N, STEP, L, c, run_flags, seed, 10 ** 6 + k * c,
                          chain0, 1.0, stream)
                if gather:
                    og.submit(k, c)
            if gather:
                og.finish()
            torch.cuda.synchronize()
            barrier()
            return reduce_max(time.perf_counter() - t0)
        chunks(True)                                 # (first use: communicator / buffer warm-up)
        t_over = chunks(True)
        t_plain = chunks(False)
        collect = {"allgather_ms": allgather_s * 1e3, "allgather_bytes_received_per_rank": gathered_bytes,
                   "allgather_GBs_per_rank": gathered_bytes / allgather_s / 1e9,
                   "overlapped": {"chunks": n_chunks, "iterations_per_chunk": c,
                                  "sampling_only_ms": t_plain * 1e3, "sampling_with_overlapped_gather_ms": t_over * 1e3,
                                  "exposed_collection_ms": (t_over - t_plain) * 1e3,
                                  "note": "chunk k's all_gather_into_tensor runs on a side stream while chunk k+1 "
                                          "samples; received in place into (world, c, D, N) blocks"}}
    if rank != 0:
        return None
    total_chains = N * world
    kernel_s = dev_ms * 1e-3 / K  # average launch duration from HIP events on the launch stream
    kernel_ss = dev_ms_s * 1e-3 / K
    flops_launch = flops_per_step_chain(D, L) * L * N
    bytes_launch = bytes_per_step_chain(D, L) * L * N
    traffic, traffic_src = profile_json("r*_pmc.json", "k_dense_hmc_hbm_bytes_per_iteration")
    # pbbi_hmc_run carries the gradient of the chain's position from one iteration to the next (an
    # accepted chain starts from the point whose gradient the last mat-vec just formed, a rejected one from
    # the point it started at) and covers up to PBBI_DENSE_FUSE (64) iterations per launch: L mat-vecs per
    # iteration are EXECUTED where SURVEY 8d's algorithmic figure counts L + 1.  Both rates are reported.
    carried = os.environ.get("PBBI_NO_CARRY") is None
    fuse_max = max(1, int(os.environ.get("PBBI_DENSE_FUSE", "64"))) if carried else 1
    # a timed run = one pbbi_hmc_run of K iterations in equal fused launches (the first iteration of the run
    # forms g(q_0) inside its launch: L + 1 mat-vecs once per run, counted below)
    n_launch = -(-K // fuse_max) if (fuse_max > 1 and K > 1) else K
    fuse = K / n_launch
    matvecs = (L + 1.0 / K) if carried else L + 1  # per iteration, averaged over the timed run
    flops_exec = (2.0 * D * D * matvecs + 11.0 * D * L + 8.0 * D) * N
    # both sample slabs + decisions written; the carried gradient read and g(q_new) written; the position is
    # read once per launch (the momentum is drawn in the kernel: SURVEY 8d's figure counts a read for it)
    bytes_exec = ((2.0 * D * 8 + 1) * N + D * 8.0 * N / fuse + 2.0 * D * 8 * N) if carried else bytes_launch
    out = {
        "metric": "leapfrog-steps*chains/sec (node); d=128 Gaussian, ensemble=65536",
        "value": K * L * total_chains / t,
        "unit": UNIT,
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": t * 1e3 / K,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "value_steady": K * L * total_chains / ts,
        "ms_per_step_steady": ts * 1e3 / K,
        "steady_after_steps": max(SETTLE, W + K),
        "config": {"workload": "C2: d=128 dense-precision Gaussian, fp64, L=10 leapfrog steps "
                               "per HMC iteration, in-kernel Philox momentum + Metropolis",
                   "chains_per_gpu": N, "total_chains": total_chains, "D": D, "L": L,
                   "stepSize": STEP, "parallelism": f"ensemble-sharded x{world}",
                   "accept_rate": accept,
                   "draw": "f64 (PBBI_DRAW_F64: double-precision Box-Muller, bit-identical to the oracle)" if f64
                   else "f32 (single-precision Box-Muller on the transcendental unit; momenta only)"},
        "roofline": roofline(
            "mfma", "k_dense_hmc<8, full, hmc, zero-mean" + (", carried gradient, fused iterations>" if carried else ">"),
            FP64_MFMA_PEAK_TFLOPS, "TFLOP/s", kernel_s, kernel_ss, flops_exec, flops_launch,
            ("the L gradient mat-vecs + updates + energies an iteration performs (the gradient at the chain's "
             "position is carried between iterations, bit-identical samples); frac_algorithmic = SURVEY 8d's count "
             "with L + 1 mat-vecs") if carried else "L + 1 gradient mat-vecs + updates + energies (SURVEY 8d)",
            traffic=traffic, traffic_source=traffic_src,
            algorithmic_bytes_per_iteration=bytes_launch, required_bytes_per_iteration=bytes_exec,
            iterations_per_launch=fuse, launch_ms=kernel_s * 1e3 * fuse, launch_ms_steady=kernel_ss * 1e3 * fuse,
            hbm_required_GBs=bytes_exec / kernel_s / 1e9,
            hbm_frac_of_8TBs=bytes_exec / kernel_s / 1e9 / HBM_PEAK_GBS),
    }
    if collect is not None:
        out["allgather_ms"] = collect["allgather_ms"]
        out["collection"] = collect
    del samples, momenta, reject   # (left in torch's cache: the f64-draw line re-uses the same blocks)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(Pm, L)
    return out


# ------------------------------------------------------------------------------ launcher
def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as a child torchrun (nothing in this
    process has touched the GPU yet), pass its output through and return its exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed iterations before the timed ones; `value` is measured after exactly "
                         "this many (~100 = 35 ms lets the chip settle its clock; value_steady reports "
                         "the settled rate whatever W is)")
    ap.add_argument("--chains", type=int, default=N_PER_GPU, help="chains per GPU (C2, stream, parity)")
    ap.add_argument("--chains-c3", type=int, default=262144)
    ap.add_argument("--chains-c5", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1: skip the C3 / C5 lines embedded under other_workloads")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "stream", "parity", "dense", "gist"],
                    help="c2 (default, the BASELINE metric, with C3/C5 embedded at N = 1); c3 / c5 / "
                         "stream / parity print that workload's own line")
    ap.add_argument("--exact-order", action="store_true",
                    help="--workload c3 / stream: the bit-exact velocity-Verlet kernels instead of PBBI_KDK_FMA")
    ap.add_argument("--dim", type=int, default=128, help="--workload stream / dense: dimension")
    ap.add_argument("--potential", default="rosenbrock", choices=["rosenbrock", "diag"],
                    help="--workload stream: potential")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="--workload stream")
    ap.add_argument("--draw", default="f32", choices=["f32", "f64"],
                    help="--workload c2: precision of the in-kernel momentum draw (f64 = PBBI_DRAW_F64)")
    ap.add_argument("--rng", default="philox", choices=["philox", "numpy"],
                    help="--workload c2 --rng numpy = --workload parity: C2 through the class API's default "
                         "mode (the reference's NumPy stream drawn on the host)")
    args = ap.parse_args()
    if args.rng == "numpy":
        if args.workload not in ("c2", "parity"):
            ap.error("--rng numpy goes with --workload c2")
        args.workload = "parity"
    if args.steps < 1 or args.warmup < 0 or args.gpus < 1:
        ap.error("--steps >= 1, --warmup >= 0, --gpus >= 1")

    if args.workload != "c2":
        fn = {"c3": lambda: bench_c3(args, args.exact_order), "c5": lambda: bench_c5(args), "dense": lambda: bench_dense(args), "gist": lambda: bench_gist(args),
              "stream": lambda: bench_stream(args), "parity": lambda: bench_parity(args)}[args.workload]
        print(json.dumps(fn()))
        return 0

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks",
              file=sys.stderr)
        return 2
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    # rehearsal switch for a one-GPU box: PBBI_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses
    # gloo, to exercise the world_size > 1 code path without RCCL (numbers are then meaningless)
    rehearse = os.environ.get("PBBI_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    try:
        out = bench_c2(args, rank, world, local_rank)
        if rank == 0 and world == 1 and not args.no_extras:
            extras = {}
            def c2_f64():
                a2 = argparse.Namespace(**vars(args))
                a2.draw, a2.no_cpu_baseline = "f64", True
                return bench_c2(a2, rank, world, local_rank)
            global RETRY_STALLED
            RETRY_STALLED = True
            for name, fn in (("c2_draw_f64", c2_f64),
                             ("c3_kdk_fma", lambda: bench_c3(args, False)),
                             ("c3_exact_order", lambda: bench_c3(args, True)),
                             ("c5", lambda: bench_c5(args)),
                             # 128 < D <= 256 (fp64): the dense kernel with P streamed through LDS (kernels_dstream.hip)
                             ("dense_d256", lambda: bench_dense(argparse.Namespace(
                                 dim=256, chains=args.chains, steps=args.steps, warmup=args.warmup))),
                             # the drop-in's default mode (the reference's NumPy stream, bit-exact): 30 iterations
                             ("c2_class_api_rng_numpy", lambda: bench_parity(argparse.Namespace(
                                 chains=args.chains, steps=min(args.steps, 30), warmup=3)))):
                try:
                    extras[name] = fn()
                except Exception as e:  # the headline line must still print
                    extras[name] = {"error": f"{type(e).__name__}: {e}"}
                # hand the slabs back and let the driver finish unmapping them: measured (tools/embedded_probe.py),
                # the first launches after freeing ~13 GB stall on the host side for ~65 ms otherwise
                torch.cuda.empty_cache()
                torch.cuda.synchronize()
                time.sleep(0.3)
            out["other_workloads"] = extras
        if rank == 0:
            if rehearse:
                out["rehearsal"] = "gloo, every rank on cuda:0: the numbers are meaningless"
            print(json.dumps(out))
            sys.stdout.flush()
    finally:
        if world > 1:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
