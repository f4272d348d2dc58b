#!/usr/bin/env python3
"""bench.py -- leapfrog-steps x chains / second of the fused ensemble-HMC hot path.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"; per GPU): d = 128 correlated Gaussian with a
dense precision matrix (Sigma = A A^T/D + I, A = RandomState(0) normals),
ensemble = 65 536 chains, fp64, stepSize 0.1, simulTime 1.0 -> L = 10 leapfrog steps.
A "step" is ONE full HMC iteration over the whole ensemble, exactly the body of the
reference's getSamples loop (src/HMC.py:154-179): momentum draw (in-kernel Philox),
L leapfrog steps with L+1 gradient evaluations, both Hamiltonians, Metropolis
accept/reject, and the store of the position AND momentum sample slabs.  Inputs are
resident in HBM when the timed region starts.  With N > 1 every rank owns its own 65 536
chains (weak scaling, C4 = 8 x 65 536); the only collective is the RCCL all-gather of the
final sample slab AFTER the timed region (reported as allgather_ms).

value = K * L * N_total_chains / t, t = max over ranks of the barrier-bracketed wall time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

D = 128
N_PER_GPU = 65536
SETTLE = 100  # untimed clock-settling iterations guaranteed before the measurement (set-up)
STEP = 0.1
SIMUL = 1.0
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix = fp64 vector peak
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


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


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/*_pmc.json, produced by tools/run_profiles.sh + tools/summarize_profiles.py with
    the FETCH_SIZE calibration described there).  None if no summary is present."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f)).get("derived", {})
            if "k_dense_hmc_hbm_bytes_per_launch" in d:
                return d["k_dense_hmc_hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, None


def cpu_baseline(Pm, L, seconds_target=4.0):
    """The oracle (CPU restatement of the reference loop; 'port'), all host threads, on a
    bounded sample of the same workload: same D, L, potential; fewer chains."""
    from oracle import oracle as orc
    pot = orc.pot_gauss_dense(np.zeros(D), Pm)
    # the box's CPU share for one GPU is 16 cores (affinity may show the whole host)
    threads = max(1, min(16, len(os.sched_getaffinity(0)), orc.max_threads()))
    orc.set_threads(threads)
    n_probe = 64 * threads
    q = orc.philox_normal(1, orc.STREAM_POSITION, 0, 0, D, n_probe)
    t0 = time.perf_counter()
    orc.hmc_run_philox(pot, "Leapfrog", q, None, STEP, L, 1, seed=1, want_momenta=True)
    t_probe = time.perf_counter() - t0
    n = int(min(4 * N_PER_GPU, max(n_probe, n_probe * seconds_target / max(t_probe, 1e-3))))
    n -= n % threads
    q = orc.philox_normal(1, orc.STREAM_POSITION, 0, 0, D, n)
    t0 = time.perf_counter()
    orc.hmc_run_philox(pot, "Leapfrog", q, None, STEP, L, 1, seed=1, want_momenta=True)
    dt = time.perf_counter() - t0
    return {"value": n * L / dt, "unit": "leapfrog-steps*chains/s", "cores": threads,
            "kind": "port",
            "sample": f"oracle/pbbi_oracle.c hmc_run_philox, 1 HMC iteration, D={D}, "
                      f"{n} chains, L={L}, OpenMP over chains on {threads} threads, "
                      f"{dt:.1f} s wall = {dt * threads:.0f} core-seconds"}


def main_c3(args):
    """BASELINE config 3: Rosenbrock (a=1, b=100, s=20), d=32, 262 144 chains, fp64, h=0.01,
    L=10.  HBM-bound: 103 B and ~870 flop per step*chain."""
    import torch
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    d, N, h, L = 32, 262144 if args.chains == N_PER_GPU else args.chains, 0.01, 10
    K, W = args.steps, max(args.warmup, 100)
    # default: kick-drift-kick with FMAs (PBBI_KDK_FMA, ~1e-13 from the reference's operation order);
    # --exact-order times the bit-exact velocity-Verlet kernel instead
    flags = _lib.COMPAT_P_FROM_OLDQ | (0 if args.exact_order else _lib.KDK_FMA)
    pot = P.Rosenbrock(d)
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.empty((d, N), dtype=torch.float64, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 0.1, None, _lib.F64, 0,
              q.data_ptr(), stream)
    q += 1.0  # q0 ~ N(1, 0.1^2): trajectories stay finite
    S_alloc = max(K, W, 1)
    samples = torch.empty((S_alloc, d, N), dtype=torch.float64, device="cuda")
    momenta = torch.empty((S_alloc, d, N), dtype=torch.float64, device="cuda")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device="cuda")

    def run(S, it0):
        _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                  momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, S, flags, 7, it0, 0, 1.0, stream)
    # this kernel's iterations are ~80 us: the chip needs a few hundred of them to settle its clock
    # (measured: the same 100 iterations run 10-15 % faster when they follow ~25 ms of the same work)
    for _ in range(3):
        run(W, 0)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(); run(K, W); ev1.record()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    ks = ev0.elapsed_time(ev1) * 1e-3 / K
    bytes_launch = bytes_per_step_chain(d, L) * L * N
    print(json.dumps({
        "metric": "leapfrog-steps*chains/sec; Rosenbrock d=32, ensemble=262144 (config C3)",
        "value": K * L * N / t, "unit": "leapfrog-steps*chains/s", "n_gpus": 1, "steps": K,
        "warmup": 3 * W, "ms_per_step": t * 1e3 / K, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C3: Rosenbrock d=32, 262144 chains, L=10, h=0.01",
                   "integrator_form": "velocity-Verlet, reference operation order (bit-exact)"
                   if args.exact_order else "kick-drift-kick with FMA (PBBI_KDK_FMA)",
                   "accept_rate": 1.0 - float(reject[:K].float().mean().item())},
        "roofline": {"bound": "hbm", "kernel": "k_ros2_hmc<unit mass, D=32> (two lanes per chain)",
                     "achieved": bytes_launch / ks / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_launch / ks / 1e9 / HBM_PEAK_GBS, "traffic": c3_traffic(args),
                     "launch_ms": ks * 1e3}}))


def c3_traffic(args):
    """HBM bytes per launch of the C3 kernel from the committed PMC passes (profiles/r01_pmc_c3.json:
    separate FETCH_SIZE / WRITE_SIZE runs of tools/profile_c3.py, kick-drift-kick form, 262144 chains)."""
    if args.exact_order or args.chains != N_PER_GPU:
        return None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_c3.json")) as f:
            return json.load(f)["derived"]["hbm_bytes_per_launch"]
    except Exception:
        return None


def main_stream(args):
    """Extra: the workspace-streaming chain-per-lane kernels (kernels_stream.hip): Rosenbrock at
    D > 64 or in fp32.  Reports the rate, the algorithmic-bytes roofline fraction (samples in/out
    only, as for C3) and the rate of the kernel's own design traffic (q, v, a read and written
    once per element-step)."""
    import torch
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    d, N, h, L = args.dim, args.chains, 0.01, 10
    f32 = args.dtype == "f32"
    tdt, w, code = (torch.float32, 4, _lib.F32) if f32 else (torch.float64, 8, _lib.F64)
    K, W = args.steps, args.warmup
    import numpy as np
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
    S_alloc = max(K, W, 1)
    samples = torch.empty((S_alloc, d, N), dtype=tdt, device="cuda")
    momenta = torch.empty((S_alloc, d, N), dtype=tdt, device="cuda")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device="cuda")

    flags = _lib.COMPAT_P_FROM_OLDQ | (0 if args.exact_order else _lib.KDK_FMA)

    def run(S, it0):
        _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                  momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, S, flags, 7, it0, 0, 1.0, stream)
    run(W, 0)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(); run(K, W); ev1.record()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    ks = ev0.elapsed_time(ev1) * 1e-3 / K
    bytes_launch = bytes_per_step_chain(d, L, w) * L * N
    design = (6 * L + 12) * w * d * N  # per launch: 6 accesses per element-step + init/energy/output sweeps
    print(json.dumps({
        "metric": f"leapfrog-steps*chains/sec; {args.potential} d={d}, ensemble={N}, {args.dtype}",
        "value": K * L * N / t, "unit": "leapfrog-steps*chains/s", "n_gpus": 1, "steps": K,
        "warmup": W, "ms_per_step": t * 1e3 / K, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.potential} d={d}, {N} chains, L=10, h={h}",
                   "integrator_form": "reference operation order" if args.exact_order
                   else "PBBI_KDK_FMA where a kernel honours it",
                   "accept_rate": 1.0 - float(reject[:K].float().mean().item())},
        "roofline": {"bound": "hbm", "kernel": "k_lane_hmc / k_ros2_hmc (D <= 64, fp64) or k_stream_hmc",
                     "achieved": bytes_launch / ks / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": bytes_launch / ks / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "design_traffic_GBs": design / ks / 1e9, "launch_ms": ks * 1e3}}))


def main_c5(args):
    """BASELINE config 5: d=4096 dense-precision Gaussian, 8192 chains, fp32, h=0.05, L=10:
    L+1 fused MFMA GEMMs per HMC iteration (kernels_big.hip).  MFMA-bound."""
    import torch
    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    d, N, h, L = 4096, 8192 if args.chains == N_PER_GPU else args.chains, 0.05, 10
    K, W = min(args.steps, 20), min(args.warmup, 2)
    pot = P.GaussianDense(None, precision=precision_matrix(d), const=0.0, dtype="float32")
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.empty((d, N), dtype=torch.float32, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 1.0, None, _lib.F32, 0,
              q.data_ptr(), stream)
    S_alloc = max(K, W, 1)
    samples = torch.empty((S_alloc, d, N), dtype=torch.float32, device="cuda")
    momenta = torch.empty((S_alloc, d, N), dtype=torch.float32, device="cuda")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device="cuda")

    def run(S, it0):
        _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                  momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, S, 1, 7, it0, 0, 1.0, stream)
    run(W, 0)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(); run(K, W); ev1.record()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    it_s = ev0.elapsed_time(ev1) * 1e-3 / K
    flops_it = 2.0 * d * d * (L + 1) * N
    print(json.dumps({
        "metric": "leapfrog-steps*chains/sec; d=4096 dense Gaussian fp32, ensemble=8192 (config C5)",
        "value": K * L * N / t, "unit": "leapfrog-steps*chains/s", "n_gpus": 1, "steps": K,
        "warmup": W, "ms_per_step": t * 1e3 / K, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C5: d=4096 dense precision, 8192 chains, fp32, L=10, h=0.05",
                   "accept_rate": 1.0 - float(reject[:K].float().mean().item())},
        "roofline": {"bound": "mfma", "kernel": "k_big_gemm<float, KDK> x (L+1) per iteration",
                     "achieved": flops_it / it_s / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": flops_it / it_s / 1e12 / 157.3, "traffic": None,
                     "iteration_ms": it_s * 1e3}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100,
                    help="untimed iterations first; ~100 (35 ms) lets the chip settle its clock: the "
                         "same timed run measures 4 %% lower after 5 warm-up iterations")
    ap.add_argument("--chains", type=int, default=N_PER_GPU, help="chains per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "stream"],
                    help="c2 (default, the BASELINE metric) or c3 (Rosenbrock d=32, 262144 chains: "
                         "the HBM-bound chain-per-lane kernel; extra, not the headline line)")
    ap.add_argument("--exact-order", action="store_true",
                    help="--workload c3: the bit-exact velocity-Verlet kernel instead of PBBI_KDK_FMA")
    ap.add_argument("--dim", type=int, default=128, help="--workload stream: dimension")
    ap.add_argument("--potential", default="rosenbrock", choices=["rosenbrock", "diag"],
                    help="--workload stream: potential")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="--workload stream")
    args = ap.parse_args()
    if args.workload == "c3":
        return main_c3(args)
    if args.workload == "stream":
        return main_stream(args)
    if args.workload == "c5":
        return main_c5(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE",
                  file=sys.stderr)
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

    import physicsbasedbayesianinference_amd as P
    from physicsbasedbayesianinference_amd import _lib
    from physicsbasedbayesianinference_amd.distributed import gather_samples

    lib = _lib.load()  # raises if the HIP extension is missing: no fallback
    dev = local_rank
    N, K, W = args.chains, args.steps, args.warmup
    L = int(SIMUL / STEP)
    Pm = precision_matrix(D)
    pot = P.GaussianDense(None, precision=Pm, const=0.0, device=dev)
    chain0 = rank * N
    seed = 42
    stream = torch.cuda.current_stream().cuda_stream

    q_state = torch.empty((D, N), dtype=torch.float64, device=f"cuda:{dev}")
    _lib.call("pbbi_philox_normal", seed, _lib.STREAM_POSITION, 0, chain0, D, N, N, 1.0, None,
              _lib.F64, dev, q_state.data_ptr(), stream)
    S_alloc = max(K, W, 1)
    samples = torch.empty((S_alloc, D, N), dtype=torch.float64, device=f"cuda:{dev}")
    momenta = torch.empty((S_alloc, D, N), dtype=torch.float64, device=f"cuda:{dev}")
    reject = torch.empty((S_alloc, N), dtype=torch.uint8, device=f"cuda:{dev}")

    def run(S, iter0):
        _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q_state.data_ptr(), None,
                  samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(), None, N, N, STEP, L,
                  S, _lib.COMPAT_P_FROM_OLDQ, seed, iter0, chain0, 1.0, stream)

    def barrier():
        if world > 1:
            dist.barrier()

    # The chip needs ~30 ms of this work to settle its clock (the same timed 100
    # launches measure 4 % slower right after 5 warm-up launches than after 100).  When the caller
    # asks for fewer than SETTLE warm-up steps, the remainder runs first, as part of the set-up, so
    # that `value` is the steady-state rate; the W warm-up steps and the K timed steps follow as
    # specified.  Reported as "settle_steps".
    settle = max(0, SETTLE - W)
    done = 0
    while done < settle:
        n = min(S_alloc, settle - done)
        run(n, 1 << 20)  # draws from a counter range the measured run never touches
        done += n
    if settle:  # restore the initial state: the warm-up and the timed run start where they always did
        _lib.call("pbbi_philox_normal", seed, _lib.STREAM_POSITION, 0, chain0, D, N, N, 1.0, None,
                  _lib.F64, dev, q_state.data_ptr(), stream)
    if W > 0:
        run(W, 0)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(K, W)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([t], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    accept = 1.0 - float(reject[:K].float().mean().item())

    # sample collection: ONE all-gather of the last slab over RCCL/xGMI, outside the timed loop
    allgather_ms = None
    if world > 1:
        torch.cuda.synchronize()
        barrier()
        g0 = time.perf_counter()
        full = gather_samples(samples[K - 1:K])
        torch.cuda.synchronize()
        allgather_ms = (time.perf_counter() - g0) * 1e3
        assert full.shape == (1, D, N * world)

    if rank == 0:
        total_chains = N * world
        value = K * L * total_chains / t
        kernel_s = dev_ms * 1e-3 / K  # average launch duration from HIP events on the launch stream
        flops_launch = flops_per_step_chain(D, L) * L * N
        bytes_launch = bytes_per_step_chain(D, L) * L * N
        traffic, traffic_src = measured_traffic()
        out = {
            "metric": "leapfrog-steps*chains/sec (node); d=128 Gaussian, ensemble=65536",
            "value": value,
            "unit": "leapfrog-steps*chains/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "settle_steps": settle,
            "ms_per_step": t * 1e3 / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "C2: d=128 dense-precision Gaussian, fp64, L=10 leapfrog steps "
                                   "per HMC iteration, in-kernel Philox momentum + Metropolis",
                       "chains_per_gpu": N, "total_chains": total_chains, "D": D, "L": L,
                       "stepSize": STEP, "parallelism": f"ensemble-sharded x{world}",
                       "accept_rate": accept},
            "roofline": {
                "bound": "mfma", "kernel": "k_dense_hmc<8, full, hmc, zero-mean>",
                "achieved": flops_launch / kernel_s / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": flops_launch / kernel_s / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_flops_per_launch": flops_launch,
                "algorithmic_bytes_per_launch": bytes_launch,
                "launch_ms": kernel_s * 1e3,
                "hbm_algorithmic_GBs": bytes_launch / kernel_s / 1e9,
                "hbm_frac_of_8TBs": bytes_launch / kernel_s / 1e9 / HBM_PEAK_GBS,
            },
        }
        if allgather_ms is not None:
            out["allgather_ms"] = allgather_ms
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Pm, L)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
