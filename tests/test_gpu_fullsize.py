"""Full-size oracle spot checks of the entry points bench.py times (-m gpu), and a reject-branch
stress of the dense MFMA kernel against the oracle.

bench.py times `pbbi_hmc_run` (in-kernel Philox draws) at BASELINE's sizes:
  C2  D=128 dense Gaussian,  N=65 536, fp64   -> k_dense_hmc (MODE 0, zero mean, FULL)
  C3  Rosenbrock D=32,       N=262 144, fp64  -> k_ros2_hmc, PBBI_KDK_FMA and reference order
  C5  D=4096 dense Gaussian, N=8 192,  fp32   -> k_big_gemm x (L+1)
The oracle cannot run those sizes in seconds, but chains are independent (src/integrator.py:73,
src/HMC.py:109-115,168-176): a few hundred chains spread over the first, last and random tiles
of the FULL-SIZE launch are replayed in the oracle from the draws `pbbi_philox_normal` /
`pbbi_philox_uniform` return for the same counters (include/pbbi.h: bit-identical to the in-kernel
draws).  That checks the grid-dependent indexing of the timed kernels at the size they are timed
at, which the small-N tests cannot.  Each case also runs with N - 5 chains (ragged last tile).

Tolerances: reject masks equal; q, p scaled error <= 1e-11 (dense MFMA kernel: summation order),
bit-exact (Rosenbrock, reference operation order), <= 1e-12 (PBBI_KDK_FMA), <= 2e-4 (C5: fp32
kernel against the fp64 oracle; decisions compared where |log u - log ratio| > 1e-2).
Reference behaviour matched: src/HMC.py:154-179.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from test_gpu_parity import device_normal, device_uniform, scaled_err

pytestmark = pytest.mark.gpu

GROUP = 16  # chains per replayed group


@pytest.fixture(scope="module")
def P():
    import physicsbasedbayesianinference_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def lib():
    from physicsbasedbayesianinference_amd import _lib
    _lib.load()
    return _lib


def chain_groups(N, n_groups, tile, seed):
    """Starts of GROUP-chain runs: the first tile, the last chains of the ensemble, the two sides
    of the last tile boundary, and random starts anywhere (they straddle tile boundaries)."""
    rs = np.random.RandomState(seed)
    starts = {0, N - GROUP, max(0, (N - 1) // tile * tile - GROUP // 2)}
    while len(starts) < n_groups:
        starts.add(int(rs.randint(0, N - GROUP)))
    return sorted(starts)


def device_normal_dtype(lib, seed, stream, it, chain0, D, N, scale, dtype):
    from physicsbasedbayesianinference_amd._device import empty, stream_ptr, to_numpy
    out = empty((D, N), dtype, 0)
    lib.call("pbbi_philox_normal", seed, stream, it, chain0, D, N, N, float(scale), None,
             lib.F64 if dtype == np.float64 else lib.F32, 0, out.data_ptr(), stream_ptr(0))
    return to_numpy(out).astype(np.float64)


def run_as_bench(lib, pot, D, N, S, h, L, flags, seed, q_scale, q_shift, dtype=np.float64, chain0=0):
    """The calls of bench.py's main / main_c3 / main_c5: device-drawn q0, one pbbi_hmc_run over S
    iterations with sample, momentum and reject slabs."""
    import torch
    from physicsbasedbayesianinference_amd._device import empty, stream_ptr
    st = stream_ptr(0)
    code = lib.F64 if dtype == np.float64 else lib.F32
    q = empty((D, N), dtype, 0)
    lib.call("pbbi_philox_normal", seed, lib.STREAM_POSITION, 0, chain0, D, N, N, float(q_scale), None,
             code, 0, q.data_ptr(), st)
    if q_shift:
        q += q_shift
    samples, momenta = empty((S, D, N), dtype, 0), empty((S, D, N), dtype, 0)
    reject = empty((S, N), np.uint8, 0)
    lib.call("pbbi_hmc_run", pot.handle, lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
             momenta.data_ptr(), reject.data_ptr(), None, N, N, float(h), int(L), int(S), int(flags),
             seed, 0, chain0, 1.0, st)
    torch.cuda.synchronize()
    return samples, momenta, reject, q


def replay_groups(lib, op, starts, samples, momenta, reject, q_final, D, S, h, L, seed, q_scale,
                  q_shift, compat, dtype, check):
    """Oracle replay of the selected chains; `check(kind, gpu, ref, extra)` asserts per array."""
    n_rej = n_tot = 0
    for g in starts:
        sl = slice(g, g + GROUP)
        q = device_normal_dtype(lib, seed, lib.STREAM_POSITION, 0, g, D, GROUP, q_scale, dtype)
        if q_shift:
            q = (q.astype(dtype) + dtype(q_shift)).astype(np.float64)
        q = np.ascontiguousarray(q)
        for i in range(S):
            p = np.ascontiguousarray(
                device_normal_dtype(lib, seed, lib.STREAM_MOMENTUM, i, g, D, GROUP, 1.0, dtype))
            u = device_uniform(lib, seed, i, g, GROUP)
            ratio, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, None, h, L,
                                      compat=orc.COMPAT_P_FROM_OLDQ if compat else 0)
            gq = samples[i, :, sl].cpu().numpy().astype(np.float64)
            gp = momenta[i, :, sl].cpu().numpy().astype(np.float64)
            grej = reject[i, sl].cpu().numpy().astype(bool)
            check(g, i, gq, gp, grej, q, p, rej, ratio, u)
            n_rej += int(rej.sum())
            n_tot += GROUP
            if dtype != np.float64:  # continue from the kernel's own state (fp32 drift is not the test)
                q = np.ascontiguousarray(gq)
        assert np.array_equal(q_final[:, sl].cpu().numpy().astype(np.float64),
                              samples[S - 1, :, sl].cpu().numpy().astype(np.float64))
    return n_rej, n_tot


def c2_precision(D):
    A = np.random.RandomState(0).standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    return 0.5 * (Pm + Pm.T)


@pytest.mark.parametrize("S,ragged", [(3, 0), (3, 5), (20, 0), (20, 5), (100, 0), (100, 5)])
def test_c2_full_size_hmc_run_vs_oracle(P, lib, S, ragged):
    """bench.py's default workload: D=128, 65 536 chains, L=10, h=0.1, seed 42, compat flag -- for the
    iteration counts bench.py times: S = 20 is the driver's `--steps 20` (ONE fused launch of k_dense_hmc: its
    first iteration forms g(q_0), the other 19 read the carried gradient), S = 100 the default (two launches of
    50).  Every iteration of the run is replayed for 16 groups of 16 chains, so the slab index and the
    carried-gradient selector of a long fused launch are checked at the size and length they are timed at;
    chains must both accept and reject INSIDE the fused launches (the selector flips and stays).
    Reference: src/HMC.py:154-179."""
    D, N, L, h, seed = 128, 65536 - ragged, 10, 0.1, 42
    Pm = c2_precision(D)
    pot, op = P.GaussianDense(None, precision=Pm, const=0.0), orc.pot_gauss_dense(np.zeros(D), Pm)
    samples, momenta, reject, qf = run_as_bench(lib, pot, D, N, S, h, L, lib.COMPAT_P_FROM_OLDQ, seed,
                                                1.0, 0.0)
    worst, rej_later = [0.0], [0]

    def check(g, i, gq, gp, grej, q, p, rej, ratio, u):
        assert np.array_equal(grej, rej), f"group {g} iteration {i}"
        worst[0] = max(worst[0], scaled_err(gq, q), scaled_err(gp, p))
        if i >= 1:
            rej_later[0] += int(rej.sum())
    n_rej, n_tot = replay_groups(lib, op, chain_groups(N, 16, 128, 1), samples, momenta, reject, qf, D, S,
                                 h, L, seed, 1.0, 0.0, True, np.float64, check)
    assert worst[0] <= 1e-11, worst[0]
    assert n_tot == 16 * GROUP * S
    if S >= 20:  # replayed chains rejected (selector stays) and accepted (selector flips) after iteration 0
        assert 0 < rej_later[0] < 16 * GROUP * (S - 1) // 4, rej_later[0]
    # the whole ensemble's accept rate is what bench.py reports as config.accept_rate
    assert 0.5 < 1.0 - float(reject.float().mean()) <= 1.0


@pytest.mark.parametrize("kdk,ragged,S", [(True, 0, 3), (False, 0, 3), (True, 5, 3), (False, 5, 3),
                                          (True, 0, 20), (False, 0, 20), (True, 5, 20), (False, 5, 20)])
def test_c3_full_size_hmc_run_vs_oracle(P, lib, kdk, ragged, S):
    """bench.py --workload c3 [--exact-order]: Rosenbrock D=32, 262 144 chains, h=0.01, L=10,
    q0 = 1 + 0.1 z, seed 7.  S = 20 is the driver's `--steps 20`: two k_ros2_hmc launches of 10 fused
    iterations (the chain and U of its position stay in registers between them)."""
    D, N, L, h, seed = 32, 262144 - ragged, 10, 0.01, 7
    pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    flags = lib.COMPAT_P_FROM_OLDQ | (lib.KDK_FMA if kdk else 0)
    samples, momenta, reject, qf = run_as_bench(lib, pot, D, N, S, h, L, flags, seed, 0.1, 1.0)
    worst = [0.0]

    def check(g, i, gq, gp, grej, q, p, rej, ratio, u):
        assert np.array_equal(grej, rej), f"group {g} iteration {i}"
        if kdk:
            worst[0] = max(worst[0], scaled_err(gq, q), scaled_err(gp, p))
        else:
            assert np.array_equal(gq, q) and np.array_equal(gp, p), f"group {g} iteration {i}"
    replay_groups(lib, op, chain_groups(N, 16, 32, 2), samples, momenta, reject, qf, D, S, h, L, seed,
                  0.1, 1.0, True, np.float64, check)
    assert worst[0] <= 1e-12, worst[0]


@pytest.mark.parametrize("ragged", [0, 5])
def test_c5_full_size_hmc_run_vs_oracle(P, lib, ragged):
    """bench.py --workload c5: D=4096 dense precision, 8 192 chains, fp32, h=0.05, L=10, seed 7.
    fp32 kernel against the fp64 oracle restarted from the kernel's own previous state."""
    D, N, L, h, S, seed = 4096, 8192 - ragged, 10, 0.05, 3, 7  # iterations 1, 2 start from carried gradients
    Pm = c2_precision(D)
    pot = P.GaussianDense(None, precision=Pm, const=0.0, dtype="float32")
    op = orc.pot_gauss_dense(np.zeros(D), Pm)
    samples, momenta, reject, qf = run_as_bench(lib, pot, D, N, S, h, L, lib.COMPAT_P_FROM_OLDQ, seed,
                                                1.0, 0.0, dtype=np.float32)
    worst, clear_n = [0.0], [0]

    def check(g, i, gq, gp, grej, q, p, rej, ratio, u):
        with np.errstate(divide="ignore"):
            clear = np.abs(np.log(u) - np.minimum(0.0, np.log(ratio))) > 1e-2
        assert np.array_equal(grej[clear], rej[clear]), f"group {g} iteration {i}"
        same = grej == rej  # compare states where the decision agrees (all of `clear`)
        worst[0] = max(worst[0], scaled_err(gq[:, same], q[:, same]), scaled_err(gp[:, same], p[:, same]))
        clear_n[0] += int(clear.sum())
    starts = [0, N - GROUP, 120, 4090]  # first column tile, last (ragged) one, two tile boundaries
    replay_groups(lib, op, starts, samples, momenta, reject, qf, D, S, h, L, seed, 1.0, 0.0, True,
                  np.float32, check)
    assert worst[0] <= 2e-4, worst[0]
    assert clear_n[0] > len(starts) * GROUP * S // 2


# ---------------------------------------------------------------------------------------------
# Reject-branch stress of k_dense_hmc (kernels_dense.hip, the branch after `if (reject)`): the old
# position is re-loaded (src/HMC.py:175) and the stored momentum is one of
#   compat:              the OLD POSITION          (src/HMC.py:176)
#   non-compat, Philox:  the draw parked in the momentum slab before the trajectory
#   non-compat, upload:  p_in re-loaded
# h = 0.5 (uploaded draws, where u[::3] = 1.5 forces rejections too) / 0.7 (in-kernel draws), L = 4:
# 40-70 % of the chains reject.
# ---------------------------------------------------------------------------------------------
def _stress_problem(D, zero_mean):
    rs = np.random.RandomState(D)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    Pm = 0.5 * (Pm + Pm.T)
    mu = np.zeros(D) if zero_mean else rs.standard_normal(D)
    return Pm, mu


@pytest.mark.parametrize("method", ["Leapfrog", "Stormer-Verlet"])
@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("D,zero_mean,mass", [(128, True, False), (128, False, True), (100, True, True),
                                              (100, False, False), (64, False, False), (24, True, True),
                                              (96, True, False), (72, False, True),
                                              # 128 < D <= 256: the same kernel with P streamed (kernels_dstream.hip)
                                              (256, True, False), (256, False, True), (200, False, False),
                                              (192, True, True), (130, False, True)])
def test_dense_reject_branch_uploaded_draws(P, lib, D, zero_mean, mass, compat, method):
    from test_gpu_parity import gpu_hmc_iter
    N, h, L = 333, 0.5, 4  # ragged: 333 = 2*128 + 77 (streamed kernel: 5*64 + 13, three waves past the end)
    Pm, mu = _stress_problem(D, zero_mean)
    pot = P.GaussianDense(None if zero_mean else mu, precision=Pm, const=0.25)
    op = orc.pot_gauss_dense(mu, Pm, 0.25)
    rs = np.random.RandomState(7 * D + N)
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    q = rs.standard_normal((D, N)) + mu[:, None]
    n_rej = 0
    for it in range(2):
        p = rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0)
        u = rs.uniform(size=N)
        u[::3] = 1.5  # forced rejections (u > min(1, ratio) always)
        qo, po, ratio, rej = gpu_hmc_iter(lib, pot, method, q, p, u, m, h, L, compat=compat)
        q_or, p_or = q.copy(), p.copy()
        r_or, rej_or = orc.hmc_iter(op, method, q_or, p_or, u, m, h, L,
                                    compat=orc.COMPAT_P_FROM_OLDQ if compat else 0)
        assert np.array_equal(rej, rej_or) and rej[::3].all()
        assert scaled_err(qo, q_or) <= 1e-11 and scaled_err(po, p_or) <= 1e-11
        # rejected chains hold exact copies, not recomputed values
        assert np.array_equal(qo[:, rej], q[:, rej])
        assert np.array_equal(po[:, rej], q[:, rej] if compat else p[:, rej])
        n_rej += int(rej.sum())
        q = qo
    assert n_rej >= 0.3 * 2 * N


@pytest.mark.parametrize("method", ["Leapfrog", "Stormer-Verlet"])
@pytest.mark.parametrize("compat", [True, False])
@pytest.mark.parametrize("D,zero_mean,mass", [(128, True, False), (128, False, True), (100, True, True),
                                              (100, False, False), (96, False, False), (80, True, True),
                                              (64, True, False),
                                              (256, True, False), (256, False, True), (200, True, True), (160, False, False)])
def test_dense_reject_branch_in_kernel_draws(P, lib, D, zero_mean, mass, compat, method):
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    N, h, L, S, seed, chain0, iter0 = 333, 0.7, 4, 3, 11, 77, 5  # 40-67 % rejects in every case
    if D > 128:
        h = 0.45  # (streamed kernel, D up to 256: the same reject rates at a shorter step)
    Pm, mu = _stress_problem(D, zero_mean)
    pot = P.GaussianDense(None if zero_mean else mu, precision=Pm, const=0.25)
    op = orc.pot_gauss_dense(mu, Pm, 0.25)
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    st = stream_ptr(0)
    q0 = device_normal(lib, seed, lib.STREAM_POSITION, iter0, chain0, D, N, 1.0) + mu[:, None]
    qd = as_device(q0, 0, np.float64)
    samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
    reject, ratio = empty((S, N), np.uint8, 0), empty((S, N), np.float64, 0)
    lib.call("pbbi_hmc_run", pot.handle, orc.METHODS[method], qd.data_ptr(),
             md.data_ptr() if mass else None, samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(),
             ratio.data_ptr(), N, N, h, L, S, lib.COMPAT_P_FROM_OLDQ if compat else 0, seed, iter0, chain0,
             1.0, st)
    torch.cuda.synchronize()
    samples, momenta, reject = to_numpy(samples), to_numpy(momenta), to_numpy(reject).astype(bool)
    q = np.ascontiguousarray(q0)
    pstd = np.sqrt(m) if mass else np.ones(N)
    n_rej = 0
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, iter0 + i, chain0, D, N, 1.0, pstd)
        u = device_uniform(lib, seed, iter0 + i, chain0, N)
        p_draw = p.copy()
        _, rej = orc.hmc_iter(op, method, q, p, u, m, h, L, compat=orc.COMPAT_P_FROM_OLDQ if compat else 0)
        assert np.array_equal(reject[i], rej), f"iteration {i}"
        assert scaled_err(samples[i], q) <= 1e-11 and scaled_err(momenta[i], p) <= 1e-11
        # rejected chains hold exact copies: of the kernel's own previous position, or of the draw
        gpu_old = q0 if i == 0 else samples[i - 1]
        assert np.array_equal(samples[i][:, rej], gpu_old[:, rej])
        assert np.array_equal(momenta[i][:, rej], gpu_old[:, rej] if compat else p_draw[:, rej])
        n_rej += int(rej.sum())
    assert n_rej >= 0.3 * S * N
    assert np.array_equal(to_numpy(qd), samples[S - 1])


@pytest.mark.parametrize("D,zero_mean,mass,compat", [(128, True, False, True), (128, False, True, False),
                                                     (100, False, True, False), (65, True, True, True),
                                                     (100, True, False, True), (64, True, False, True),
                                                     (64, False, True, False), (48, True, True, True),
                                                     (33, False, False, False), (96, True, False, True),
                                                     (80, False, True, False), (70, True, True, True),
                                                     (32, True, False, True), (24, False, True, False),
                                                     (7, True, True, True),
                                                     # streamed P (kernels_dstream.hip): DPS = 256 and 192
                                                     (256, True, False, True), (256, False, True, False),
                                                     (200, False, True, False), (192, True, False, True),
                                                     (129, False, False, True)])
@pytest.mark.parametrize("method", [0, 1])   # Leapfrog, Stormer-Verlet (its extra mat-vec at q_{L+1} is g(q_new) too)
def test_dense_run_carries_the_gradient_bit_identically(P, lib, D, zero_mean, mass, compat, method):
    """pbbi_hmc_run on the dense kernel at D <= 128 (round 3: padded D and every tile size DP = 32 / 64 / 96 / 128, fused launches
    included -- rows d >= D are handled by bounded buffer descriptors, not guards) keeps the gradient of the chain's position between
    iterations (kernels_dense.hip CARRY: accepted chains take g(q_new) of the previous launch, rejected
    ones the gradient they started from) instead of forming it again.  One run of S iterations must equal
    S runs of one iteration (which cannot carry anything) bit for bit -- samples, momenta, ratios and
    decisions -- with a quarter to half of the chains rejecting; a burn-in run (two scratch slabs) ends in the same state."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    N, h, L, S, seed, chain0, iter0 = 1000, 0.7, 4, 7, 3, 19, 2
    if D > 128:
        h = 0.4 if method == 0 else 0.06   # (Stormer-Verlet's returned velocity is a half step behind: its energy error is first order)
    Pm, mu = _stress_problem(D, zero_mean)
    pot = P.GaussianDense(None if zero_mean else mu, precision=Pm, const=0.25)
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    st = stream_ptr(0)
    flags = lib.COMPAT_P_FROM_OLDQ if compat else 0
    q0 = device_normal(lib, seed, lib.STREAM_POSITION, iter0, chain0, D, N, 1.0) + mu[:, None]

    def run(s_per_call):
        qd = as_device(q0, 0, np.float64)
        samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
        reject, ratio = empty((S, N), np.uint8, 0), empty((S, N), np.float64, 0)
        for i in range(0, S, s_per_call):
            lib.call("pbbi_hmc_run", pot.handle, method, qd.data_ptr(), md.data_ptr() if mass else None,
                     samples[i].data_ptr(), momenta[i].data_ptr(), reject[i].data_ptr(), ratio[i].data_ptr(),
                     N, N, h, L, min(s_per_call, S - i), flags, seed, iter0 + i, chain0, 1.0, st)
        torch.cuda.synchronize()
        return to_numpy(samples), to_numpy(momenta), to_numpy(reject), to_numpy(ratio), to_numpy(qd)

    one = run(S)        # iterations 1 .. S-1 read the carried gradient
    each = run(1)       # every iteration forms its own
    for a, b in zip(one, each):
        assert np.array_equal(a, b)
    assert 0.02 < one[2].mean() < 0.8   # (rejections happen: the selector both stays and flips)
    qd = as_device(q0, 0, np.float64)   # burn-in form: nothing recorded, same final state
    lib.call("pbbi_hmc_run", pot.handle, method, qd.data_ptr(), md.data_ptr() if mass else None, None, None, None,
             None, N, N, h, L, S, flags, seed, iter0, chain0, 1.0, st)
    torch.cuda.synchronize()
    assert np.array_equal(to_numpy(qd), one[4])


@pytest.mark.parametrize("D,N", [(256, 1), (200, 17), (256, 64), (192, 65), (140, 200)])
def test_dense_stream_strided_state_and_small_ensembles(P, lib, D, N):
    """kernels_dstream.hip, the corners of its launch shape: a caller's state with a leading stride > N (the run's
    first iteration then reads another stride than it writes: a fused launch of one), ensembles of 1 chain, of less
    than a workgroup, of exactly one and of one plus a chain (three waves past the end keep serving the ring).
    The strided run equals the contiguous one bit for bit, both equal the oracle's replay, padding untouched."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    h, L, S, seed, chain0, ld = 0.3, 5, 4, 8, 3, N + 9
    Pm, mu = _stress_problem(D, False)
    pot, op = P.GaussianDense(mu, precision=Pm, const=0.1), orc.pot_gauss_dense(mu, Pm, 0.1)
    st = stream_ptr(0)
    q0 = device_normal(lib, seed, lib.STREAM_POSITION, 0, chain0, D, N, 1.0) + mu[:, None]
    out = []
    for stride in (N, ld):
        qd = torch.full((D, stride), 7.25, dtype=torch.float64, device="cuda")
        qd[:, :N] = as_device(q0, 0, np.float64)
        samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
        reject = empty((S, N), np.uint8, 0)
        lib.call("pbbi_hmc_run", pot.handle, 0, qd.data_ptr(), None, samples.data_ptr(), momenta.data_ptr(),
                 reject.data_ptr(), None, N, stride, h, L, S, lib.COMPAT_P_FROM_OLDQ, seed, 0, chain0, 1.0, st)
        torch.cuda.synchronize()
        assert bool((qd[:, N:] == 7.25).all())
        out.append((to_numpy(samples), to_numpy(momenta), to_numpy(reject), to_numpy(qd[:, :N].contiguous())))
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    samples, momenta, reject, qf = out[0]
    q = np.ascontiguousarray(q0)
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, chain0, D, N, 1.0)
        u = device_uniform(lib, seed, i, chain0, N)
        _, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, None, h, L, compat=orc.COMPAT_P_FROM_OLDQ)
        assert np.array_equal(reject[i].astype(bool), rej)
        assert scaled_err(samples[i], q) <= 1e-11 and scaled_err(momenta[i], p) <= 1e-11
    assert np.array_equal(qf, samples[S - 1])


@pytest.mark.parametrize("D,N,dtype,zero_mean,mass", [(512, 768, "float32", True, False),
                                                      (200, 150, "float64", False, True),
                                                      (384, 300, "float32", False, True)])
def test_gemm_path_run_carries_the_gradient_bit_identically(P, lib, D, N, dtype, zero_mean, mass):
    """The same on the GEMM path (kernels_big.hip, D > 128 / fp32): inside pbbi_hmc_run the first of the
    L + 1 GEMMs of an iteration is replaced by one elementwise pass over the gradient the previous iteration
    kept (its last GEMM's for accepted chains, the older one for rejected chains), and H_old by the kept x.g
    partial sums.  One run of S iterations == S runs of one, bit for bit, rejections included."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    h, L, S, seed, chain0, iter0 = 0.35, 3, 5, 4, 9, 1
    npdt = np.float32 if dtype == "float32" else np.float64
    rs = np.random.RandomState(D)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    Pm = 0.5 * (Pm + Pm.T)
    mu = None if zero_mean else rs.standard_normal(D) * 0.5
    pot = P.GaussianDense(mu, precision=Pm, const=0.1, dtype=dtype)
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, npdt) if mass else None
    st = stream_ptr(0)
    q0 = rs.standard_normal((D, N))

    def run(s_per_call):
        qd = as_device(q0, 0, npdt)
        samples, momenta = empty((S, D, N), npdt, 0), empty((S, D, N), npdt, 0)
        reject, ratio = empty((S, N), np.uint8, 0), empty((S, N), npdt, 0)
        for i in range(0, S, s_per_call):
            lib.call("pbbi_hmc_run", pot.handle, 0, qd.data_ptr(), md.data_ptr() if mass else None,
                     samples[i].data_ptr(), momenta[i].data_ptr(), reject[i].data_ptr(), ratio[i].data_ptr(),
                     N, N, h, L, min(s_per_call, S - i), lib.COMPAT_P_FROM_OLDQ, seed, iter0 + i, chain0, 1.0, st)
        torch.cuda.synchronize()
        return to_numpy(samples), to_numpy(momenta), to_numpy(reject), to_numpy(ratio), to_numpy(qd)

    one, each = run(S), run(1)
    for a, b in zip(one, each):
        assert np.array_equal(a, b)
    assert 0.05 < one[2].mean() < 0.95


@pytest.mark.parametrize("kind", ["dense128", "ros32", "diag8"])
def test_run_dyn_without_length_flags_reports_L_for_every_iteration(P, lib, kind):
    """pbbi_hmc_run_dyn with neither PBBI_PER_CHAIN_STEPS nor PBBI_UTURN_STOP is pbbi_hmc_run plus a steps
    array that reads L everywhere -- also for the iterations a fused launch covers (dense Gaussian D = 128,
    two-lane Rosenbrock), which used to get their first row only."""
    import torch
    from physicsbasedbayesianinference_amd._device import empty, stream_ptr, to_numpy
    N, S, L, h = 300, 7, 4, 0.05
    if kind == "dense128":
        D = 128
        Pm, _ = _stress_problem(D, True)
        pot = P.GaussianDense(None, precision=Pm, const=0.0)
    elif kind == "ros32":
        D, pot = 32, P.Rosenbrock(32)
    else:
        D, pot = 8, P.GaussianDiag(np.zeros(8), prec=np.ones(8), const=0.0)
    st = stream_ptr(0)
    qd = torch.full((D, N), 1.0, dtype=torch.float64, device="cuda") + 0.1 * torch.randn((D, N), dtype=torch.float64,
                                                                                        device="cuda")
    samples = empty((S, D, N), np.float64, 0)
    steps = torch.full((S, N), -7, dtype=torch.int32, device="cuda")
    lib.call("pbbi_hmc_run_dyn", pot.handle, 0, qd.data_ptr(), None, samples.data_ptr(), None, None, None,
             steps.data_ptr(), N, N, h, L, S, lib.COMPAT_P_FROM_OLDQ, 3, 0, 0, 1.0, st)
    torch.cuda.synchronize()
    assert np.array_equal(to_numpy(steps), np.full((S, N), L, dtype=np.int32))


@pytest.mark.parametrize("D,mass,method,harmonic", [(64, False, 0, False), (100, True, 0, False), (48, True, 1, True),
                                                     (256, False, 0, False)])
def test_separable_run_fused_bit_identically(P, lib, D, mass, method, harmonic):
    """k_sep_hmc (harmonic / diagonal Gaussian, 16 < D <= 256, PBBI_KDK_FMA) keeps a workgroup's chains in
    registers for up to 16 iterations of a run.  One run of S iterations == S runs of one, bit for bit
    (positions are re-formed from the stored value minus the mean, as a launch of its own does), with
    rejections, ragged N, masses, both integrators; the burn-in form ends in the same state."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    N, h, L, S, seed, chain0, iter0 = 1003, 0.9, 3, 19, 6, 40, 3
    rs = np.random.RandomState(D)
    prec = rs.uniform(0.5, 2.0, D)
    pot = P.Harmonic(prec) if harmonic else P.GaussianDiag(rs.standard_normal(D), prec=prec, const=0.0)
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    st = stream_ptr(0)
    flags = lib.COMPAT_P_FROM_OLDQ | lib.KDK_FMA
    q0 = rs.standard_normal((D, N))

    def run(s_per_call, record=True):
        qd = as_device(q0, 0, np.float64)
        samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
        reject, ratio = empty((S, N), np.uint8, 0), empty((S, N), np.float64, 0)
        for i in range(0, S, s_per_call):
            lib.call("pbbi_hmc_run", pot.handle, method, qd.data_ptr(), md.data_ptr() if mass else None,
                     samples[i].data_ptr() if record else None, momenta[i].data_ptr() if record else None,
                     reject[i].data_ptr() if record else None, ratio[i].data_ptr() if record else None,
                     N, N, h, L, min(s_per_call, S - i), flags, seed, iter0 + i, chain0, 1.0, st)
        torch.cuda.synchronize()
        return to_numpy(samples), to_numpy(momenta), to_numpy(reject), to_numpy(ratio), to_numpy(qd)

    one, each = run(S), run(1)
    for a, b in zip(one, each):
        assert np.array_equal(a, b)
    assert 0.02 < one[2].mean() < 0.98
    assert np.array_equal(run(S, record=False)[4], one[4])


@pytest.mark.parametrize("case", ["quartic9", "quartic9_ad", "quartic9_f32", "quartic48_workspace", "logistic"])
def test_custom_potential_run_fused_bit_identically(P, lib, case):
    """User potentials (plugin kernels): pbbi_hmc_run hands the plugin 16 iterations at a time; the register
    kernels keep the chain and its potential energy on chip (one evaluation of the user's potential per
    iteration instead of two), the workspace kernels unroll the call.  One run of S iterations == S runs of
    one, bit for bit, with rejections, masses, ragged N; burn-in form ends in the same state."""
    import torch
    from physicsbasedbayesianinference_amd import CustomPotential
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    from custom_sources import LOGISTIC, QUARTIC, QUARTIC_AD
    rs = np.random.RandomState(3)
    npdt, h = np.float64, 0.35
    if case == "quartic9":
        D, pot = 9, CustomPotential(9, QUARTIC, [1.0, 0.5])
    elif case == "quartic9_ad":
        D, pot = 9, CustomPotential(9, QUARTIC_AD, [1.0, 0.5])
    elif case == "quartic9_f32":
        D, pot, npdt = 9, CustomPotential(9, QUARTIC, [1.0, 0.5], dtype="float32"), np.float32
    elif case == "quartic48_workspace":
        D, pot, h = 48, CustomPotential(48, QUARTIC, [1.0, 0.5]), 0.2
    else:
        M, D = 64, 6
        X = rs.standard_normal((M, D))
        y = (rs.uniform(size=M) < 0.5).astype(np.float64)
        pot, h = CustomPotential(D, LOGISTIC, np.concatenate([[float(M)], X.ravel(), y, [1.0]])), 0.25
    N, L, S, seed, chain0, iter0 = 777, 4, 21, 8, 5, 2
    m = 1.0 + (np.arange(N) % 3) * 0.5
    md = as_device(m, 0, npdt)
    st = stream_ptr(0)
    q0 = rs.standard_normal((D, N))

    def run(s_per_call, record=True):
        qd = as_device(q0, 0, npdt)
        samples, momenta = empty((S, D, N), npdt, 0), empty((S, D, N), npdt, 0)
        reject, ratio = empty((S, N), np.uint8, 0), empty((S, N), npdt, 0)
        for i in range(0, S, s_per_call):
            lib.call("pbbi_hmc_run", pot.handle, 0, qd.data_ptr(), md.data_ptr(),
                     samples[i].data_ptr() if record else None, momenta[i].data_ptr() if record else None,
                     reject[i].data_ptr() if record else None, ratio[i].data_ptr() if record else None,
                     N, N, h, L, min(s_per_call, S - i), lib.COMPAT_P_FROM_OLDQ, seed, iter0 + i, chain0, 1.0, st)
        torch.cuda.synchronize()
        return to_numpy(samples), to_numpy(momenta), to_numpy(reject), to_numpy(ratio), to_numpy(qd)

    one, each = run(S), run(1)
    for a, b in zip(one, each):
        assert np.array_equal(a, b)
    assert 0.02 < one[2].mean() < 0.98
    assert np.array_equal(run(S, record=False)[4], one[4])


@pytest.mark.parametrize("D,mass,method", [(64, False, 0), (100, True, 0), (128, False, 0), (24, True, 1)])
def test_rosenbrock_multilane_run_fused_bit_identically(P, lib, D, mass, method):
    """k_rosg_hmc (Rosenbrock, 4 / 8 lanes per chain, PBBI_KDK_FMA; Stormer-Verlet from D = 17) takes up to 16
    iterations of a run per launch and carries U of the chain's position between them.  One run of S
    iterations == S runs of one, bit for bit, with rejections, ragged N, masses; burn-in ends alike."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    N, h, L, S, seed, chain0, iter0 = 501, 0.05, 6, 19, 9, 11, 4
    rs = np.random.RandomState(D)
    pot = P.Rosenbrock(D)
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    st = stream_ptr(0)
    flags = lib.COMPAT_P_FROM_OLDQ | lib.KDK_FMA
    q0 = 1.0 + 0.3 * rs.standard_normal((D, N))

    def run(s_per_call, record=True):
        qd = as_device(q0, 0, np.float64)
        samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
        reject, ratio = empty((S, N), np.uint8, 0), empty((S, N), np.float64, 0)
        for i in range(0, S, s_per_call):
            lib.call("pbbi_hmc_run", pot.handle, method, qd.data_ptr(), md.data_ptr() if mass else None,
                     samples[i].data_ptr() if record else None, momenta[i].data_ptr() if record else None,
                     reject[i].data_ptr() if record else None, ratio[i].data_ptr() if record else None,
                     N, N, h, L, min(s_per_call, S - i), flags, seed, iter0 + i, chain0, 1.0, st)
        torch.cuda.synchronize()
        return to_numpy(samples), to_numpy(momenta), to_numpy(reject), to_numpy(ratio), to_numpy(qd)

    one, each = run(S), run(1)
    for a, b in zip(one, each):
        assert np.array_equal(a, b)
    assert 0.02 < one[2].mean() < 0.98
    assert np.array_equal(run(S, record=False)[4], one[4])


@pytest.mark.parametrize("kind,D,mass,method,h", [("diag", 8, False, 0, 0.9), ("harm", 3, True, 1, 0.7),
                                                  ("ros", 12, True, 0, 0.06), ("diag", 16, True, 0, 0.9),
                                                  ("std", 1, False, 0, 1.3)])
def test_chain_per_lane_run_fused_bit_identically(P, lib, kind, D, mass, method, h):
    """k_lane_hmc (D <= 16: the launch is most of an iteration) keeps a lane's chain for up to 16 iterations of
    a run and carries U of its position between them.  One run of S iterations == S runs of one, bit for bit."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    N, L, S, seed, chain0, iter0 = 1003, 3, 21, 2, 7, 1
    rs = np.random.RandomState(D)
    if kind == "diag":
        pot = P.GaussianDiag(rs.standard_normal(D), prec=rs.uniform(0.5, 2.0, D), const=0.0)
    elif kind == "harm":
        pot = P.Harmonic(rs.uniform(0.5, 2.0, D))
    elif kind == "ros":
        pot = P.Rosenbrock(D)
    else:
        pot = P.StandardGaussian(1)
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    st = stream_ptr(0)
    q0 = (1.0 if kind == "ros" else 0.0) + (0.3 if kind == "ros" else 1.0) * rs.standard_normal((D, N))

    def run(s_per_call, record=True):
        qd = as_device(q0, 0, np.float64)
        samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
        reject, ratio = empty((S, N), np.uint8, 0), empty((S, N), np.float64, 0)
        for i in range(0, S, s_per_call):
            lib.call("pbbi_hmc_run", pot.handle, method, qd.data_ptr(), md.data_ptr() if mass else None,
                     samples[i].data_ptr() if record else None, momenta[i].data_ptr() if record else None,
                     reject[i].data_ptr() if record else None, ratio[i].data_ptr() if record else None,
                     N, N, h, L, min(s_per_call, S - i), lib.COMPAT_P_FROM_OLDQ, seed, iter0 + i, chain0, 1.0, st)
        torch.cuda.synchronize()
        return to_numpy(samples), to_numpy(momenta), to_numpy(reject), to_numpy(ratio), to_numpy(qd)

    one, each = run(S), run(1)
    for a, b in zip(one, each):
        assert np.array_equal(a, b)
    assert 0.02 < one[2].mean() < 0.98
    assert np.array_equal(run(S, record=False)[4], one[4])


def test_dense_carried_run_beyond_two_to_the_31_bytes(P, lib):
    """The carried-gradient slabs of the dense kernel are addressed with unsigned 32-bit byte offsets from one
    descriptor.  Round 2 stopped carrying at D*N*16 >= 2^31 (N = 2^20 at D = 128); the offsets are good to 2^32
    (pbbi_describe_run: N <= 2 096 639).  N = 2^20 + 37 puts the last chains' slab-1 offsets past 2^31: a fused,
    carried run of 4 iterations there against the oracle, first / last / boundary chains."""
    D, N, L, h, S, seed = 128, (1 << 20) + 37, 10, 0.1, 4, 42
    Pm = c2_precision(D)
    pot, op = P.GaussianDense(None, precision=Pm, const=0.0), orc.pot_gauss_dense(np.zeros(D), Pm)
    import ctypes
    buf = ctypes.create_string_buffer(512)
    lib.call("pbbi_describe_run", pot.handle, 0, N, N, L, S, 1, buf, len(buf))
    assert "carried between iterations: yes" in buf.value.decode()
    samples, momenta, reject, qf = run_as_bench(lib, pot, D, N, S, h, L, lib.COMPAT_P_FROM_OLDQ, seed, 1.0, 0.0)
    worst = [0.0]

    def check(g, i, gq, gp, grej, q, p, rej, ratio, u):
        assert np.array_equal(grej, rej), f"group {g} iteration {i}"
        worst[0] = max(worst[0], scaled_err(gq, q), scaled_err(gp, p))
    starts = [0, N - GROUP, (1 << 20) - 8, (1 << 19) + 3, N - 128 - 5]
    replay_groups(lib, op, starts, samples, momenta, reject, qf, D, S, h, L, seed, 1.0, 0.0, True, np.float64, check)
    assert worst[0] <= 1e-11, worst[0]


@pytest.mark.parametrize("N,carried", [((1 << 20) - 300, True), ((1 << 20) + 4001, False)])
def test_dense_stream_beyond_two_to_the_31_bytes(P, lib, N, carried):
    """The streamed dense kernel (D = 256) where its 32-bit byte offsets are large: one (D, N) slab is 2.1 GB, so
    the last rows of the state arrays and slab 1 of the carried gradient sit past 2^31 (unsigned offsets: good to
    2^32).  N just below the carried path's limit (DPS * N * 16 < 2^32 - 2^20: a fused, carried run) and just above
    it (the same kernel family without the carry, one launch per iteration), 3 iterations against the oracle for
    the first / last / boundary chains."""
    import ctypes
    D, L, h, S, seed = 256, 10, 0.1, 3, 42
    Pm = c2_precision(D)
    pot, op = P.GaussianDense(None, precision=Pm, const=0.0), orc.pot_gauss_dense(np.zeros(D), Pm)
    buf = ctypes.create_string_buffer(1024)
    lib.call("pbbi_describe_run", pot.handle, 0, N, N, L, S, 1, buf, len(buf))
    d = buf.value.decode()
    assert "streamed P" in d and ("carried between iterations: yes" in d) == carried, d
    samples, momenta, reject, qf = run_as_bench(lib, pot, D, N, S, h, L, lib.COMPAT_P_FROM_OLDQ, seed, 1.0, 0.0)
    worst = [0.0]

    def check(g, i, gq, gp, grej, q, p, rej, ratio, u):
        assert np.array_equal(grej, rej), f"group {g} iteration {i}"
        worst[0] = max(worst[0], scaled_err(gq, q), scaled_err(gp, p))
    starts = [0, N - GROUP, (1 << 19) + 3, N - 64 - 5, N // 64 * 64 - 8]
    replay_groups(lib, op, starts, samples, momenta, reject, qf, D, S, h, L, seed, 1.0, 0.0, True, np.float64, check)
    assert worst[0] <= 1e-11, worst[0]
