"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI
(include/pbbi.h via physicsbasedbayesianinference_amd._lib), against
  (1) the golden vectors recorded from the reference's own Python, and
  (2) the CPU oracle on the same seeded inputs.

Tolerances (fp64):
  chain-per-lane kernels (harmonic / diagonal Gaussian / Rosenbrock) keep the
    oracle's operation order exactly -> q, p BIT-EXACT vs the oracle, 1e-12 vs golden;
  dense-Gaussian MFMA kernel sums dot products in a different order (k-ordered
    MFMA chain + lane butterfly) -> RTOL_DENSE = 1e-11 scaled by max|ref|;
  reject masks: EQUAL in every case (the fixtures hold no |u - ratio| tie).
"""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.constants import k as kB

from conftest import load_golden
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

RTOL_GOLDEN = 1e-12
RTOL_DENSE = 1e-11


@pytest.fixture(scope="module")
def P():
    import physicsbasedbayesianinference_amd as pkg
    return pkg


@pytest.fixture(scope="module")
def lib():
    from physicsbasedbayesianinference_amd import _lib
    _lib.load()
    return _lib


def scaled_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern differs"
    inf = np.isinf(a) | np.isinf(b)
    assert np.array_equal(a[inf], b[inf]), "inf pattern differs"
    ok = ~(np.isnan(a) | inf)
    if not ok.any():
        return 0.0
    return float(np.max(np.abs(a[ok] - b[ok]))) / max(1.0, float(np.max(np.abs(b[ok]))))


def make_pot(P, g, kind):
    if kind == "dense":
        return P.GaussianDense(g["mean"], precision=g["precision"], const=float(g["const"]))
    if kind == "std":
        return P.StandardGaussian(int(g["D"]))
    if kind == "rosenbrock":
        return P.Rosenbrock(int(g["D"]), float(g["a"]), float(g["b"]), float(g["s"]))
    raise KeyError(kind)


def orc_pot(g, kind):
    if kind == "dense":
        return orc.pot_gauss_dense(g["mean"], g["precision"], float(g["const"]))
    if kind == "std":
        return orc.pot_gauss_diag(np.zeros(int(g["D"])), np.ones(int(g["D"])))
    return orc.pot_rosenbrock(int(g["D"]), float(g["a"]), float(g["b"]), float(g["s"]))


def gpu_hmc_iter(lib, pot, method, q, p, u, mass, h, L, compat=True):
    """One pbbi_hmc_iter call on host arrays; returns q_out, p_out, ratio, reject."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    D, N = q.shape
    dev = pot.device
    qd, pd, ud = (as_device(x, dev, np.float64) for x in (q, p, u))
    md = None if mass is None or np.all(np.asarray(mass) == 1.0) else as_device(mass, dev, np.float64)
    qo, po = empty((D, N), np.float64, dev), empty((D, N), np.float64, dev)
    ro, rj = empty((N,), np.float64, dev), empty((N,), np.uint8, dev)
    lib.call("pbbi_hmc_iter", pot.handle, orc.METHODS[method], qd.data_ptr(), pd.data_ptr(),
             ud.data_ptr(), md.data_ptr() if md is not None else None, qo.data_ptr(), po.data_ptr(),
             ro.data_ptr(), rj.data_ptr(), N, N, float(h), int(L),
             lib.COMPAT_P_FROM_OLDQ if compat else 0, stream_ptr(dev))
    torch.cuda.synchronize()
    return to_numpy(qo), to_numpy(po), to_numpy(ro), to_numpy(rj).astype(bool)


GETSAMPLES = [
    ("G3_getsamples_c1", "std", True),
    ("G4_getsamples_dense_d8", "dense", False),
    ("G4_getsamples_dense_d128", "dense", False),
    ("G4b_getsamples_dense_mean_d16", "dense", False),
    ("G5_getsamples_rosenbrock_d32", "rosenbrock", True),
    ("G6_getsamples_rejects", "std", True),
    ("G7_getsamples_nan", "std", True),
    ("G8_getsamples_mass", "dense", False),
    ("G11_getsamples_test2", "dense", False),
    ("G12_getsamples_stormerverlet", "dense", False),
]


@pytest.mark.parametrize("name,kind,exact", GETSAMPLES)
def test_hmc_iter_vs_golden_and_oracle(P, lib, name, kind, exact):
    g = load_golden(name)
    pot, opot = make_pot(P, g, kind), orc_pot(g, kind)
    S, L, h, method = int(g["S"]), int(g["numSteps"]), float(g["stepSize"]), str(g["method"])
    q = np.ascontiguousarray(g["q0"])
    for i in range(S):
        p, u = np.ascontiguousarray(g["p_draw"][i]), g["u"][i]
        qo, po, ratio, rej = gpu_hmc_iter(lib, pot, method, q, p, u, g["mass"], h, L)
        # --- vs the reference's own run
        assert np.array_equal(rej, g["reject_mask"][i]), f"{name} it {i}: reject mask differs"
        tol = RTOL_GOLDEN if exact else RTOL_DENSE
        assert scaled_err(qo, g["samples"][:, :, i]) <= tol
        assert scaled_err(po, g["momenta"][:, :, i]) <= tol
        # --- vs the oracle on the same inputs
        q_or, p_or = q.copy(), p.copy()
        r_or, rej_or = orc.hmc_iter(opot, method, q_or, p_or, u, g["mass"], h, L)
        assert np.array_equal(rej, rej_or)
        if exact:
            assert np.array_equal(qo, q_or, equal_nan=True) and np.array_equal(po, p_or, equal_nan=True)
        else:
            assert scaled_err(qo, q_or) <= RTOL_DENSE and scaled_err(po, p_or) <= RTOL_DENSE
        fin = np.isfinite(r_or) & (r_or > 0) & np.isfinite(ratio) & (ratio > 0)
        assert np.array_equal(np.isnan(ratio), np.isnan(r_or))
        if fin.any():
            assert np.max(np.abs(np.log(ratio[fin]) - np.log(r_or[fin]))) < 1e-8
        q = qo


@pytest.mark.parametrize("name,kind,exact", GETSAMPLES)
def test_getsamples_class_api_vs_golden(P, name, kind, exact):
    """The drop-in class API on the reference's seed: HMC(...).getSamples(...)."""
    g = load_golden(name)
    D, N, S = int(g["D"]), int(g["N"]), int(g["S"])
    np.random.seed(int(g["seed"]))
    ens = P.Ensemble(D, N)
    ens.mass = g["mass"].copy()
    pot = make_pot(P, g, kind)
    hmc = P.HMC(ens, float(g["simulTime"]), float(g["stepSize"]), None, potential=pot,
                method=str(g["method"]), verbose=False)
    assert hmc.integrator.numSteps == int(g["numSteps"])
    samples, momenta = hmc.getSamples(S, float(g["temperature"]), float(g["qStd"]))
    assert samples.shape == (D, N, S) and momenta.shape == (D, N, S)
    assert samples.dtype == np.float64
    assert np.array_equal(hmc.reject_masks, g["reject_mask"])
    tol = RTOL_GOLDEN if exact else RTOL_DENSE
    assert scaled_err(samples, g["samples"]) <= tol
    assert scaled_err(momenta, g["momenta"]) <= tol
    # aliasing after getSamples (src/HMC.py:148): integrator.q is ensemble.q = last sample
    assert hmc.integrator.q is ens.q
    assert scaled_err(ens.q, g["samples"][:, :, -1]) <= tol


INTEG = [
    ("G1_leapfrog_harmonic", "Leapfrog", ("_h0.1", "_h0.01")),
    ("G2_stormerverlet_harmonic", "Stormer-Verlet", ("_h0.1", "_h0.01")),
    ("G8b_leapfrog_mass", "Leapfrog", ("",)),
    ("G8c_stormerverlet_mass", "Stormer-Verlet", ("",)),
]


@pytest.mark.parametrize("name,method,sfxs", INTEG)
def test_integrator_classes_harmonic(P, name, method, sfxs):
    g = load_golden(name)
    pot = P.Harmonic(g["springConsts"])
    cls = P.Leapfrog if method == "Leapfrog" else P.StormerVerlet
    for s in sfxs:
        D, N = g["q0" + s].shape
        ens = P.Ensemble(D, N)
        ens.mass = g["mass" + s].copy()
        ens.q[...] = g["q0" + s]
        ens.p[...] = g["p0" + s]
        integ = cls(ens, float(g["stepSize" + s]), float(g["finalTime" + s]), pot.gradient)
        assert integ.numSteps == int(g["numSteps" + s])
        q, p = integ.integrate()
        assert q is ens.q and p is ens.p  # in place and aliased (src/integrator.py:123)
        assert scaled_err(q, g["q" + s]) <= RTOL_GOLDEN
        assert scaled_err(p, g["p" + s]) <= RTOL_GOLDEN
        assert scaled_err(integ.v, g["v" + s]) <= RTOL_GOLDEN
        # bit-exact vs the oracle
        qo, po = np.ascontiguousarray(g["q0" + s]), np.ascontiguousarray(g["p0" + s])
        vo = orc.integrate(orc.pot_harmonic(g["springConsts"]), method, qo, po, g["mass" + s],
                           float(g["stepSize" + s]), int(g["numSteps" + s]))
        assert np.array_equal(q, qo) and np.array_equal(p, po) and np.array_equal(integ.v, vo)


def test_leapfrog_rosenbrock_vs_golden(P):
    g = load_golden("G5b_leapfrog_rosenbrock_d32")
    pot = P.Rosenbrock(32, float(g["a"]), float(g["b"]), float(g["s"]))
    ens = P.Ensemble(32, int(g["N"]))
    ens.q[...] = g["q0"]
    ens.p[...] = g["p0"]
    q, p = P.Leapfrog(ens, float(g["stepSize"]), float(g["finalTime"]), pot).integrate()
    assert scaled_err(q, g["q"]) <= RTOL_GOLDEN and scaled_err(p, g["p"]) <= RTOL_GOLDEN


@pytest.mark.parametrize("method", ["Leapfrog", "Stormer-Verlet"])
@pytest.mark.parametrize("D", [8, 32, 72, 96, 100, 128])
def test_dense_integrate_vs_oracle(P, D, method):
    rs = np.random.RandomState(D)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    Pm = 0.5 * (Pm + Pm.T)
    mu = rs.standard_normal(D)
    N = 70  # ragged: 4 full 16-chain tiles + 6
    mass = 1.0 + (np.arange(N) % 3) * 0.5
    pot = P.GaussianDense(mu, precision=Pm, const=0.0)
    for m in (None, mass):
        ens = P.Ensemble(D, N)
        if m is not None:
            ens.mass = m.copy()
        ens.q[...] = rs.standard_normal((D, N)) * 2
        ens.p[...] = rs.standard_normal((D, N))
        qo, po = ens.q.copy(), ens.p.copy()
        cls = P.Leapfrog if method == "Leapfrog" else P.StormerVerlet
        integ = cls(ens, 0.1, 1.0, pot.gradient)
        q, p = integ.integrate()
        vo = orc.integrate(orc.pot_gauss_dense(mu, Pm), method, qo, po, m, 0.1, 10)
        assert scaled_err(q, qo) <= RTOL_DENSE and scaled_err(p, po) <= RTOL_DENSE
        assert scaled_err(integ.v, vo) <= RTOL_DENSE


def test_dense_matvec_layout_with_asymmetric_matrix(P):
    """A symmetric precision would hide a transposed A-fragment map (cdna guide: 'A=I check
    with ASYMMETRIC B'); feed a deliberately non-symmetric matrix straight to the handle."""
    D, N = 128, 33
    rs = np.random.RandomState(1)
    M = rs.standard_normal((D, D))
    mu = rs.standard_normal(D)
    pot = P.GaussianDense(mu, precision=M, const=0.25, symmetrize=False)
    q = rs.standard_normal((D, N))
    U, g = pot.value_and_gradient(q)
    g_ref = M @ (q - mu[:, None])
    U_ref = 0.5 * np.sum((q - mu[:, None]) * g_ref, axis=0) + 0.25
    assert scaled_err(g, g_ref) <= 1e-13
    assert scaled_err(U, U_ref) <= 1e-13
    Uo, go = orc.potential(orc.pot_gauss_dense(mu, M, 0.25), q, want_grad=True)
    assert scaled_err(g, go) <= 1e-13 and scaled_err(U, Uo) <= 1e-13


def test_known_answers(P):
    g = load_golden("G10_known_answers")
    # src/tests/test_potential.py:13-25: harmonic potential at (3, 4) with k = (2, 3) is 33
    U = P.harmonicPotentialND(g["harmonic_q"], g["harmonic_k"])
    assert U[0] == 33.0
    assert np.array_equal(U, g["harmonic_U"])
    assert P.harmonicPotentialND(np.array([3.0, 4.0]), np.array([2, 3])) == 33.0


@pytest.mark.parametrize("kind,D", [("harmonic", 1), ("harmonic", 3), ("diag", 5), ("diag", 33),
                                     ("rosenbrock", 2), ("rosenbrock", 7), ("rosenbrock", 32),
                                     ("rosenbrock", 40)])
def test_potential_eval_bitexact_vs_oracle(P, kind, D):
    rs = np.random.RandomState(D)
    q = rs.standard_normal((D, 257))
    if kind == "harmonic":
        k = rs.uniform(0.5, 3, D)
        pot, op = P.Harmonic(k), orc.pot_harmonic(k)
    elif kind == "diag":
        mu, prec = rs.standard_normal(D), rs.uniform(0.5, 3, D)
        pot, op = P.GaussianDiag(mu, prec=prec, const=0.5), orc.pot_gauss_diag(mu, prec, 0.5)
    else:
        pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    U, g = pot.value_and_gradient(q)
    Uo, go = orc.potential(op, q, want_grad=True)
    assert np.array_equal(U, Uo) and np.array_equal(g, go)
    # (D,) calling convention of the reference
    assert pot(q[:, 3]) == Uo[3] and np.array_equal(pot.gradient(q[:, 3]), go[:, 3])


def test_weights_and_ratio_api(P):
    g = load_golden("G4_getsamples_dense_d8")
    pot = make_pot(P, g, "dense")
    ens = P.Ensemble(8, int(g["N"]))
    hmc = P.HMC(ens, 1.0, 0.1, None, potential=pot, verbose=False)
    newQ, newP = g["q_prop"][0], -g["p_prop"][0]
    r = hmc.getWeightsRatio(newQ, newP, g["q0"], g["p_draw"][0])
    assert np.max(np.abs(np.log(r) - np.log(g["ratio"][0]))) < 1e-9
    w = hmc.getWeights(g["q0"], g["p_draw"][0])
    wo, Ho = orc.weights(orc_pot(g, "dense"), np.ascontiguousarray(g["q0"]),
                         np.ascontiguousarray(g["p_draw"][0]))
    assert np.max(np.abs(np.log(w) + Ho)) < 1e-9


# ------------------------------------------------------------------------ Philox
def device_normal(lib, seed, stream, it, chain0, D, N, scale=1.0, scale_n=None):
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    out = empty((D, N), np.float64, 0)
    sn = as_device(scale_n, 0, np.float64) if scale_n is not None else None
    lib.call("pbbi_philox_normal", seed, stream, it, chain0, D, N, N, float(scale),
             sn.data_ptr() if sn is not None else None, lib.F64, 0, out.data_ptr(), stream_ptr(0))
    return to_numpy(out)


def device_uniform(lib, seed, it, chain0, N):
    from physicsbasedbayesianinference_amd._device import empty, stream_ptr, to_numpy
    u = empty((N,), np.float64, 0)
    lib.call("pbbi_philox_uniform", seed, it, chain0, N, lib.F64, 0, u.data_ptr(), stream_ptr(0))
    return to_numpy(u)


def test_philox_device_matches_oracle(lib):
    """Integer Philox part bit-exact (uniforms); the single-precision Box-Muller part of the
    normals agrees with the host mirror to 5e-6 (hardware v_log/v_sin/v_cos vs libm)."""
    D, N = 19, 4000
    z = device_normal(lib, 77, lib.STREAM_MOMENTUM, 3, 123456789012, D, N, 1.5)
    zo = orc.philox_normal(77, orc.STREAM_MOMENTUM, 3, 123456789012, D, N, 1.5)
    assert np.max(np.abs(z - zo)) < 5e-6 * 1.5 * 6
    assert np.array_equal(device_uniform(lib, 77, 3, 5, N), orc.philox_uniform(77, 3, 5, N))
    # float draws widened to double: every value is exactly representable in float32 / scale
    z1 = device_normal(lib, 77, lib.STREAM_MOMENTUM, 3, 0, D, N)
    assert np.array_equal(z1, z1.astype(np.float32).astype(np.float64))


def test_philox_normal_distribution(lib):
    """The device draws are N(0,1): moments, tails and a KS test on 1.3M variates."""
    from scipy import stats
    z = device_normal(lib, 5, lib.STREAM_MOMENTUM, 0, 0, 128, 10000)
    assert abs(z.mean()) < 4e-3 and abs(z.std() - 1) < 3e-3
    assert abs(stats.skew(z.ravel())) < 0.01 and abs(stats.kurtosis(z.ravel())) < 0.02
    assert stats.kstest(z.ravel()[::7], "norm").pvalue > 1e-3
    assert 3.5 < np.abs(z).max() < 6.8
    # rows of one Philox block (d, d+4, d+8, d+12) and different blocks are uncorrelated
    c = np.corrcoef(z[:32])
    assert np.max(np.abs(c - np.eye(32))) < 0.05
    # sharding: chain offset reproduces the matching columns; iteration changes the draw
    zs = device_normal(lib, 5, lib.STREAM_MOMENTUM, 0, 3000, 128, 500)
    assert np.array_equal(zs, z[:, 3000:3500])
    assert not np.array_equal(z, device_normal(lib, 5, lib.STREAM_MOMENTUM, 1, 0, 128, 10000))


@pytest.mark.parametrize("kind,D,N,method,mass", [
    ("std", 1, 32, "Leapfrog", False), ("rosenbrock", 32, 200, "Leapfrog", False),
    ("dense", 128, 150, "Leapfrog", False), ("dense", 128, 150, "Leapfrog", True),
    ("dense", 24, 64, "Leapfrog", False), ("dense", 64, 70, "Stormer-Verlet", True),
    ("diag", 5, 100, "Stormer-Verlet", True)])
def test_hmc_run_philox_vs_oracle(P, lib, kind, D, N, method, mass):
    """pbbi_hmc_run (in-kernel draws): the oracle replays the same iterations from the draws
    that pbbi_philox_normal / pbbi_philox_uniform return for the same counters (documented to
    be bit-identical to the in-kernel draws)."""
    S, L, h = 4, 10, 0.1 if kind != "rosenbrock" else 0.01
    rs = np.random.RandomState(D)
    if kind == "dense":
        A = rs.standard_normal((D, D))
        Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
        Pm = 0.5 * (Pm + Pm.T)
        pot, op = P.GaussianDense(None, precision=Pm, const=0.0), orc.pot_gauss_dense(np.zeros(D), Pm)
    elif kind == "std":
        pot, op = P.StandardGaussian(D), orc.pot_gauss_diag(np.zeros(D), np.ones(D))
    elif kind == "diag":
        mu, prec = rs.standard_normal(D), rs.uniform(0.5, 2, D)
        pot, op = P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec)
    else:
        pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    m = (1.0 + (np.arange(N) % 4) * 0.5) if mass else None
    ens = P.Ensemble(D, N)
    if mass:
        ens.mass = m.copy()
    seed, chain0, iter0, kT = 2024, 1000, 7, 1.0
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, method=method, rng="philox", seed=seed,
                kdk_fma=False, verbose=False)  # reference operation order: exact comparison below
    assert hmc.integrator.numSteps == L
    samples, momenta = hmc.getSamples(S, 1.0 / kB, 0.7, chain0=chain0, iter0=iter0)
    q = device_normal(lib, seed, lib.STREAM_POSITION, iter0, chain0, D, N, 0.7)
    pstd = np.sqrt((m if mass else np.ones(N)) * kT)
    exact = kind != "dense"
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, iter0 + i, chain0, D, N, 1.0, pstd)
        u = device_uniform(lib, seed, iter0 + i, chain0, N)
        ratio, rej = orc.hmc_iter(op, method, q, p, u, m, h, L)
        assert np.array_equal(hmc.reject_masks[i], rej), f"iteration {i}"
        if exact:
            assert np.array_equal(samples[:, :, i], q) and np.array_equal(momenta[:, :, i], p)
        else:
            assert scaled_err(samples[:, :, i], q) <= RTOL_DENSE
            assert scaled_err(momenta[:, :, i], p) <= RTOL_DENSE


# ------------------------------------------------------------------ edge cases
def test_empty_and_tiny_ensembles(P):
    pot = P.StandardGaussian(2)
    for N in (0, 1):
        np.random.seed(0)
        ens = P.Ensemble(2, N)
        s, m = P.HMC(ens, 1.0, 0.1, None, potential=pot, verbose=False).getSamples(3, 1 / kB, 1.0)
        assert s.shape == (2, N, 3) and m.shape == (2, N, 3)
    s, m = P.HMC(P.Ensemble(2, 5), 1.0, 0.1, None, potential=pot, verbose=False).getSamples(
        0, 1 / kB, 1.0)
    assert s.shape == (2, 5, 0)


def test_zero_steps_keeps_state(P):
    """numSteps = int(0.05/0.1) = 0: nothing moves, ratio == 1, nothing rejected."""
    pot = P.StandardGaussian(3)
    np.random.seed(3)
    ens = P.Ensemble(3, 40)
    hmc = P.HMC(ens, 0.05, 0.1, None, potential=pot, verbose=False)
    assert hmc.integrator.numSteps == 0
    s, m = hmc.getSamples(2, 1 / kB, 1.0)
    assert np.array_equal(s[:, :, 0], s[:, :, 1]) and not hmc.reject_masks.any()
    assert np.all(hmc.ratios == 1.0)


def test_noncompat_rejected_momentum_is_the_draw(P, lib):
    g = load_golden("G6_getsamples_rejects")
    pot = make_pot(P, g, "std")
    q, p, u = g["q0"], g["p_draw"][0], g["u"][0]
    qo, po, ratio, rej = gpu_hmc_iter(lib, pot, "Leapfrog", q, p, u, None, 1.5, 2, compat=False)
    assert rej.sum() > 50
    assert np.array_equal(po[:, rej], p[:, rej]) and np.array_equal(qo[:, rej], q[:, rej])
    qc, pc, _, rejc = gpu_hmc_iter(lib, pot, "Leapfrog", q, p, u, None, 1.5, 2, compat=True)
    assert np.array_equal(rej, rejc) and np.array_equal(pc[:, rej], q[:, rej])
    assert np.array_equal(po[:, ~rej], pc[:, ~rej])


def test_error_behaviour(P, lib):
    pot = P.StandardGaussian(2)
    ens = P.Ensemble(2, 4)
    with pytest.raises(ValueError, match="Invalid integration method selected."):
        P.HMC(ens, 1.0, 0.1, None, potential=pot, method="Euler")       # src/HMC.py:70-71
    with pytest.raises(TypeError):
        P.HMC(ens, 1.0, 0.1, None, potential=lambda q: float(q[0]) ** 2)  # untraceable: no host fallback
    with pytest.raises(NotImplementedError):
        P.Integrator(ens, 0.1, 1.0, pot.gradient).integrate()           # src/integrator.py:87-91
    with pytest.raises(IndexError):
        ens.particle(5)                                                 # src/ensemble.py:102-107
    with pytest.raises(ValueError):                                     # D mismatch
        P.Leapfrog(P.Ensemble(3, 4), 0.1, 1.0, pot).integrate()


# ------------------------------------------------------------------ sharded API / properties
def test_get_samples_sharded_single_process_matches_class_api(P):
    """distributed.get_samples_sharded without a process group == HMC.getSamples (philox and
    numpy-stream modes), returned as (D, N, S) device views."""
    from physicsbasedbayesianinference_amd.distributed import get_samples_sharded
    D, N, S = 16, 200, 3
    rs = np.random.RandomState(3)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    pot = P.GaussianDense(rs.standard_normal(D), precision=Pm, const=0.0)
    for rng in ("philox", "numpy"):
        np.random.seed(11)
        s, m, hmc = get_samples_sharded(pot, D, N, 1.0, 0.1, S, 1 / kB, 1.0, rng=rng, seed=5)
        np.random.seed(11)
        ens = P.Ensemble(D, N)
        s2, m2 = P.HMC(ens, 1.0, 0.1, None, potential=pot, rng=rng, seed=5, verbose=False).getSamples(
            S, 1 / kB, 1.0)
        assert tuple(s.shape) == (D, N, S)
        assert np.array_equal(s.cpu().numpy(), s2) and np.array_equal(m.cpu().numpy(), m2)


def test_sharding_invariance_on_device(P, lib):
    """Philox mode: chains [0, N) in one call == two half-ensembles with chain0 offsets, bit for
    bit (what makes the 8-GPU run reproduce the 1-GPU chains)."""
    D, N, S = 128, 512, 3
    rs = np.random.RandomState(4)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    pot = P.GaussianDense(None, precision=Pm, const=0.0)

    def run(n, chain0):
        hmc = P.HMC(P.Ensemble(D, n), 1.0, 0.1, None, potential=pot, rng="philox", seed=9,
                    verbose=False)
        return hmc.getSamples(S, 1 / kB, 1.0, chain0=chain0)[0]
    full = run(N, 0)
    lo, hi = run(N // 2, 0), run(N // 2, N // 2)
    assert np.array_equal(full[:, :N // 2], lo) and np.array_equal(full[:, N // 2:], hi)


def test_full_size_properties_c2(P, lib):
    """Size-independent properties at BASELINE's full C2 size (D=128, 65 536 chains):
    determinism, linearity of the Gaussian flow, time reversibility, energy error ~ h^2."""
    import torch
    from physicsbasedbayesianinference_amd._device import empty, stream_ptr
    D, N, L, h = 128, 65536, 10, 0.1
    A = np.random.RandomState(0).standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    Pm = 0.5 * (Pm + Pm.T)
    pot = P.GaussianDense(None, precision=Pm, const=0.0)
    g = torch.Generator(device="cuda").manual_seed(1)
    q0 = torch.randn((D, N), dtype=torch.float64, device="cuda", generator=g)
    p0 = torch.randn((D, N), dtype=torch.float64, device="cuda", generator=g)

    def leap(q, p, hh=h, steps=L):
        q, p = q.clone(), p.clone()
        lib.call("pbbi_leapfrog", pot.handle, q.data_ptr(), p.data_ptr(), None, N, N, hh, steps,
                 stream_ptr(0))
        return q, p

    def energy(q, p):
        H = empty((N,), np.float64, 0)
        lib.call("pbbi_energy", pot.handle, q.data_ptr(), p.data_ptr(), None, N, N, H.data_ptr(),
                 None, stream_ptr(0))
        return H
    q1, p1 = leap(q0, p0)
    q1b, p1b = leap(q0, p0)
    assert torch.equal(q1, q1b) and torch.equal(p1, p1b)                       # deterministic
    qs, ps = leap(3.0 * q0, 3.0 * p0)                                          # linear flow (mu = 0)
    assert float((qs - 3.0 * q1).abs().max() / q1.abs().max()) < 1e-13
    qr, pr = leap(q1, -p1)                                                     # reversible
    assert float((qr - q0).abs().max()) < 1e-11 and float((pr + p0).abs().max()) < 1e-11
    dH = (energy(q1, p1) - energy(q0, p0)).abs()
    q2, p2 = leap(q0, p0, h / 2, 2 * L)
    dH2 = (energy(q2, p2) - energy(q0, p0)).abs()
    assert 3.0 < float(dH.mean() / dH2.mean()) < 5.0                           # 2nd order: ~4x


# ------------------------------------------------------------------ large-D dense path (streaming GEMM)
def _dense_problem(D, seed=0):
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    return 0.5 * (Pm + Pm.T), rs.standard_normal(D) * 0.5


@pytest.mark.parametrize("D,N,dtype,tol", [(200, 150, "float64", 1e-12), (256, 130, "float64", 1e-12),
                                           (384, 70, "float32", 2e-5)])
def test_big_dense_eval_vs_oracle(P, D, N, dtype, tol):
    Pm, mu = _dense_problem(D)
    pot = P.GaussianDense(mu, precision=Pm, const=0.25, dtype=dtype)
    q = np.random.RandomState(1).standard_normal((D, N))
    U, g = pot.value_and_gradient(q)
    Uo, go = orc.potential(orc.pot_gauss_dense(mu, Pm, 0.25), q, want_grad=True)
    assert scaled_err(g, go) <= tol and scaled_err(U, Uo) <= tol * 10


def test_big_dense_asymmetric_matrix_layout(P):
    """P^T is what the kernel reads: a non-symmetric matrix must still give grad = P x."""
    D, N = 160, 40
    rs = np.random.RandomState(2)
    M = rs.standard_normal((D, D)) / np.sqrt(D)
    pot = P.GaussianDense(None, precision=M, const=0.0, symmetrize=False)
    q = rs.standard_normal((D, N))
    assert scaled_err(pot.gradient(q), M @ q) <= 1e-13


@pytest.mark.parametrize("D,mass,method,L", [(192, False, "Leapfrog", 10), (300, True, "Leapfrog", 10),
                                             (192, True, "Stormer-Verlet", 10),
                                             (260, False, "Stormer-Verlet", 0),
                                             (200, True, "Leapfrog", 0),
                                             # 128 < D <= 256, L >= 1: integrate() on the streamed-P kernel (MODE 1)
                                             (256, True, "Leapfrog", 7), (160, False, "Stormer-Verlet", 5),
                                             (129, True, "Stormer-Verlet", 1), (256, False, "Leapfrog", 1)])
def test_big_dense_integrate_vs_oracle(P, D, mass, method, L):
    Pm, mu = _dense_problem(D, 3)
    N = 90
    pot = P.GaussianDense(mu, precision=Pm, const=0.0)
    rs = np.random.RandomState(4)
    ens = P.Ensemble(D, N)
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    if mass:
        ens.mass = m.copy()
    ens.q[...] = rs.standard_normal((D, N))
    ens.p[...] = rs.standard_normal((D, N))
    qo, po = ens.q.copy(), ens.p.copy()
    cls = P.Leapfrog if method == "Leapfrog" else P.StormerVerlet
    integ = cls(ens, 0.1, 0.1 * L + 1e-9, pot)
    assert integ.numSteps == L
    q, p = integ.integrate()
    vo = orc.integrate(orc.pot_gauss_dense(mu, Pm), method, qo, po, m, 0.1, L)
    assert scaled_err(q, qo) <= RTOL_DENSE and scaled_err(p, po) <= RTOL_DENSE
    assert scaled_err(integ.v, vo) <= RTOL_DENSE


@pytest.mark.parametrize("method,L", [("Stormer-Verlet", 6), ("Stormer-Verlet", 0), ("Leapfrog", 0)])
def test_big_dense_hmc_iter_methods_vs_oracle(P, lib, method, L):
    """D > 128: Stormer-Verlet and the zero-step trajectory on the GEMM path (padded input stride
    is covered by gpu_hmc_iter's contiguous arrays; masks equal, state within the dense tolerance)."""
    D, N, h = 200, 150, 0.3
    Pm, mu = _dense_problem(D, 8)
    pot, op = P.GaussianDense(mu, precision=Pm, const=0.0), orc.pot_gauss_dense(mu, Pm)
    rs = np.random.RandomState(9)
    q, p, u = rs.standard_normal((D, N)), rs.standard_normal((D, N)), rs.uniform(size=N)
    m = 1.0 + (np.arange(N) % 3) * 0.5
    p *= np.sqrt(m)
    qo, po, ratio, rej = gpu_hmc_iter(lib, pot, method, q, p, u, m, h, L)
    q_or, p_or = q.copy(), p.copy()
    r_or, rej_or = orc.hmc_iter(op, method, q_or, p_or, u, m, h, L)
    assert np.array_equal(rej, rej_or)
    assert scaled_err(qo, q_or) <= RTOL_DENSE and scaled_err(po, p_or) <= RTOL_DENSE
    assert np.max(np.abs(np.log(ratio) - np.log(r_or))) < 1e-8
    if L:
        assert 0 < rej.sum() < N


@pytest.mark.parametrize("rng", ["numpy", "philox"])
def test_big_dense_getsamples_vs_oracle(P, lib, rng):
    D, N, S, L, h = 256, 140, 3, 10, 0.1
    Pm, mu = _dense_problem(D, 5)
    pot, op = P.GaussianDense(mu, precision=Pm, const=0.0), orc.pot_gauss_dense(mu, Pm)
    m = 1.0 + (np.arange(N) % 2) * 0.5
    ens = P.Ensemble(D, N)
    ens.mass = m.copy()
    np.random.seed(21)
    hmc = P.HMC(ens, 1.0, h, None, potential=pot, rng=rng, seed=6, verbose=False)
    samples, momenta = hmc.getSamples(S, 1 / kB, 1.0, chain0=50)
    if rng == "numpy":
        ref = orc.get_samples_numpy_stream(op, "Leapfrog", D, N, S, 1.0, h, 1 / kB, 1.0, 21, mass=m)
        assert np.array_equal(hmc.reject_masks, ref["reject_mask"])
        assert scaled_err(samples, ref["samples"]) <= RTOL_DENSE
        assert scaled_err(momenta, ref["momenta"]) <= RTOL_DENSE
    else:
        q = device_normal(lib, 6, lib.STREAM_POSITION, 0, 50, D, N, 1.0)
        for i in range(S):
            p = device_normal(lib, 6, lib.STREAM_MOMENTUM, i, 50, D, N, 1.0, np.sqrt(m))
            u = device_uniform(lib, 6, i, 50, N)
            _, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, m, h, L)
            assert np.array_equal(hmc.reject_masks[i], rej)
            assert scaled_err(samples[:, :, i], q) <= RTOL_DENSE
            assert scaled_err(momenta[:, :, i], p) <= RTOL_DENSE


def test_c5_shape_fp32_vs_fp64_oracle(P, lib):
    """BASELINE config 5's shape at small N: D = 4096 dense precision in fp32, h = 0.05, L = 10,
    against the fp64 oracle.  fp32 MFMA accumulates 4096-term dot products in single precision:
    tolerance 2e-4 scaled; decisions must agree wherever |log u - log ratio| > 1e-2."""
    D, N, L, h = 4096, 64, 10, 0.05
    Pm, _ = _dense_problem(D, 0)
    pot = P.GaussianDense(None, precision=Pm, const=0.0, dtype="float32")
    rs = np.random.RandomState(9)
    q0 = rs.standard_normal((D, N)).astype(np.float32).astype(np.float64)
    p0 = rs.standard_normal((D, N)).astype(np.float32).astype(np.float64)
    u = rs.uniform(size=N).astype(np.float32).astype(np.float64)
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    import torch
    qd, pd, ud = (as_device(x, 0, np.float32) for x in (q0, p0, u))
    qo, po = empty((D, N), np.float32, 0), empty((D, N), np.float32, 0)
    ro, rj = empty((N,), np.float32, 0), empty((N,), np.uint8, 0)
    lib.call("pbbi_hmc_iter", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(), None,
             qo.data_ptr(), po.data_ptr(), ro.data_ptr(), rj.data_ptr(), N, N, h, L, 1, stream_ptr(0))
    torch.cuda.synchronize()
    qr, pr = q0.copy(), p0.copy()
    ratio, rej = orc.hmc_iter(orc.pot_gauss_dense(np.zeros(D), Pm), "Leapfrog", qr, pr, u, None, h, L)
    assert scaled_err(to_numpy(qo).astype(np.float64), qr) <= 2e-4
    assert scaled_err(to_numpy(po).astype(np.float64), pr) <= 2e-4
    clear = np.abs(np.log(u) - np.minimum(0.0, np.log(ratio))) > 1e-2
    assert clear.sum() > N // 2
    assert np.array_equal(to_numpy(rj).astype(bool)[clear], rej[clear])


@pytest.mark.parametrize("method,mass,L", [("Leapfrog", False, 10), ("Leapfrog", True, 5), ("Stormer-Verlet", True, 6)])
def test_big_wide_tile_fp32_vs_oracle(P, lib, method, mass, L):
    """k_big_gemm_wide (fp32, zero mean, whole 256-row / 256-chain tiles): D = 512, N = 768 = 2 x 3 block
    tiles, against the fp64 oracle (2e-4 scaled as at C5's shape; decisions equal where they are not
    marginal) and against the 128 x 128 kernel, which the same call takes for N - 4 chains."""
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    import torch
    D, N, h = 512, 768, 0.15
    Pm, _ = _dense_problem(D, 12)
    pot = P.GaussianDense(None, precision=Pm, const=0.0, dtype="float32")
    rs = np.random.RandomState(13)
    f32 = lambda x: x.astype(np.float32).astype(np.float64)
    m = f32(1.0 + (np.arange(N) % 3) * 0.5) if mass else None
    q0, u = f32(rs.standard_normal((D, N))), f32(rs.uniform(size=N))
    p0 = f32(rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0))
    mid = 0 if method == "Leapfrog" else 1

    def run(n):
        qd, pd, ud = (as_device(np.ascontiguousarray(x[..., :n]), 0, np.float32) for x in (q0, p0, u))
        md = as_device(m[:n], 0, np.float32) if mass else None
        qo, po = empty((D, n), np.float32, 0), empty((D, n), np.float32, 0)
        ro, rj = empty((n,), np.float32, 0), empty((n,), np.uint8, 0)
        lib.call("pbbi_hmc_iter", pot.handle, mid, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
                 md.data_ptr() if mass else None, qo.data_ptr(), po.data_ptr(), ro.data_ptr(), rj.data_ptr(),
                 n, n, h, L, 1, stream_ptr(0))
        torch.cuda.synchronize()
        return (to_numpy(qo).astype(np.float64), to_numpy(po).astype(np.float64),
                to_numpy(ro).astype(np.float64), to_numpy(rj).astype(bool))

    qw, pw, rw, jw = run(N)
    qr, pr = q0.copy(), p0.copy()
    ratio, rej = orc.hmc_iter(orc.pot_gauss_dense(np.zeros(D), Pm), method, qr, pr, u, m, h, L)
    assert scaled_err(qw, qr) <= 2e-4 and scaled_err(pw, pr) <= 2e-4
    clear = np.abs(np.log(u) - np.minimum(0.0, np.log(ratio))) > 1e-2
    assert clear.sum() > N // 2 and np.array_equal(jw[clear], rej[clear])
    assert 0 < rej.sum() < N
    qn, pn, rn, jn = run(N - 4)                       # ragged: the 128 x 128 kernel
    same = jw[:N - 4] == jn
    assert same.mean() > 0.99
    assert scaled_err(qw[:, :N - 4][:, same], qn[:, same]) <= 2e-5
    assert np.max(np.abs(np.log(rw[:N - 4]) - np.log(rn))) < 5e-3


# ------------------------------------------------------------------ two-lanes-per-chain Rosenbrock kernel
@pytest.mark.parametrize("D,N,mass,compat", [(32, 1000, False, True), (32, 77, True, False),
                                             (20, 130, True, True), (17, 33, False, True),
                                             (31, 64, False, False)])
def test_rosenbrock_two_lane_kernel_bitexact(P, lib, D, N, mass, compat):
    """kernels_lane2.hip (16 < D <= 32, Leapfrog): bit-exact vs the oracle for full / padded D,
    ragged N, non-unit masses, both momentum-restore conventions, with rejections present."""
    rs = np.random.RandomState(D * 1000 + N)
    pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    q = 1.0 + 0.3 * rs.standard_normal((D, N))
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    h, L = 0.05, 10  # large enough step for rejections in the big case
    n_rej = 0
    for it in range(3):
        p = rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0)
        u = rs.uniform(size=N)
        qo, po, ratio, rej = gpu_hmc_iter(lib, pot, "Leapfrog", q, p, u, m, h, L, compat=compat)
        q_or, p_or = q.copy(), p.copy()
        r_or, rej_or = orc.hmc_iter(op, "Leapfrog", q_or, p_or, u, m, h, L,
                                    compat=orc.COMPAT_P_FROM_OLDQ if compat else 0)
        assert np.array_equal(rej, rej_or)
        assert np.array_equal(qo, q_or) and np.array_equal(po, p_or)
        fin = np.isfinite(r_or) & (r_or > 0)
        assert np.max(np.abs(np.log(ratio[fin]) - np.log(r_or[fin]))) < 1e-9
        n_rej += int(rej.sum())
        q = qo
    assert n_rej > 0 or N < 1000


# ------------------------------------------------------------------ more edge cases
@pytest.mark.parametrize("D", [1, 3, 128, 130])
@pytest.mark.parametrize("N", [0, 1, 17])
def test_dense_tiny_shapes(P, D, N):
    """Dense Gaussian at degenerate sizes (D = 1 pads to a 32-wide tile; D = 130 takes the GEMM path)."""
    rs = np.random.RandomState(D + N)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    Pm = 0.5 * (Pm + Pm.T)
    mu = rs.standard_normal(D)
    pot, op = P.GaussianDense(mu, precision=Pm, const=0.0), orc.pot_gauss_dense(mu, Pm)
    np.random.seed(3)
    ens = P.Ensemble(D, N)
    hmc = P.HMC(ens, 0.5, 0.1, None, potential=pot, verbose=False)
    s, m = hmc.getSamples(2, 1 / kB, 1.0)
    assert s.shape == (D, N, 2)
    if N:
        ref = orc.get_samples_numpy_stream(op, "Leapfrog", D, N, 2, 0.5, 0.1, 1 / kB, 1.0, 3)
        assert np.array_equal(hmc.reject_masks, ref["reject_mask"])
        assert scaled_err(s, ref["samples"]) <= RTOL_DENSE and scaled_err(m, ref["momenta"]) <= RTOL_DENSE


def test_hmc_run_without_optional_outputs(P, lib):
    """momenta_out / reject_out / ratio_out are optional in pbbi_hmc_run; q_state carries over."""
    import torch
    from physicsbasedbayesianinference_amd._device import empty, stream_ptr
    for pot, D in ((P.GaussianDense(None, precision=np.eye(40) * 2.0, const=0.0), 40), (P.Rosenbrock(32), 32),
                   (P.StandardGaussian(5), 5)):
        N, S = 200, 3
        q = torch.full((D, N), 0.9, dtype=torch.float64, device="cuda")
        q2 = q.clone()
        s1 = empty((S, D, N), np.float64, 0)
        s2, m2 = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
        r2 = empty((S, N), np.uint8, 0)
        lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, s1.data_ptr(), None, None, None,
                 N, N, 0.01, 5, S, 1, 3, 0, 0, 1.0, stream_ptr(0))
        lib.call("pbbi_hmc_run", pot.handle, 0, q2.data_ptr(), None, s2.data_ptr(), m2.data_ptr(),
                 r2.data_ptr(), None, N, N, 0.01, 5, S, 1, 3, 0, 0, 1.0, stream_ptr(0))
        torch.cuda.synchronize()
        assert torch.equal(s1, s2) and torch.equal(q, q2) and torch.equal(q, s1[S - 1])


def test_padded_leading_stride(P, lib):
    """ldn > N: kernels honour the leading stride of the (D, N) layout."""
    import torch
    from physicsbasedbayesianinference_amd._device import stream_ptr
    for pot in (P.GaussianDense(None, precision=np.eye(16) + 0.1, const=0.0), P.Harmonic(np.arange(1, 7.0))):
        D, N, ld = pot.numDimensions, 50, 64
        g = torch.Generator(device="cuda").manual_seed(0)
        qp = torch.randn((D, ld), dtype=torch.float64, device="cuda", generator=g)
        pp = torch.randn((D, ld), dtype=torch.float64, device="cuda", generator=g)
        qc, pc = qp[:, :N].contiguous(), pp[:, :N].contiguous()
        pad_q = qp[:, N:].clone()
        lib.call("pbbi_leapfrog", pot.handle, qp.data_ptr(), pp.data_ptr(), None, N, ld, 0.1, 7, stream_ptr(0))
        lib.call("pbbi_leapfrog", pot.handle, qc.data_ptr(), pc.data_ptr(), None, N, N, 0.1, 7, stream_ptr(0))
        torch.cuda.synchronize()
        assert torch.equal(qp[:, :N], qc) and torch.equal(pp[:, :N], pc)
        assert torch.equal(qp[:, N:], pad_q)  # padding columns untouched


# ------------------------------------------------------------------ SURVEY 8f rows 1 and 4
def test_linear_regression_posterior_sampling_and_moments(P):
    """Model -> potential for Bayesian linear regression, sampled with the ensemble kernels, with
    the moments taken by the on-device sample sink: posterior mean and marginal variances are
    recovered (statistical check) and the moments kernel equals NumPy on the same draws."""
    rs = np.random.RandomState(0)
    M, D, N, S = 200, 6, 4096, 100
    X = rs.standard_normal((M, D))
    w_true = rs.standard_normal(D)
    y = X @ w_true + 0.5 * rs.standard_normal(M)
    pot = P.linear_regression_posterior(X, y, noise_var=0.25, prior_precision=1.0)
    Sigma = np.linalg.inv(pot.precision)
    ens = P.Ensemble(D, N)
    # omega*T in (1.4, 2.0) for every eigen-direction of P: away from the k*pi resonances
    hmc = P.HMC(ens, 0.06, 0.006, None, potential=pot, rng="philox", seed=1, verbose=False)
    s_dev, _ = hmc.getSamples(S, 1 / kB, 1.0, device_output=True)
    mean, var = hmc.sampleMoments(s_dev[:, :, 50:])          # discard burn-in draws
    draws = s_dev[:, :, 50:].cpu().numpy()
    assert np.allclose(mean, draws.mean(axis=(1, 2)), rtol=0, atol=1e-12)
    assert np.allclose(var, draws.var(axis=(1, 2)), rtol=1e-10, atol=1e-14)
    assert np.max(np.abs(mean - pot.mean)) < 0.005
    assert np.max(np.abs(var / np.diag(Sigma) - 1.0)) < 0.1
    assert hmc.acceptRate > 0.9


# ------------------------------------------------------------------ D > 64 / fp32 chain-per-lane
def _stream_case(P, kind, D, rs, dtype="float64"):
    if kind == "harmonic":
        k = rs.uniform(0.5, 2.0, D)
        return P.Harmonic(k, dtype=dtype), orc.pot_harmonic(k)
    if kind == "diag":
        mu, prec = rs.standard_normal(D), rs.uniform(0.5, 2.0, D)
        return P.GaussianDiag(mu, prec=prec, const=0.25, dtype=dtype), orc.pot_gauss_diag(mu, prec, 0.25)
    return P.Rosenbrock(D, dtype=dtype), orc.pot_rosenbrock(D)


@pytest.mark.parametrize("kind,D,N,method,mass", [
    ("harmonic", 65, 100, "Leapfrog", False), ("diag", 100, 130, "Stormer-Verlet", True),
    ("diag", 72, 64, "Leapfrog", True), ("rosenbrock", 70, 200, "Leapfrog", False),
    ("rosenbrock", 128, 65, "Stormer-Verlet", False), ("rosenbrock", 257, 70, "Leapfrog", True)])
def test_streaming_lane_path_bit_exact(P, lib, kind, D, N, method, mass):
    """D > 64: the chain's q, v, a live in a device workspace (kernels_stream.hip); same
    arithmetic in the same order as the oracle => bit-identical q, p, masks, energies."""
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(D)
    pot, op = _stream_case(P, kind, D, rs)
    L, h = 7, (0.02 if kind == "rosenbrock" else 0.9)
    q = rs.standard_normal((D, N)) * (0.3 if kind == "rosenbrock" else 1.0)
    p = rs.standard_normal((D, N))
    u = rs.uniform(size=N)
    u[::3] = 1.5  # u > min(1, ratio) whatever the ratio: every third chain takes the reject branch
    m = (1.0 + (np.arange(N) % 4) * 0.5) if mass else None
    if mass:
        p *= np.sqrt(m)
    for compat in (True, False):
        qo, po, ratio, rej = gpu_hmc_iter(lib, pot, method, q, p, u, m, h, L, compat=compat)
        q_or, p_or = q.copy(), p.copy()
        r_or, rej_or = orc.hmc_iter(op, method, q_or, p_or, u, m, h, L,
                                      compat=orc.COMPAT_P_FROM_OLDQ if compat else 0)
        assert np.array_equal(rej, rej_or)
        assert np.array_equal(qo, q_or) and np.array_equal(po, p_or)
        assert np.max(np.abs(np.log(ratio) - np.log(r_or))) < 1e-8
    assert 0 < rej.sum() < N
    # potential / gradient / integrate through the class API
    U, g = pot.value_and_gradient(q)
    U_or, g_or = orc.potential(op, q, want_grad=True)
    assert np.array_equal(U, U_or) and np.array_equal(g, g_or)
    ens = P.Ensemble(D, N)
    ens.q, ens.p = q.copy(), p.copy()
    if mass:
        ens.mass = m.copy()
    cls = P.Leapfrog if method == "Leapfrog" else P.StormerVerlet
    qi, pi = cls(ens, h, L * h + 1e-9, pot).integrate()
    q2, p2 = q.copy(), p.copy()
    orc.integrate(op, method, q2, p2, m, h, L)
    assert np.array_equal(qi, q2) and np.array_equal(pi, p2)


def test_streaming_lane_path_philox_and_fp32(P, lib):
    """In-kernel draws on the streaming path (oracle replay from the device draws, bit-exact),
    and the fp32 build of the same kernels against the fp64 oracle (single-precision tolerance)."""
    D, N, S, L, h = 96, 150, 3, 5, 0.01
    pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    ens = P.Ensemble(D, N)
    seed, chain0, iter0 = 11, 5, 2
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, rng="philox", seed=seed, kdk_fma=False,
                verbose=False)
    samples, momenta = hmc.getSamples(S, 1.0 / kB, 0.3, chain0=chain0, iter0=iter0)
    q = device_normal(lib, seed, lib.STREAM_POSITION, iter0, chain0, D, N, 0.3)
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, iter0 + i, chain0, D, N, 1.0, np.ones(N))
        u = device_uniform(lib, seed, iter0 + i, chain0, N)
        _, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, None, h, L)
        assert np.array_equal(hmc.reject_masks[i], rej)
        assert np.array_equal(samples[:, :, i], q) and np.array_equal(momenta[:, :, i], p)
    # fp32: D = 32 (lane kernels are fp64-only, so this runs the streaming kernels) and D = 80
    for D32 in (32, 80):
        rs = np.random.RandomState(D32)
        pot32, op = _stream_case(P, "diag", D32, rs, dtype="float32")
        q = rs.standard_normal((D32, N)).astype(np.float32).astype(np.float64)
        p = rs.standard_normal((D32, N)).astype(np.float32).astype(np.float64)
        ens = P.Ensemble(D32, N)
        ens.q, ens.p = q.copy(), p.copy()
        qi, pi = P.Leapfrog(ens, 0.1, 1.0 + 1e-6, pot32).integrate()
        q2, p2 = q.copy(), p.copy()
        orc.integrate(op, "Leapfrog", q2, p2, None, 0.1, 10)
        assert scaled_err(qi, q2) <= 2e-5 and scaled_err(pi, p2) <= 2e-5
        U, g = pot32.value_and_gradient(q)
        U_or, g_or = orc.potential(op, q, want_grad=True)
        assert scaled_err(U, U_or) <= 1e-5 and scaled_err(g, g_or) <= 1e-5


# ------------------------------------------------------------------ user-defined potentials
@pytest.mark.parametrize("D", [11, 40])  # 11: register-resident plugin kernels; 40: workspace kernels
@pytest.mark.parametrize("method,mass", [("Leapfrog", False), ("Leapfrog", True), ("Stormer-Verlet", True)])
def test_custom_potential_polynomial_bit_exact(P, lib, method, mass, D):
    """CustomPotential: the user's C++ source inlined into the plugin kernels vs the same source
    compiled for the host inside the oracle.  Polynomial potential => bit-identical."""
    from custom_sources import QUARTIC
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    N, L, h = 300, 9, 0.07
    prm = [1.5, 0.75]
    pot, op = CustomPotential(D, QUARTIC, prm), orc.pot_custom(QUARTIC, D, prm)
    rs = np.random.RandomState(5)
    q, p, u = rs.standard_normal((D, N)), rs.standard_normal((D, N)), rs.uniform(size=N)
    u[::4] = 1.5  # forced rejects
    m = (1.0 + (np.arange(N) % 4) * 0.5) if mass else None
    for compat in (True, False):
        qo, po, ratio, rej = gpu_hmc_iter(lib, pot, method, q, p, u, m, h, L, compat=compat)
        q_or, p_or = q.copy(), p.copy()
        r_or, rej_or = orc.hmc_iter(op, method, q_or, p_or, u, m, h, L,
                                      compat=orc.COMPAT_P_FROM_OLDQ if compat else 0)
        assert np.array_equal(rej, rej_or) and 0 < rej.sum() < N
        assert np.array_equal(qo, q_or) and np.array_equal(po, p_or)
        assert np.max(np.abs(np.log(ratio) - np.log(r_or))) < 1e-8
    U, g = pot.value_and_gradient(q)
    U_or, g_or = orc.potential(op, q, want_grad=True)
    assert np.array_equal(U, U_or) and np.array_equal(g, g_or)
    assert pot.check_gradient(q[:, :20]) < 1e-6
    # weights / Hamiltonians through the class API
    ens = P.Ensemble(D, N)
    ens.q, ens.p = q.copy(), p.copy()
    if mass:
        ens.mass = m.copy()
    hmc = P.HMC(ens, 1.0, 0.1, None, potential=pot, verbose=False)
    w_or, _ = orc.weights(op, q, p, m)
    assert np.allclose(hmc.getWeights(q, p), w_or, rtol=1e-13)


def test_custom_potential_logistic_regression(P, lib):
    """A Bayesian model as a user potential: logistic regression, the data set travels in `params`.
    exp / log1p differ in the last ulp between the device and the host libm: 1e-12 tolerance.
    Sampling (in-kernel draws) is replayed by the oracle from the device draws, and the posterior
    mean agrees with a long NumPy-stream run of the same sampler (statistical check)."""
    from custom_sources import LOGISTIC, logistic_numpy, logistic_problem
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    X, y, lam, prm = logistic_problem(M=40, D=5)
    D, N = X.shape[1], 500
    pot, op = CustomPotential(D, LOGISTIC, prm), orc.pot_custom(LOGISTIC, D, prm)
    rs = np.random.RandomState(1)
    q = rs.standard_normal((D, N))
    U, g = pot.value_and_gradient(q)
    Un, gn = logistic_numpy(X, y, lam, q)
    assert scaled_err(U, Un) <= 1e-12 and scaled_err(g, gn) <= 1e-12
    assert pot.check_gradient(q[:, :10]) < 1e-6
    S, L, h, seed = 4, 8, 0.1, 3
    ens = P.Ensemble(D, N)
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, rng="philox", seed=seed, verbose=False)
    samples, momenta = hmc.getSamples(S, 1.0 / kB, 1.0)
    qs = device_normal(lib, seed, lib.STREAM_POSITION, 0, 0, D, N, 1.0)
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, 0, D, N, 1.0, np.ones(N))
        u = device_uniform(lib, seed, i, 0, N)
        _, rej = orc.hmc_iter(op, "Leapfrog", qs, p, u, None, h, L)
        assert np.array_equal(hmc.reject_masks[i], rej)
        assert scaled_err(samples[:, :, i], qs) <= 1e-11 and scaled_err(momenta[:, :, i], p) <= 1e-11
    assert 0.0 < hmc.reject_masks.mean() < 0.5
    # posterior mean: 4000 chains x 40 kept draws vs the Laplace-free ground truth from a long
    # independent run in the other RNG mode
    ens = P.Ensemble(D, 4000)
    hmc = P.HMC(ens, 0.8, 0.1, None, potential=pot, rng="philox", seed=9, verbose=False)
    s_dev, _ = hmc.getSamples(60, 1.0 / kB, 1.0, device_output=True)
    mean, var = hmc.sampleMoments(s_dev[:, :, 20:])
    np.random.seed(4)
    ens2 = P.Ensemble(D, 4000)
    s2, _ = P.HMC(ens2, 0.8, 0.1, None, potential=pot, verbose=False).getSamples(60, 1.0 / kB, 1.0)
    assert np.max(np.abs(mean - s2[:, :, 20:].mean(axis=(1, 2)))) < 0.02
    assert np.max(np.abs(var / s2[:, :, 20:].var(axis=(1, 2)) - 1.0)) < 0.05


def test_custom_potential_fp32_and_errors(P, lib):
    from custom_sources import QUARTIC
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    D, N = 7, 100
    pot32, op = CustomPotential(D, QUARTIC, [1.0, 0.5], dtype="float32"), orc.pot_custom(QUARTIC, D, [1.0, 0.5])
    rs = np.random.RandomState(2)
    q = rs.standard_normal((D, N)).astype(np.float32).astype(np.float64)
    p = rs.standard_normal((D, N)).astype(np.float32).astype(np.float64)
    ens = P.Ensemble(D, N)
    ens.q, ens.p = q.copy(), p.copy()
    qi, pi = P.Leapfrog(ens, 0.05, 0.5 + 1e-6, pot32).integrate()
    q2, p2 = q.copy(), p.copy()
    orc.integrate(op, "Leapfrog", q2, p2, None, 0.05, 10)
    assert scaled_err(qi, q2) <= 2e-5 and scaled_err(pi, p2) <= 2e-5
    with pytest.raises(RuntimeError, match="hipcc failed"):
        CustomPotential(D, "template <class Q> PBBI_FN T potential(const Q& q, int D, const T* prm) { return nope; }")
    with pytest.raises(lib.PbbiError):   # a file that is not a plugin
        h = C.c_void_p()
        lib.call("pbbi_potential_create_custom", b"/nonexistent/plugin.so", D, None, 0, lib.F64, 0,
                 C.byref(h))


def test_coin_toss_posterior(P):
    """SURVEY 8f row 1's first model (samples/NumpyroExamples/CoinToss): Beta(2,2) prior, 7 heads
    in 10 tosses, sampled in logit space; sigmoid(samples) must be Beta(9, 5)."""
    from physicsbasedbayesianinference_amd.custom import coin_toss_posterior
    pot = coin_toss_posterior(7, 10, a=2.0, b=2.0)
    ens = P.Ensemble(1, 8192)
    hmc = P.HMC(ens, 1.0, 0.1, None, potential=pot, rng="philox", seed=2, verbose=False)
    s, _ = hmc.getSamples(40, 1.0 / kB, 1.0)
    theta = 1.0 / (1.0 + np.exp(-s[0, :, 10:]))
    A, B = 9.0, 5.0
    assert abs(theta.mean() - A / (A + B)) < 3e-3
    assert abs(theta.var() / (A * B / ((A + B) ** 2 * (A + B + 1))) - 1.0) < 0.05
    assert hmc.acceptRate > 0.95
    # the reference's example: TWO coins with their own tosses and a uniform prior
    # (samples/NumpyroExamples/CoinToss/CoinToss.py:18-22): theta_j ~ Beta(k_j + 1, n_j - k_j + 1)
    k, n = np.array([62.0, 11.0]), np.array([100.0, 40.0])
    pot2 = coin_toss_posterior(k, n)
    assert pot2.numDimensions == 2
    hmc = P.HMC(P.Ensemble(2, 8192), 1.0, 0.1, None, potential=pot2, rng="philox", seed=3, verbose=False)
    s, _ = hmc.getSamples(40, 1.0 / kB, 1.0, jitter=0.5)   # omega*T is near pi for coin 2 at T = 1
    theta = 1.0 / (1.0 + np.exp(-s[:, :, 10:]))
    for j in range(2):
        A, B = k[j] + 1, n[j] - k[j] + 1
        assert abs(theta[j].mean() - A / (A + B)) < 3e-3
        assert abs(theta[j].var() / (A * B / ((A + B) ** 2 * (A + B + 1))) - 1.0) < 0.06
    with pytest.raises(ValueError):
        coin_toss_posterior([5], [3])


def test_dual_averaging_step_size(P):
    """adaptStepSize: from a far too small and a far too large step the ensemble-mean acceptance of
    a following run lands near the target (dense Gaussian, D = 48)."""
    D, N = 48, 4096
    rs = np.random.RandomState(0)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + 0.1 * np.eye(D))
    pot = P.GaussianDense(None, precision=0.5 * (Pm + Pm.T), const=0.0)
    steps = []
    for h0 in (0.01, 1.5):
        hmc = P.HMC(P.Ensemble(D, N), 1.0, h0, None, potential=pot, rng="philox", seed=5, verbose=False)
        h = hmc.adaptStepSize(1.0 / kB, 1.0, target=0.8, iterations=150)
        assert hmc.stepSize == h and hmc.integrator.numSteps == max(1, int(1.0 / h))
        hmc.getSamples(20, 1.0 / kB, 1.0)
        assert abs(np.minimum(1.0, hmc.ratios[5:]).mean() - 0.8) < 0.1, (h0, h)
        steps.append(h)
    assert abs(steps[0] / steps[1] - 1.0) < 0.35


def test_warmup_draws_are_disjoint_from_sampling_draws(P, lib):
    """adaptStepSize draws under its own Philox key (seed ^ WARMUP_SEED_MASK): the counter holds only
    32 iteration bits, so an iteration offset such as 1 << 40 would alias the sampling run's draws
    (iteration_lo32, include/pbbi.h).  pbbi_hmc_run / pbbi_philox_* refuse indices beyond 2^32."""
    from physicsbasedbayesianinference_amd.HMC import WARMUP_SEED_MASK
    from physicsbasedbayesianinference_amd._device import empty, stream_ptr
    seed, D, N = 5, 8, 64
    z = device_normal(lib, seed, lib.STREAM_MOMENTUM, 3, 0, D, N)
    zw = device_normal(lib, seed ^ WARMUP_SEED_MASK, lib.STREAM_MOMENTUM, 3, 0, D, N)
    assert not np.any(z == zw)
    assert not np.any(device_uniform(lib, seed, 3, 0, N) == device_uniform(lib, seed ^ WARMUP_SEED_MASK, 3, 0, N))
    # what the old offset did: iteration (1 << 32) + 3 would be iteration 3 again -> now an error
    q = empty((D, N), np.float64, 0)
    pot = P.StandardGaussian(D)
    for it0, S in (((1 << 32) + 3, 1), ((1 << 32) - 1, 2), (1 << 40, 1)):
        with pytest.raises(lib.PbbiError):
            lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, None, None, None, None, N, N, 0.1, 3,
                     S, 0, seed, it0, 0, 1.0, stream_ptr(0))
    with pytest.raises(lib.PbbiError):
        lib.call("pbbi_philox_normal", seed, lib.STREAM_MOMENTUM, 1 << 32, 0, D, N, N, 1.0, None, lib.F64, 0,
                 q.data_ptr(), stream_ptr(0))
    lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, None, None, None, None, N, N, 0.1, 3,
             1, 0, seed, (1 << 32) - 1, 0, 1.0, stream_ptr(0))  # the last valid index


@pytest.mark.parametrize("D,N,mass,compat", [(32, 2500, False, True), (32, 333, True, False),
                                             (23, 1000, False, True)])
def test_rosenbrock_run_with_many_rejections_bitexact(P, lib, D, N, mass, compat):
    """pbbi_hmc_run on the two-lane kernel (several iterations fused into one launch, the chain in
    registers in between) with a step large enough for frequent rejections: replayed by the oracle
    from the device draws, and by per-iteration pbbi_hmc_iter calls with the same draws --
    bit-exact, including chains rejected after earlier rejections."""
    S, L, h, seed, chain0, iter0 = 6, 10, 0.05, 77, 12345, 3
    pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    m = (1.0 + (np.arange(N) % 3) * 0.5) if mass else None
    ens = P.Ensemble(D, N)
    if mass:
        ens.mass = m.copy()
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, rng="philox", seed=seed, compat=compat,
                kdk_fma=False, verbose=False)
    samples, momenta = hmc.getSamples(S, 1.0 / kB, 0.3, chain0=chain0, iter0=iter0)
    q = 0.0 + device_normal(lib, seed, lib.STREAM_POSITION, iter0, chain0, D, N, 0.3)
    pstd = np.sqrt(m) if mass else np.ones(N)
    n_rej = 0
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, iter0 + i, chain0, D, N, 1.0, pstd)
        u = device_uniform(lib, seed, iter0 + i, chain0, N)
        qg, pg, ratio_g, rej_g = gpu_hmc_iter(lib, pot, "Leapfrog", q, p, u, m, h, L, compat=compat)
        r_or, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, m, h, L,
                                 compat=orc.COMPAT_P_FROM_OLDQ if compat else 0)
        assert np.array_equal(hmc.reject_masks[i], rej) and np.array_equal(rej_g, rej)
        assert np.array_equal(samples[:, :, i], q) and np.array_equal(momenta[:, :, i], p)
        assert np.array_equal(qg, q) and np.array_equal(pg, p)
        fin = np.isfinite(r_or) & (r_or > 0)
        assert np.max(np.abs(np.log(hmc.ratios[i][fin]) - np.log(r_or[fin]))) < 1e-9
        n_rej += int(rej.sum())
    assert n_rej > S  # rejections in several iterations, including after earlier rejections


def test_sample_chunks_equal_one_run(P, tmp_path):
    """sampleChunks: chunked long run (state and Philox counters carried on the GPU) == one
    getSamples call, bit for bit; the .npy spill holds the same (D, N, c) blocks."""
    D, N, S = 16, 300, 11
    rs = np.random.RandomState(1)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    pot = P.GaussianDense(rs.standard_normal(D), precision=0.5 * (Pm + Pm.T), const=0.0)
    kw = dict(potential=pot, rng="philox", seed=8, verbose=False)
    ref_s, ref_m = P.HMC(P.Ensemble(D, N), 1.0, 0.25, None, **kw).getSamples(S, 1 / kB, 1.0, chain0=7, iter0=2)
    hmc = P.HMC(P.Ensemble(D, N), 1.0, 0.25, None, **kw)
    got_s, got_m = [], []
    for s_dev, m_dev in hmc.sampleChunks(S, 4, 1 / kB, 1.0, chain0=7, iter0=2, spill_dir=str(tmp_path),
                                         momenta=True):
        got_s.append(s_dev.cpu().numpy().copy())
        got_m.append(m_dev.cpu().numpy().copy())
    assert [g.shape[2] for g in got_s] == [4, 4, 3]
    assert np.array_equal(np.concatenate(got_s, axis=2), ref_s)
    assert np.array_equal(np.concatenate(got_m, axis=2), ref_m)
    spilled = np.concatenate([np.load(tmp_path / f"samples_{k:05d}.npy") for k in range(3)], axis=2)
    assert np.array_equal(spilled, ref_s)
    assert 0.0 < hmc.acceptRate <= 1.0


@pytest.mark.parametrize("D,N,mass", [(32, 3000, False), (27, 500, True)])
def test_rosenbrock_kdk_fma_form(P, lib, D, N, mass):
    """PBBI_KDK_FMA (kick-drift-kick with fused multiply-adds, the two-lane kernel's throughput
    form): same integrator algebraically, so q, p agree with the oracle's velocity-Verlet to 1e-12
    (fp64 tolerance for this mode) and the accept masks are equal."""
    S, L, h, seed = 5, 10, 0.03, 5
    pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    m = (1.0 + (np.arange(N) % 3) * 0.5) if mass else None
    ens = P.Ensemble(D, N)
    if mass:
        ens.mass = m.copy()
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, rng="philox", seed=seed, kdk_fma=True,
                verbose=False)
    samples, momenta = hmc.getSamples(S, 1.0 / kB, 0.3)
    q = device_normal(lib, seed, lib.STREAM_POSITION, 0, 0, D, N, 0.3)
    pstd = np.sqrt(m) if mass else np.ones(N)
    n_rej = 0
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, 0, D, N, 1.0, pstd)
        u = device_uniform(lib, seed, i, 0, N)
        _, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, m, h, L)
        assert np.array_equal(hmc.reject_masks[i], rej)
        assert scaled_err(samples[:, :, i], q) <= 1e-12 and scaled_err(momenta[:, :, i], p) <= 1e-12
        q = samples[:, :, i].copy()  # continue from the device state: compare step by step
        n_rej += int(rej.sum())
    assert n_rej > 0
    # host-supplied momenta / uniforms through pbbi_hmc_iter with the flag (compat store of p)
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(D)
    p, u = rs.standard_normal((D, N)) * pstd, rs.uniform(size=N)
    u[::5] = 1.5
    qd, pd, ud = (as_device(x, 0, np.float64) for x in (q, p, u))
    md = as_device(m, 0, np.float64) if mass else None
    qo, po, rj = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0), empty((N,), np.uint8, 0)
    lib.call("pbbi_hmc_iter", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
             md.data_ptr() if mass else None, qo.data_ptr(), po.data_ptr(), None, rj.data_ptr(), N, N, h, L,
             lib.COMPAT_P_FROM_OLDQ | lib.KDK_FMA, stream_ptr(0))
    q_or, p_or = q.copy(), p.copy()
    _, rej = orc.hmc_iter(op, "Leapfrog", q_or, p_or, u, m, h, L)
    assert np.array_equal(to_numpy(rj).astype(bool), rej) and rej.sum() >= N // 5
    assert scaled_err(to_numpy(qo), q_or) <= 1e-12 and scaled_err(to_numpy(po), p_or) <= 1e-12


@pytest.mark.parametrize("kind,D,N,mass", [("diag", 32, 700, False), ("harmonic", 17, 100, True),
                                           ("diag", 64, 333, True), ("diag", 50, 64, False),
                                           ("harmonic", 128, 1000, False), ("diag", 100, 257, True),
                                           ("diag", 256, 130, False), ("harmonic", 241, 70, True)])
@pytest.mark.parametrize("method", ["Leapfrog", "Stormer-Verlet"])
def test_separable_multilane_kdk(P, lib, kind, D, N, mass, method):
    """kernels_sepn.hip: harmonic / diagonal Gaussian, a chain's 16-dim parts in different waves of
    one workgroup, PBBI_KDK_FMA form (both RNG modes): q, p within 1e-12 of the oracle's
    velocity-Verlet, masks equal."""
    rs = np.random.RandomState(D + N)
    pot, op = _stream_case(P, kind, D, rs)
    S, L, h, seed = 4, 7, 0.45, 21
    m = (1.0 + (np.arange(N) % 4) * 0.5) if mass else None
    ens = P.Ensemble(D, N)
    if mass:
        ens.mass = m.copy()
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, method=method, rng="philox", seed=seed,
                kdk_fma=True, compat=False, verbose=False)
    samples, momenta = hmc.getSamples(S, 1.0 / kB, 1.0, chain0=9)
    q = device_normal(lib, seed, lib.STREAM_POSITION, 0, 9, D, N, 1.0)
    pstd = np.sqrt(m) if mass else np.ones(N)
    n_rej = 0
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, 9, D, N, 1.0, pstd)
        u = device_uniform(lib, seed, i, 9, N)
        r_or, rej = orc.hmc_iter(op, method, q, p, u, m, h, L, compat=0)
        assert np.array_equal(hmc.reject_masks[i], rej)
        assert scaled_err(samples[:, :, i], q) <= 1e-12 and scaled_err(momenta[:, :, i], p) <= 1e-12
        assert np.max(np.abs(np.log(hmc.ratios[i]) - np.log(r_or))) < 1e-9
        q = samples[:, :, i].copy()
        n_rej += int(rej.sum())
    assert 0 < n_rej < S * N
    # host-supplied momenta / uniforms through pbbi_hmc_iter with the flag
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    p, u = rs.standard_normal((D, N)) * pstd, rs.uniform(size=N)
    qd, pd, ud = (as_device(x, 0, np.float64) for x in (q, p, u))
    md = as_device(m, 0, np.float64) if mass else None
    qo, po, rj = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0), empty((N,), np.uint8, 0)
    lib.call("pbbi_hmc_iter", pot.handle, orc.METHODS[method], qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
             md.data_ptr() if mass else None, qo.data_ptr(), po.data_ptr(), None, rj.data_ptr(), N, N, h, L,
             lib.COMPAT_P_FROM_OLDQ | lib.KDK_FMA, stream_ptr(0))
    q_or, p_or = q.copy(), p.copy()
    _, rej = orc.hmc_iter(op, method, q_or, p_or, u, m, h, L)
    assert np.array_equal(to_numpy(rj).astype(bool), rej)
    assert scaled_err(to_numpy(qo), q_or) <= 1e-12 and scaled_err(to_numpy(po), p_or) <= 1e-12


@pytest.mark.parametrize("D,N,mass", [(48, 500, False), (64, 333, True), (100, 130, False),
                                      (128, 1000, False), (256, 70, True), (33, 64, False),
                                      (144, 200, True), (200, 64, False)])
def test_rosenbrock_multiwave_kdk(P, lib, D, N, mass):
    _rosenbrock_kdk_case(P, lib, D, N, mass, "Leapfrog")


@pytest.mark.parametrize("D,N,mass", [(20, 100, False), (32, 300, True), (64, 200, False), (100, 70, True)])
def test_rosenbrock_kdk_stormer_verlet(P, lib, D, N, mass):
    """Stormer-Verlet under PBBI_KDK_FMA on kernels_rosg.hip (2 / 4 / 8 lanes per chain): the same
    recurrence without the closing half kick and with one more drift."""
    _rosenbrock_kdk_case(P, lib, D, N, mass, "Stormer-Verlet")


def _rosenbrock_kdk_case(P, lib, D, N, mass, method):
    """Rosenbrock at 32 < D <= 256 under PBBI_KDK_FMA: kernels_rosg.hip up to D = 128 (4 / 8 lanes of
    one wave per chain, in-wave boundary exchange) and kernels_rosn.hip above (a chain's 16-dim
    parts in different waves, boundary values through LDS): q, p within 1e-12 of the oracle's
    velocity-Verlet, accept masks equal, both RNG modes."""
    S, L, h, seed = 4, 8, 0.03, 31
    pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    m = (1.0 + (np.arange(N) % 3) * 0.5) if mass else None
    ens = P.Ensemble(D, N)
    if mass:
        ens.mass = m.copy()
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, method=method, rng="philox", seed=seed,
                kdk_fma=True, verbose=False)
    samples, momenta = hmc.getSamples(S, 1.0 / kB, 0.3, chain0=3)
    q = device_normal(lib, seed, lib.STREAM_POSITION, 0, 3, D, N, 0.3)
    pstd = np.sqrt(m) if mass else np.ones(N)
    n_rej = 0
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, 3, D, N, 1.0, pstd)
        u = device_uniform(lib, seed, i, 3, N)
        r_or, rej = orc.hmc_iter(op, method, q, p, u, m, h, L)
        assert np.array_equal(hmc.reject_masks[i], rej)
        assert scaled_err(samples[:, :, i], q) <= 1e-12 and scaled_err(momenta[:, :, i], p) <= 1e-12
        fin = np.isfinite(r_or) & (r_or > 0)
        assert np.max(np.abs(np.log(hmc.ratios[i][fin]) - np.log(r_or[fin]))) < 1e-9
        q = samples[:, :, i].copy()
        n_rej += int(rej.sum())
    assert 0 < n_rej < S * N
    # host-supplied momenta / uniforms, zero steps and one step
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(D)
    for Lx in (0, 1):
        p, u = rs.standard_normal((D, N)) * pstd, rs.uniform(size=N)
        qd, pd, ud = (as_device(x, 0, np.float64) for x in (q, p, u))
        md = as_device(m, 0, np.float64) if mass else None
        qo, po, rj = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0), empty((N,), np.uint8, 0)
        lib.call("pbbi_hmc_iter", pot.handle, orc.METHODS[method], qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
                 md.data_ptr() if mass else None, qo.data_ptr(), po.data_ptr(), None, rj.data_ptr(), N, N,
                 h, Lx, lib.KDK_FMA, stream_ptr(0))
        q_or, p_or = q.copy(), p.copy()
        _, rej = orc.hmc_iter(op, method, q_or, p_or, u, m, h, Lx, compat=0)
        assert np.array_equal(to_numpy(rj).astype(bool), rej)
        assert scaled_err(to_numpy(qo), q_or) <= 1e-12 and scaled_err(to_numpy(po), p_or) <= 1e-12


@pytest.mark.parametrize("case", ["lane_diag8", "lane2_ros32", "sepn_diag64", "rosn_ros64", "stream_diag100",
                                  "dense64", "big200", "custom_reg", "custom_ws"])
def test_hmc_iter_in_place_aliasing(P, lib, case):
    """include/pbbi.h: q_out may alias q_in and p_out may alias p_in.  Every kernel family gives the
    same result in place as out of place (rejections forced so the old state is re-read)."""
    import torch
    from custom_sources import QUARTIC
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(7)
    flags = lib.COMPAT_P_FROM_OLDQ
    if case == "lane_diag8":
        D, pot = 8, P.GaussianDiag(rs.standard_normal(8), prec=rs.uniform(0.5, 2, 8), const=0.0)
    elif case == "lane2_ros32":
        D, pot = 32, P.Rosenbrock(32)
    elif case == "sepn_diag64":
        D, pot, flags = 64, P.GaussianDiag(rs.standard_normal(64), prec=rs.uniform(0.5, 2, 64), const=0.0), flags | lib.KDK_FMA
    elif case == "rosn_ros64":
        D, pot, flags = 64, P.Rosenbrock(64), flags | lib.KDK_FMA
    elif case == "stream_diag100":
        D, pot = 100, P.GaussianDiag(rs.standard_normal(100), prec=rs.uniform(0.5, 2, 100), const=0.0)
    elif case in ("dense64", "big200"):
        D = 64 if case == "dense64" else 200
        A = rs.standard_normal((D, D))
        Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
        pot = P.GaussianDense(rs.standard_normal(D), precision=0.5 * (Pm + Pm.T), const=0.0)
    else:
        D = 9 if case == "custom_reg" else 20
        pot = CustomPotential(D, QUARTIC, [1.0, 0.5])
    N, L = 200, 5
    h = 0.02 if "ros" in case else 0.2
    q = rs.standard_normal((D, N)) * (0.3 if "ros" in case else 1.0)
    p, u = rs.standard_normal((D, N)), rs.uniform(size=N)
    u[::3] = 1.5
    m = as_device(1.0 + (np.arange(N) % 3) * 0.5, 0, np.float64)
    qd, pd, ud = (as_device(x, 0, np.float64) for x in (q, p, u))
    qo, po = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0)
    rj1, rj2 = empty((N,), np.uint8, 0), empty((N,), np.uint8, 0)
    args = (N, N, h, L, flags, stream_ptr(0))
    lib.call("pbbi_hmc_iter", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(), m.data_ptr(),
             qo.data_ptr(), po.data_ptr(), None, rj1.data_ptr(), *args)
    lib.call("pbbi_hmc_iter", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(), m.data_ptr(),
             qd.data_ptr(), pd.data_ptr(), None, rj2.data_ptr(), *args)
    torch.cuda.synchronize()
    assert np.array_equal(to_numpy(rj1), to_numpy(rj2)) and 0 < to_numpy(rj1).sum() < N
    assert np.array_equal(to_numpy(qo), to_numpy(qd)) and np.array_equal(to_numpy(po), to_numpy(pd))


def test_rhat_matches_numpy_and_detects_disagreement(P):
    """HMC.rhat (per-chain Welford moments + ensemble moments on the GPU) == the NumPy formula on the
    same draws; ~1 for a converged ensemble, large when half the chains sit elsewhere."""
    import torch
    D, N, S = 6, 512, 40
    pot = P.GaussianDiag(np.zeros(D), prec=np.ones(D), const=0.0)
    hmc = P.HMC(P.Ensemble(D, N), 1.3, 0.1, None, potential=pot, rng="philox", seed=3, verbose=False)
    s_dev, _ = hmc.getSamples(S, 1 / kB, 1.0, device_output=True)

    def rhat_np(x):                      # x: (D, N, S)
        m, v = x.mean(axis=2), x.var(axis=2, ddof=1)
        W, B_over_S = v.mean(axis=1), m.var(axis=1, ddof=1)
        return np.sqrt(((S - 1.0) / S * W + B_over_S) / W)
    r = hmc.rhat(s_dev)
    assert np.allclose(r, rhat_np(s_dev.cpu().numpy()), rtol=1e-10)
    assert np.all(np.abs(r - 1.0) < 0.05)
    shifted = s_dev.clone()
    shifted[0, : N // 2, :] += 5.0
    r2 = hmc.rhat(shifted)
    assert np.allclose(r2, rhat_np(shifted.cpu().numpy()), rtol=1e-10)
    assert r2[0] > 2.0 and np.all(np.abs(r2[1:] - 1.0) < 0.05)


@pytest.mark.parametrize("D,mass", [(11, False), (24, True), (40, False)])
def test_custom_potential_kdk_form(P, lib, D, mass):
    """User potential under PBBI_KDK_FMA (the in-kernel-draw default): D = 11 and 24 run the
    register-resident kick-drift-kick kernel (q, v, g on chip: twice the dimension of the
    reference-order variant), D = 40 the workspace kernels; 1e-12 against the oracle, masks equal."""
    from custom_sources import QUARTIC
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    N, S, L, h, seed = 400, 4, 9, 0.07, 13
    prm = [1.5, 0.75]
    pot, op = CustomPotential(D, QUARTIC, prm), orc.pot_custom(QUARTIC, D, prm)
    m = (1.0 + (np.arange(N) % 4) * 0.5) if mass else None
    ens = P.Ensemble(D, N)
    if mass:
        ens.mass = m.copy()
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, rng="philox", seed=seed, verbose=False)
    assert hmc.kdk_fma
    samples, momenta = hmc.getSamples(S, 1.0 / kB, 1.0)
    q = device_normal(lib, seed, lib.STREAM_POSITION, 0, 0, D, N, 1.0)
    pstd = np.sqrt(m) if mass else np.ones(N)
    n_rej = 0
    for i in range(S):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, 0, D, N, 1.0, pstd)
        u = device_uniform(lib, seed, i, 0, N)
        _, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, m, h, L)
        assert np.array_equal(hmc.reject_masks[i], rej)
        assert scaled_err(samples[:, :, i], q) <= 1e-12 and scaled_err(momenta[:, :, i], p) <= 1e-12
        q = samples[:, :, i].copy()
        n_rej += int(rej.sum())
    assert 0 < n_rej < S * N


@pytest.mark.parametrize("case", ["lane_diag8", "sepn_diag64", "lane2_ros32", "rosg_ros64", "rosn_ros160",
                                  "dense16", "big200", "stream_diag300", "custom9"])
def test_momentum_scale_follows_temperature(P, lib, case):
    """In-kernel momentum draws are sqrt(mass * kB * T) * z in every kernel family (src/ensemble.py:88):
    with zero leapfrog steps the stored momentum is the draw itself (unit mass: exactly)."""
    from custom_sources import QUARTIC
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    rs = np.random.RandomState(3)
    diag = lambda D: P.GaussianDiag(rs.standard_normal(D), prec=rs.uniform(0.5, 2, D), const=0.0)
    if case == "lane_diag8":
        D, pot = 8, diag(8)
    elif case == "sepn_diag64":
        D, pot = 64, diag(64)
    elif case == "stream_diag300":
        D, pot = 300, diag(300)
    elif case == "lane2_ros32":
        D, pot = 32, P.Rosenbrock(32)
    elif case == "rosg_ros64":
        D, pot = 64, P.Rosenbrock(64)
    elif case == "rosn_ros160":
        D, pot = 160, P.Rosenbrock(160)
    elif case in ("dense16", "big200"):
        D = 16 if case == "dense16" else 200
        A = rs.standard_normal((D, D))
        Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
        pot = P.GaussianDense(None, precision=0.5 * (Pm + Pm.T), const=0.0)
    else:
        D, pot = 9, CustomPotential(9, QUARTIC, [1.0, 0.5])
    N, T, seed = 130, 2.5 / kB, 17
    hmc = P.HMC(P.Ensemble(D, N), 0.05, 0.1, None, potential=pot, rng="philox", seed=seed, verbose=False)
    assert hmc.integrator.numSteps == 0
    samples, momenta = hmc.getSamples(2, T, 0.4, chain0=11)
    q0 = device_normal(lib, seed, lib.STREAM_POSITION, 0, 11, D, N, 0.4)
    for i in range(2):
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, 11, D, N, float(np.sqrt(2.5)))
        assert np.array_equal(momenta[:, :, i], p)
        tol = 1e-13 if case in ("sepn_diag64",) else 0.0   # x = q - mu, q = x + mu round trip
        assert np.max(np.abs(samples[:, :, i] - q0)) <= tol
    assert not hmc.reject_masks.any() and np.all(hmc.ratios == 1.0)


def test_trajectory_length_jitter_removes_resonance(P):
    """A fixed trajectory length T leaves the modes with omega*T near k*pi unmixed (here: one
    eigen-direction with omega*T = pi exactly: q -> -q every iteration); jitter of the step count
    per iteration mixes it.  Bayesian-linear-regression-like posterior, chains started at 0."""
    D, N, S = 4, 4096, 60
    prec = np.array([1.0, 4.0, (np.pi / 0.5) ** 2, 9.0])          # omega_2 * T = pi for T = 0.5
    mu = np.array([1.0, -2.0, 3.0, 0.5])
    pot = P.GaussianDiag(mu, prec=prec, const=0.0)

    def run(jitter):
        hmc = P.HMC(P.Ensemble(D, N), 0.5, 0.01, None, potential=pot, rng="philox", seed=4, verbose=False)
        s_dev, _ = hmc.getSamples(S, 1 / kB, 0.05, device_output=True, jitter=jitter)
        return hmc.sampleMoments(s_dev[:, :, 20:])
    mean0, var0 = run(0.0)
    mean1, var1 = run(0.3)
    # without jitter the resonant coordinate never moves away from +-(q0 - mu): its spread over draws
    # stays far from the posterior variance; with jitter mean and variance are recovered
    assert abs(var0[2] * prec[2] - 1.0) > 0.5
    assert np.max(np.abs(mean1 - mu)) < 0.02
    assert np.max(np.abs(var1 * prec - 1.0)) < 0.06


@pytest.mark.parametrize("kind", ["dense", "ros32", "diag64", "big200", "custom"])
def test_burn_in_equals_discarding_draws(P, lib, kind):
    """pbbi_hmc_run with samples_out = NULL (burn-in: the state ping-pongs between two scratch slabs,
    nothing is recorded): getSamples(S, burn_in=B) is bit-identical to the last S draws of
    getSamples(B + S), in every kernel family."""
    from custom_sources import QUARTIC
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    rs = np.random.RandomState(2)
    if kind in ("dense", "big200"):
        D = 24 if kind == "dense" else 200
        A = rs.standard_normal((D, D))
        Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
        pot = P.GaussianDense(rs.standard_normal(D), precision=0.5 * (Pm + Pm.T), const=0.0)
    elif kind == "ros32":
        D, pot = 32, P.Rosenbrock(32)
    elif kind == "diag64":
        D, pot = 64, P.GaussianDiag(rs.standard_normal(64), prec=rs.uniform(0.5, 2, 64), const=0.0)
    else:
        D, pot = 12, CustomPotential(12, QUARTIC, [1.0, 0.5])
    N, B, S = 257, 3, 4
    h = 0.02 if kind == "ros32" else 0.1
    kw = dict(potential=pot, rng="philox", seed=6, verbose=False)
    full, _ = P.HMC(P.Ensemble(D, N), 10 * h + 1e-9, h, None, **kw).getSamples(B + S, 1 / kB, 0.5, chain0=5)
    hmc = P.HMC(P.Ensemble(D, N), 10 * h + 1e-9, h, None, **kw)
    part, _ = hmc.getSamples(S, 1 / kB, 0.5, chain0=5, burn_in=B)
    assert np.array_equal(part, full[:, :, B:])
    with pytest.raises(lib.PbbiError):   # a momentum slab without a sample slab
        from physicsbasedbayesianinference_amd._device import empty, stream_ptr
        q = empty((D, N), np.float64, 0)
        m = empty((1, D, N), np.float64, 0)
        lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, None, m.data_ptr(), None, None, N, N,
                 h, 3, 1, 0, 1, 0, 0, 1.0, stream_ptr(0))


# ------------------------------------------------------------------ SURVEY 8a rows a5 / a9
@pytest.mark.parametrize("kind", ["harmonic", "dense", "rosenbrock"])
def test_integrator_getaccel_with_masses(P, kind):
    """Integrator.getAccel(i) = -gradient(q[:, i]) / mass[i]  (src/integrator.py:61-73): sign and
    the division by the chain's own mass, against the oracle's gradient."""
    rs = np.random.RandomState(3)
    if kind == "harmonic":
        D, k = 3, np.array([2.0, 3.0, 0.5])
        pot, op = P.Harmonic(k), orc.pot_harmonic(k)
    elif kind == "dense":
        D = 8
        A = rs.standard_normal((D, D))
        Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
        Pm, mu = 0.5 * (Pm + Pm.T), rs.standard_normal(D)
        pot, op = P.GaussianDense(mu, precision=Pm, const=0.0), orc.pot_gauss_dense(mu, Pm)
    else:
        D = 5
        pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    N = 7
    ens = P.Ensemble(D, N)
    ens.q = rs.standard_normal((D, N))
    ens.mass = 1.0 + 0.75 * np.arange(N)           # all different, none equal to 1 except chain 0
    integ = P.Leapfrog(ens, 0.1, 1.0, pot.gradient)
    _, g = orc.potential(op, ens.q, want_grad=True)
    for i in range(N):
        a = integ.getAccel(i)
        assert a.shape == (D,)
        ref = -g[:, i] / ens.mass[i]
        if kind == "dense":
            assert scaled_err(a, ref) <= RTOL_DENSE
        else:
            assert np.array_equal(a, ref)
    # one chain, checked by hand: harmonic a = -k*q/m
    if kind == "harmonic":
        assert np.allclose(integ.getAccel(2), -k * ens.q[:, 2] / 2.5, rtol=0, atol=1e-15)


def test_hmc_from_density_only(P):
    """HMC(ens, T, h, density) with potential=None (src/HMC.py:52-56): the potential is
    potentialFunc = -log(density) (src/HMC.py:75-84) and sampling reproduces config C1 (G3)."""
    g = load_golden("G3_getsamples_c1")
    D, N, S = int(g["D"]), int(g["N"]), int(g["S"])
    pot = P.StandardGaussian(D)
    np.random.seed(int(g["seed"]))
    ens = P.Ensemble(D, N)
    hmc = P.HMC(ens, float(g["simulTime"]), float(g["stepSize"]), pot.density, verbose=False)
    assert hmc.potential == hmc.potentialFunc and hmc.density == pot.density
    q = np.array([[0.3, -1.2, 2.0]])
    U = hmc.potentialFunc(q)                                    # -log(exp(-U)) on the HIP eval kernel
    assert U.shape == (3,)
    assert np.max(np.abs(U - orc.potential(orc.pot_gauss_diag(np.zeros(D), np.ones(D)), q))) < 1e-14
    assert abs(hmc.potentialFunc(np.array([0.5])) - 0.125) < 1e-15
    samples, momenta = hmc.getSamples(S, float(g["temperature"]), float(g["qStd"]))
    assert np.array_equal(hmc.reject_masks, g["reject_mask"])
    assert scaled_err(samples, g["samples"]) <= RTOL_GOLDEN
    assert scaled_err(momenta, g["momenta"]) <= RTOL_GOLDEN


# ------------------------------------------------------------------ beta in the accept test, ensemble weights
@pytest.mark.parametrize("case", ["lane_diag5", "dense24", "ros32", "diag64_kdk", "big200", "stream_ros70"])
@pytest.mark.parametrize("rng", ["upload", "philox"])
def test_beta_accept_vs_oracle(P, lib, case, rng):
    """PBBI_BETA_ACCEPT: ratio = exp((oldH - newH) / kT) in every kernel family, against the oracle's
    hmc_iter(beta=1/kT) on the same draws; without the flag the reference's exp(oldH - newH)
    (src/HMC.py:115) whatever kT is."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(11)
    kdk = case.endswith("_kdk")
    if case == "lane_diag5":
        D, mu, prec = 5, rs.standard_normal(5), rs.uniform(0.5, 2, 5)
        pot, op, h = P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec), 0.4
    elif case in ("dense24", "big200"):
        D = 24 if case == "dense24" else 200
        A = rs.standard_normal((D, D))
        Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
        Pm = 0.5 * (Pm + Pm.T)
        pot, op, h = P.GaussianDense(None, precision=Pm, const=0.0), orc.pot_gauss_dense(np.zeros(D), Pm), 0.4
    elif case == "ros32":
        D, pot, op, h = 32, P.Rosenbrock(32), orc.pot_rosenbrock(32), 0.13
    elif case == "stream_ros70":
        D, pot, op, h = 70, P.Rosenbrock(70), orc.pot_rosenbrock(70), 0.12
    else:
        D, mu, prec = 64, rs.standard_normal(64), rs.uniform(0.5, 2, 64)
        pot, op, h = P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec), 0.3
    N, L, kT, seed = 300, 6, 2.5, 3
    tol = 0.0 if case in ("lane_diag5", "ros32", "stream_ros70") else 1e-11
    st = stream_ptr(0)
    flags = lib.COMPAT_P_FROM_OLDQ | lib.BETA_ACCEPT | (lib.KDK_FMA if kdk else 0)
    q0 = (1.0 if "ros" in case else 0.0) + 0.5 * rs.standard_normal((D, N))
    qo, po = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0)
    ro, rj = empty((N,), np.float64, 0), empty((N,), np.uint8, 0)
    qd = as_device(q0, 0, np.float64)
    if rng == "upload":
        p = rs.standard_normal((D, N)) * np.sqrt(kT)
        u = rs.uniform(size=N)
        pd, ud = as_device(p, 0, np.float64), as_device(u, 0, np.float64)
        lib.call("pbbi_hmc_iter_kt", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(), None,
                 qo.data_ptr(), po.data_ptr(), ro.data_ptr(), rj.data_ptr(), N, N, h, L, flags, kT, st)
    else:
        p = device_normal(lib, seed, lib.STREAM_MOMENTUM, 0, 0, D, N, np.sqrt(kT))
        u = device_uniform(lib, seed, 0, 0, N)
        lib.call("pbbi_hmc_run", pot.handle, 0, qd.data_ptr(), None, qo.data_ptr(), po.data_ptr(), rj.data_ptr(),
                 ro.data_ptr(), N, N, h, L, 1, flags, seed, 0, 0, kT, st)
    torch.cuda.synchronize()
    q_or, p_or = q0.copy(), p.copy()
    r_or, rej_or = orc.hmc_iter(op, "Leapfrog", q_or, p_or, u, None, h, L, beta=1.0 / kT)
    r_ref, rej_ref = orc.hmc_iter(op, "Leapfrog", q0.copy(), p.copy(), u, None, h, L)      # reference test
    assert np.array_equal(to_numpy(rj).astype(bool), rej_or)
    assert rej_or.any() and not np.array_equal(r_or, r_ref, equal_nan=True)
    if "ros" not in case:  # (Rosenbrock trajectories are either near-exact or diverged: the same decisions)
        assert not np.array_equal(rej_or, rej_ref)                # the flag changes decisions at kT != 1
    if tol == 0.0:
        assert np.array_equal(to_numpy(qo), q_or) and np.array_equal(to_numpy(po), p_or)
    else:
        assert scaled_err(to_numpy(qo), q_or) <= tol and scaled_err(to_numpy(po), p_or) <= tol
    fin = np.isfinite(r_or) & (r_or > 0)
    assert np.max(np.abs(np.log(to_numpy(ro)[fin]) - np.log(r_or[fin]))) < 1e-8


def test_tempered_sampling_moments(P):
    """HMC(..., beta_accept=True) at T = 2.5/kB samples exp(-U/kT): for a Gaussian potential with
    covariance Sigma the draws have covariance kT*Sigma; the reference's test (beta = 1 while p is drawn
    at kT, src/HMC.py:115 vs src/ensemble.py:88) does not."""
    D, N, kT = 6, 8192, 2.5
    rs = np.random.RandomState(5)
    A = rs.standard_normal((D, D))
    cov = A @ A.T / D + 0.5 * np.eye(D)
    mu = rs.standard_normal(D)
    pot = P.GaussianDense(mu, cov=cov)
    out = {}
    for beta_accept in (True, False):
        hmc = P.HMC(P.Ensemble(D, N), 1.8, 0.6, None, potential=pot, rng="philox", seed=2, verbose=False,
                    beta_accept=beta_accept)   # a coarse step: 8-18 % of the proposals are rejected
        assert hmc.integrator.numSteps == 3
        s, _ = hmc.getSamples(40, kT / kB, 1.0, device_output=True)
        x = s[:, :, 20:].permute(0, 2, 1).reshape(D, -1).double()
        out[beta_accept] = (x.mean(1).cpu().numpy(), torch_cov(x))
    mean, c = out[True]
    assert np.max(np.abs(mean - mu)) < 0.03
    assert np.max(np.abs(c - kT * cov)) < 0.02 * kT * np.max(np.abs(cov))
    # the reference's accept test at this temperature: visibly not the canonical ensemble at kT
    # (oracle, same draws: 0.5 % against 7 % off)
    assert np.max(np.abs(out[False][1] - kT * cov)) > 0.04 * kT * np.max(np.abs(cov))


@pytest.mark.parametrize("D,method", [(256, "Leapfrog"), (200, "Leapfrog"), (160, "Leapfrog")])
def test_dense_stream_samples_the_target(P, D, method):
    """The streamed dense kernel (128 < D <= 256, kernels_dstream.hip) as a sampler: 16 384 chains from N(0, 1),
    150 (400) iterations of a carried, fused run with in-kernel draws -- the ensemble's mean and covariance
    are the Gaussian's, to Monte-Carlo error (an end-to-end check no replay against the oracle gives: a wrong row
    of P, a stale carried gradient or a misplaced momentum row would bias the covariance).  Leapfrog only: the
    reference's Stormer-Verlet returns a velocity half a step behind its position (src/integrator.py:142-163), its
    proposal is not the reversible map the accept test assumes, and its chains settle 4 % below the target's
    variance here -- on the oracle as on the kernels, which the replay tests compare."""
    import torch
    N = 16384
    rs = np.random.RandomState(D)
    A = rs.standard_normal((D, D))
    cov = A @ A.T / D + 0.5 * np.eye(D)
    mu = rs.standard_normal(D)
    pot = P.GaussianDense(mu, cov=cov)
    cls = {"Leapfrog": "Leapfrog", "Stormer-Verlet": "Stormer-Verlet"}[method]
    h = 0.15 if method == "Leapfrog" else 0.03
    hmc = P.HMC(P.Ensemble(D, N), 10.5 * h, h, None, method=cls, potential=pot, rng="philox", seed=3, verbose=False)
    assert hmc.integrator.numSteps == 10
    s, _ = hmc.getSamples(150 if method == "Leapfrog" else 400, 1 / kB, 1.0, device_output=True)
    assert "streamed P" in hmc.describeRun(150)
    x = s[:, :, -1].double()                               # (D, N): the ensemble after the last iteration
    mean, c = x.mean(1).cpu().numpy(), torch_cov(x)
    scale = np.max(np.abs(cov))
    assert 0.3 < hmc.acceptRate < 1.0
    assert np.max(np.abs(mean - mu)) < 6.0 * np.sqrt(scale / N)
    assert np.max(np.abs(c - cov)) < 0.06 * scale          # sqrt(2 / N) = 1.1 % per entry, 256^2 entries


def torch_cov(x):
    xc = x - x.mean(1, keepdim=True)
    return (xc @ xc.T / (x.shape[1] - 1)).cpu().numpy()


def test_ensemble_weights_on_device(P, lib):
    """pbbi_reduce_min / pbbi_canonical_weights / pbbi_scale_inverse and HMC.ensembleWeights against
    NumPy: w = exp(-beta (H - Hmin)) / sum, NaN energies skipped by the minimum, fp32 handles too."""
    import torch
    from physicsbasedbayesianinference_amd import distributed
    rs = np.random.RandomState(1)
    for N in (1, 257, 70001):
        H = rs.standard_normal(N) * 30 + 1000.0
        for dtype in (torch.float64, torch.float32):
            Hd = torch.tensor(H, dtype=dtype, device="cuda")
            w, logz = distributed.ensemble_weights(Hd, beta=0.7)
            Hh = Hd.cpu().double().numpy()
            e = np.exp(-0.7 * (Hh - Hh.min()))
            assert w.dtype == dtype
            assert np.allclose(w.cpu().double().numpy(), e / e.sum(), rtol=1e-12 if dtype == torch.float64 else 2e-6)
            assert abs(logz - (np.log(e.sum()) - 0.7 * Hh.min())) < 1e-9 * abs(logz) + 1e-6
    D, N = 3, 500
    pot = P.Harmonic(np.array([2.0, 3.0, 0.5]))
    ens = P.Ensemble(D, N)
    ens.mass = 1.0 + np.arange(N) % 3
    hmc = P.HMC(ens, 1.0, 0.1, None, potential=pot, verbose=False)
    q, p = rs.standard_normal((D, N)), rs.standard_normal((D, N))
    w = hmc.ensembleWeights(q, p, temperature=2.0 / kB)
    _, Hh = orc.weights(orc.pot_harmonic(np.array([2.0, 3.0, 0.5])), q, p, ens.mass)
    e = np.exp(-(Hh - Hh.min()) / 2.0)
    assert np.allclose(w, e / e.sum(), rtol=1e-12) and ens.weights is w and abs(w.sum() - 1) < 1e-12
    # unnormalised weights are what getWeights returns (src/HMC.py:103)
    assert np.allclose(hmc.getWeights(q, p), np.exp(-Hh), rtol=1e-12)


# ------------------------------------------------------------------ automatic gradient
@pytest.mark.parametrize("name,D,rtol", [("quartic", 11, 1e-12), ("quartic", 40, 1e-12), ("logistic", 5, 1e-10),
                                         ("coin", 2, 1e-10)])
def test_autodiff_gradient_matches_handwritten(P, lib, name, D, rtol):
    """A source WITHOUT `gradient` (the reference's default path, jax.grad(potential), src/HMC.py:57-60)
    is differentiated by dual numbers inside the kernels (csrc/pbbi_autodiff.h): its gradient equals the
    hand-written one (1e-12 on polynomials), on the register-resident (D <= 16) and the workspace
    kernels, and sampling with it reproduces sampling with the hand-written gradient."""
    from custom_sources import (COIN_TOSS_AD, COIN_TOSS_SOURCE, LOGISTIC, LOGISTIC_AD, QUARTIC, QUARTIC_AD,
                                logistic_problem)
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    rs = np.random.RandomState(D)
    if name == "quartic":
        hand, auto, prm = QUARTIC, QUARTIC_AD, [1.5, 0.5]
    elif name == "logistic":
        hand, auto = LOGISTIC, LOGISTIC_AD
        prm = logistic_problem(M=40, D=D)[3]
    else:
        hand, auto, prm = COIN_TOSS_SOURCE, COIN_TOSS_AD, [63.0, 39.0, 12.0, 30.0]
    ph, pa = CustomPotential(D, hand, prm), CustomPotential(D, auto, prm)
    assert pa.autodiff and not ph.autodiff
    q = rs.standard_normal((D, 300))
    Uh, gh = ph.value_and_gradient(q)
    Ua, ga = pa.value_and_gradient(q)
    assert np.allclose(Ua, Uh, rtol=1e-13, atol=1e-13)
    assert np.max(np.abs(ga - gh)) <= rtol * max(1.0, np.max(np.abs(gh)))
    assert pa.check_gradient(q[:, :8]) < 1e-6            # and both agree with central differences
    # one HMC run each: same draws, same decisions, states within the gradient's rounding
    out = []
    for pot in (ph, pa):
        hmc = P.HMC(P.Ensemble(D, 256), 0.5, 0.05, None, potential=pot, rng="philox", seed=4, verbose=False,
                    kdk_fma=False)
        s, m = hmc.getSamples(3, 1.0 / kB, 0.5)
        out.append((s, m, hmc.reject_masks.copy()))
    assert np.array_equal(out[0][2], out[1][2])
    assert scaled_err(out[1][0], out[0][0]) <= 1e-10 and scaled_err(out[1][1], out[0][1]) <= 1e-10


def test_coin_toss_sampled_without_a_gradient(P):
    """The reference's two-coin example end to end with potential only (its NumPyro model gives
    log_density and jax.grad gives the gradient, CoinTossExample.py:75-107): Beta posteriors recovered."""
    from custom_sources import COIN_TOSS_AD
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    k, n = np.array([62.0, 11.0]), np.array([100.0, 40.0])
    pot = CustomPotential(2, COIN_TOSS_AD, np.stack([k + 1, n - k + 1], axis=1).ravel())
    hmc = P.HMC(P.Ensemble(2, 8192), 1.0, 0.1, None, potential=pot, rng="philox", seed=3, verbose=False)
    s, _ = hmc.getSamples(40, 1.0 / kB, 1.0, jitter=0.5)   # omega*T is near pi for coin 2 at T = 1
    theta = 1.0 / (1.0 + np.exp(-s[:, :, 10:]))
    for j in range(2):
        A, B = k[j] + 1, n[j] - k[j] + 1
        assert abs(theta[j].mean() - A / (A + B)) < 3e-3
        assert abs(theta[j].var() / (A * B / ((A + B) ** 2 * (A + B + 1))) - 1.0) < 0.06


# ------------------------------------------------------------------ covariance and ESS in the sink
def test_sample_covariance_and_ess_match_numpy(P, lib):
    """pbbi_sample_covariance and pbbi_chain_autocov / HMC.ess against NumPy on AR(1) chains with known
    autocorrelation phi: ESS ~ N*S*(1-phi)/(1+phi); D not a multiple of 16, fp32 slabs too."""
    import torch
    rs = np.random.RandomState(0)
    D, N, S = 19, 700, 120
    phi = np.linspace(-0.3, 0.8, D)
    x = np.empty((S, D, N))
    x[0] = rs.standard_normal((D, N))
    for s in range(1, S):
        x[s] = phi[:, None] * x[s - 1] + np.sqrt(1 - phi[:, None] ** 2) * rs.standard_normal((D, N))
    x[:, 3] += 0.7 * x[:, 5]                                  # some cross-covariance
    x += 5.0 * np.arange(D)[None, :, None]                    # large means: the shift matters
    hmc = P.HMC(P.Ensemble(D, N), 1.0, 0.1, None, potential=P.StandardGaussian(D), verbose=False)
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 2e-5)):
        xd = torch.tensor(x, dtype=dtype, device="cuda")
        dns = xd.permute(1, 2, 0)
        mean, cov = hmc.sampleCovariance(dns)
        flat = xd.double().cpu().numpy().transpose(1, 0, 2).reshape(D, -1)
        assert np.allclose(mean, flat.mean(1), rtol=1e-6 if dtype == torch.float32 else 1e-12)
        ref = np.cov(flat, bias=True) if dtype == torch.float64 else np.cov(flat - mean[:, None] + flat.mean(1)[:, None], bias=True)
        assert np.max(np.abs(cov - np.cov(flat, bias=True))) < tol * max(1.0, np.max(np.abs(ref))) + (2e-4 if dtype == torch.float32 else 0)
        assert np.array_equal(cov, cov.T)
    # autocovariance kernel against NumPy, lag by lag
    xd = torch.tensor(x, dtype=torch.float64, device="cuda")
    from physicsbasedbayesianinference_amd._device import stream_ptr
    cm = xd.mean(0).contiguous()
    T = 20
    acov = torch.empty((T + 1, D), dtype=torch.float64, device="cuda")
    lib.call("pbbi_chain_autocov", xd.data_ptr(), cm.data_ptr(), S, D, N, T, lib.F64, 0, acov.data_ptr(), stream_ptr(0))
    torch.cuda.synchronize()
    xc = x - x.mean(0, keepdims=True)
    for t in (0, 1, 7, T):
        ref = (xc[:S - t] * xc[t:]).sum(0).mean(1) / S
        assert np.allclose(acov[t].cpu().numpy(), ref, rtol=1e-11, atol=1e-13)
    ess = hmc.ess(xd.permute(1, 2, 0))
    expect = N * S * (1 - phi) / (1 + phi)
    ok = np.ones(D, dtype=bool)
    ok[3] = False                                             # the mixed dimension has another spectrum
    assert np.all(np.abs(ess[ok] / expect[ok] - 1.0) < 0.15), ess / expect
    with pytest.raises(lib.PbbiError):
        lib.call("pbbi_chain_autocov", xd.data_ptr(), cm.data_ptr(), S, D, N, 33, lib.F64, 0, acov.data_ptr(), stream_ptr(0))


# ------------------------------------------------------------------ per-chain trajectory lengths
def _dyn_case(P, case, rs):
    if case == "diag5":
        D, mu, prec = 5, rs.standard_normal(5), rs.uniform(0.5, 2, 5)
        return D, P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec), 0.1
    if case == "diag30":
        D, mu, prec = 30, rs.standard_normal(30), rs.uniform(0.5, 2, 30)
        return D, P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec), 0.1
    if case == "harm3":
        k = np.array([2.0, 3.0, 0.5])
        return 3, P.Harmonic(k), orc.pot_harmonic(k), 0.1
    if case == "ros12":
        return 12, P.Rosenbrock(12), orc.pot_rosenbrock(12), 0.03
    if case == "ros32":
        return 32, P.Rosenbrock(32), orc.pot_rosenbrock(32), 0.03
    if case == "ros20":
        return 20, P.Rosenbrock(20), orc.pot_rosenbrock(20), 0.03
    D = int(case[5:])
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    Pm, mu = 0.5 * (Pm + Pm.T), rs.standard_normal(D)
    return D, P.GaussianDense(mu, precision=Pm, const=0.0), orc.pot_gauss_dense(mu, Pm), 0.1


@pytest.mark.parametrize("case,mass", [("diag5", False), ("diag30", True), ("harm3", True), ("ros12", False),
                                       ("ros32", True), ("ros32", False), ("ros20", True)])  # ros20/32: two-lane kernel
@pytest.mark.parametrize("mode", ["steps", "uturn", "both"])
def test_per_chain_steps_lane_kernels_bitexact(P, lib, case, mass, mode):
    """pbbi_hmc_iter_dyn on the chain-per-lane kernels (k_lane_dyn_hmc; Rosenbrock with 16 < D <= 32 on the
    two-lane kernel's DYN instantiation): per-chain step counts (uploaded) and / or the U-turn stop,
    against the oracle's leapfrog_chain_dyn -- step counts, decisions, q and p bit for bit."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(len(case) * 7 + len(mode))
    D, pot, op, h = _dyn_case(P, case, rs)
    N, L = 777, 40
    q0 = (1.0 if "ros" in case else 0.0) + 0.5 * rs.standard_normal((D, N))
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    p0 = rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0)
    u = rs.uniform(size=N)
    steps_in = rs.randint(0, L + 3, size=N).astype(np.int32) if mode != "uturn" else None  # some beyond L: clamped
    flags = lib.COMPAT_P_FROM_OLDQ | (lib.PER_CHAIN_STEPS if mode != "uturn" else 0) | \
        (lib.UTURN_STOP if mode != "steps" else 0)
    qd, pd, ud = (as_device(x, 0, np.float64) for x in (q0, p0, u))
    md = as_device(m, 0, np.float64) if mass else None
    sd = torch.tensor(steps_in, device="cuda") if steps_in is not None else None
    qo, po = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0)
    ro, rj, so = empty((N,), np.float64, 0), empty((N,), np.uint8, 0), empty((N,), np.int32, 0)
    lib.call("pbbi_hmc_iter_dyn", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
             md.data_ptr() if mass else None, sd.data_ptr() if sd is not None else None, qo.data_ptr(),
             po.data_ptr(), ro.data_ptr(), rj.data_ptr(), so.data_ptr(), N, N, h, L, flags, 1.0, stream_ptr(0))
    torch.cuda.synchronize()
    q_or, p_or = q0.copy(), p0.copy()
    r_or, rej_or, st_or = orc.hmc_iter_dyn(op, q_or, p_or, u, m, h, L, steps_in=steps_in, uturn=(mode != "steps"))
    assert np.array_equal(to_numpy(so), st_or)
    assert np.array_equal(to_numpy(rj).astype(bool), rej_or)
    assert np.array_equal(to_numpy(qo), q_or) and np.array_equal(to_numpy(po), p_or)
    if mode != "steps":
        assert len(np.unique(st_or)) > 3                          # genuinely divergent lengths inside a wave
    if mode == "uturn":
        assert st_or.min() >= 1
    if mode == "steps":
        assert np.array_equal(st_or, np.clip(steps_in, 0, L))


@pytest.mark.parametrize("case,mass", [("dense24", False), ("dense100", True), ("dense128", False),
                                       ("dense160", True), ("dense256", False), ("dense200", True)])
@pytest.mark.parametrize("rng", ["upload", "philox"])
def test_per_chain_steps_dense_kernel(P, lib, case, mass, rng):
    """PBBI_PER_CHAIN_STEPS on the dense MFMA kernel (finished chains frozen by per-lane coefficients, the
    tile runs its longest chain): counts, decisions and states against the oracle.  D > 128: the streamed-P
    kernel, whose four waves step together (a workgroup-wide vote per step keeps the ring's mat-vec sequence one)."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(5)
    D, pot, op, h = _dyn_case(P, case, rs)
    N, L, seed = 333, 9, 21
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    flags = lib.COMPAT_P_FROM_OLDQ | lib.PER_CHAIN_STEPS
    q0 = rs.standard_normal((D, N))
    qd = as_device(q0, 0, np.float64)
    qo, po = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0)
    ro, rj, so = empty((N,), np.float64, 0), empty((N,), np.uint8, 0), empty((N,), np.int32, 0)
    if rng == "upload":
        p0 = rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0)
        u = rs.uniform(size=N)
        steps_in = rs.randint(0, L + 1, size=N).astype(np.int32)   # [0, L] as include/pbbi.h says: 0 = no move
        steps_in[-16:] = 0                                          # a whole 16-chain tile that takes no step
        steps_in[:3] = (0, L, 0)
        pd, ud, sd = as_device(p0, 0, np.float64), as_device(u, 0, np.float64), torch.tensor(steps_in, device="cuda")
        lib.call("pbbi_hmc_iter_dyn", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
                 md.data_ptr() if mass else None, sd.data_ptr(), qo.data_ptr(), po.data_ptr(), ro.data_ptr(),
                 rj.data_ptr(), so.data_ptr(), N, N, h, L, flags, 1.0, stream_ptr(0))
    else:
        p0 = device_normal(lib, seed, lib.STREAM_MOMENTUM, 0, 0, D, N, 1.0, np.sqrt(m) if mass else None)
        u = device_uniform(lib, seed, 0, 0, N)
        steps_in = orc.philox_steps(seed, 0, 0, N, L)
        lib.call("pbbi_hmc_run_dyn", pot.handle, 0, qd.data_ptr(), md.data_ptr() if mass else None, qo.data_ptr(),
                 po.data_ptr(), rj.data_ptr(), ro.data_ptr(), so.data_ptr(), N, N, h, L, 1, flags, seed, 0, 0, 1.0,
                 stream_ptr(0))
        dsteps = empty((N,), np.int32, 0)
        lib.call("pbbi_philox_steps", seed, 0, 0, N, L, 0, dsteps.data_ptr(), stream_ptr(0))
        assert np.array_equal(to_numpy(dsteps), steps_in)         # integer draw: device == oracle
    torch.cuda.synchronize()
    q_or, p_or = q0.copy(), p0.copy()
    r_or, rej_or, st_or = orc.hmc_iter_dyn(op, q_or, p_or, u, m, h, L, steps_in=steps_in)
    assert np.array_equal(to_numpy(so), st_or) and len(np.unique(st_or)) == L + (1 if rng == "upload" else 0)
    assert np.array_equal(to_numpy(rj).astype(bool), rej_or)
    assert scaled_err(to_numpy(qo), q_or) <= RTOL_DENSE and scaled_err(to_numpy(po), p_or) <= RTOL_DENSE
    if rng == "upload":   # chains with no step: the position they started from, bit for bit; never rejected
        still = steps_in == 0
        assert still.sum() >= 18 and np.array_equal(to_numpy(qo)[:, still], q0[:, still])
        assert not to_numpy(rj).astype(bool)[still].any()


@pytest.mark.parametrize("case,mass", [("dense24", False), ("dense100", True), ("dense128", False), ("dense128", True), ("dense80", True),
                                       ("dense192", False), ("dense256", True)])
@pytest.mark.parametrize("mode", ["uturn", "both"])
def test_uturn_stop_dense_kernel(P, lib, case, mass, mode):
    """PBBI_UTURN_STOP on the dense MFMA kernel: a chain stops at the first step where (q - q_0) . v < 0 (the
    tile keeps stepping for its other chains; the chain's last full kick is brought back to a half kick at the
    next executed step).  Step counts, decisions and states against the oracle's leapfrog_chain_dyn; the test
    quantity is formed from kick-drift-kick values that differ from the oracle's in the last bits, so a chain
    whose dot product passes zero within rounding may stop one step apart: such chains (none or very few) are
    left out of the state comparison."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(8)
    D, pot, op, h = _dyn_case(P, case, rs)
    N, L = 333, 40
    h = 0.15
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    flags = lib.COMPAT_P_FROM_OLDQ | lib.UTURN_STOP | (lib.PER_CHAIN_STEPS if mode == "both" else 0)
    q0 = rs.standard_normal((D, N))
    p0 = rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0)
    u = rs.uniform(size=N)
    steps_in = rs.randint(1, L + 1, size=N).astype(np.int32) if mode == "both" else None
    qd, pd, ud = as_device(q0, 0, np.float64), as_device(p0, 0, np.float64), as_device(u, 0, np.float64)
    sd = torch.tensor(steps_in, device="cuda") if mode == "both" else None
    qo, po = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0)
    ro, rj, so = empty((N,), np.float64, 0), empty((N,), np.uint8, 0), empty((N,), np.int32, 0)
    lib.call("pbbi_hmc_iter_dyn", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
             md.data_ptr() if mass else None, sd.data_ptr() if mode == "both" else None, qo.data_ptr(),
             po.data_ptr(), ro.data_ptr(), rj.data_ptr(), so.data_ptr(), N, N, h, L, flags, 1.0, stream_ptr(0))
    torch.cuda.synchronize()
    q_or, p_or = q0.copy(), p0.copy()
    r_or, rej_or, st_or = orc.hmc_iter_dyn(op, q_or, p_or, u, m, h, L, steps_in=steps_in, uturn=True)
    st = to_numpy(so)
    same = st == st_or
    assert same.mean() > 0.99, (same.mean(), st[:20], st_or[:20])
    cap = steps_in if mode == "both" else np.full(N, L)
    assert (st_or < cap).mean() > (0.4 if mode == "uturn" else 0.2) and len(np.unique(st_or)) > 5   # U-turns did end trajectories
    assert np.array_equal(to_numpy(rj).astype(bool)[same], rej_or[same])
    assert scaled_err(to_numpy(qo)[:, same], q_or[:, same]) <= RTOL_DENSE
    assert scaled_err(to_numpy(po)[:, same], p_or[:, same]) <= RTOL_DENSE


def test_per_chain_steps_sampling_and_uturn_adaptation(P):
    """getSamples(per_chain_steps=True) is a valid sampler (lengths are drawn independently of the state):
    it recovers mean and variance of a Gaussian INCLUDING the coordinate a fixed length leaves unmixed
    (omega*T = 2 pi), and adaptTrajectoryLength sets a length near the ensemble's U-turn time (about a
    quarter to half a period of the slowest mode)."""
    D, N = 4, 16384
    prec = np.array([1.0, 4.0, (2 * np.pi) ** 2 / 4.0, 0.25])   # dimension 2: omega * T = 2 pi at T = 2
    mu = np.array([0.5, -1.0, 2.0, 0.0])
    pot = P.GaussianDiag(mu, prec=prec, const=0.0)
    hmc = P.HMC(P.Ensemble(D, N), 2.0, 0.05, None, potential=pot, rng="philox", seed=1, verbose=False, kdk_fma=False)
    s, _ = hmc.getSamples(60, 1.0 / kB, 1.0, per_chain_steps=True)
    assert hmc.steps.shape == (60, N) and hmc.steps.min() == 1 and hmc.steps.max() == 40
    x = s[:, :, 20:].reshape(D, -1)
    assert np.max(np.abs(x.mean(1) - mu)) < 0.02
    assert np.max(np.abs(x.var(1) * prec - 1.0)) < 0.05
    T = hmc.adaptTrajectoryLength(1.0 / kB, 1.0, iterations=12, max_steps=400)
    assert hmc.integrator.numSteps == int(round(T / 0.05)) and hmc.uturn_steps.shape == (6, N)
    # the same measurement on the dense MFMA kernel: an isotropic Gaussian of unit frequency turns after
    # about a quarter to half a period (pi/2 .. pi)
    Dd, Nd = 24, 4096
    potd = P.GaussianDense(None, precision=np.eye(Dd), const=0.0)
    hd = P.HMC(P.Ensemble(Dd, Nd), 1.0, 0.05, None, potential=potd, rng="philox", seed=2, verbose=False)
    Td = hd.adaptTrajectoryLength(1.0 / kB, 1.0, iterations=8, max_steps=200)
    assert 1.2 < Td < 3.5 and hd.uturn_steps.max() < 200
    # slowest mode omega = 0.5: its U-turn comes after ~ a quarter period (pi) when started at the mode's
    # edge, later otherwise; faster modes pull the multivariate criterion earlier
    assert 0.5 < T < 2 * np.pi, T


# ------------------------------------------------------------------ reference-order separable kernel, 16 < D <= 256
@pytest.mark.parametrize("kind,D,N,mass,compat", [("diag", 17, 100, False, True), ("diag", 32, 700, True, False),
                                                  ("harmonic", 48, 333, True, True), ("diag", 64, 130, False, True),
                                                  ("diag", 100, 257, True, True), ("harmonic", 128, 64, False, False),
                                                  ("diag", 250, 70, True, True), ("diag", 256, 129, False, True)])
@pytest.mark.parametrize("rng", ["upload", "philox"])
@pytest.mark.parametrize("method", ["Leapfrog", "Stormer-Verlet"])
def test_separable_multiwave_reference_order_bitexact(P, lib, kind, D, N, mass, compat, rng, method):
    """k_sep_exact_hmc (kernels_sepn.hip): 16-dim parts of a chain in the waves of a workgroup, the
    reference's operation order, energy sums continued from part to part -> q, p, ratio's decision
    BIT-EXACT with the oracle for harmonic / diagonal Gaussian potentials up to D = 256, the path the
    drop-in's default (kdk_fma=False) takes; forced rejections, masses, both momentum-restore conventions."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(D + N)
    if kind == "diag":
        mu, prec = rs.standard_normal(D), rs.uniform(0.5, 2.0, D)
        pot, op = P.GaussianDiag(mu, prec=prec, const=0.3), orc.pot_gauss_diag(mu, prec, 0.3)
    else:
        k = rs.uniform(0.5, 2.0, D)
        pot, op = P.Harmonic(k), orc.pot_harmonic(k)
    h, L, S, seed, chain0 = 0.3, 7, 3, 5, 1000
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    flags = lib.COMPAT_P_FROM_OLDQ if compat else 0
    st = stream_ptr(0)
    q = np.ascontiguousarray(rs.standard_normal((D, N)))
    n_rej = 0
    if rng == "upload":
        for it in range(S):
            p = rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0)
            u = rs.uniform(size=N)
            u[::4] = 1.5
            qo, po, ratio, rej = gpu_hmc_iter(lib, pot, method, q, p, u, m, h, L, compat=compat)
            q_or, p_or = q.copy(), p.copy()
            r_or, rej_or = orc.hmc_iter(op, method, q_or, p_or, u, m, h, L, compat=flags)
            assert np.array_equal(rej, rej_or)
            assert np.array_equal(qo, q_or) and np.array_equal(po, p_or)
            fin = np.isfinite(r_or) & (r_or > 0)
            assert np.max(np.abs(np.log(ratio[fin]) - np.log(r_or[fin]))) < 1e-9
            n_rej += int(rej.sum())
            q = qo
        assert n_rej >= S * (N // 4)
    else:
        qd = as_device(q, 0, np.float64)
        samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
        reject = empty((S, N), np.uint8, 0)
        lib.call("pbbi_hmc_run", pot.handle, orc.METHODS[method], qd.data_ptr(), md.data_ptr() if mass else None,
                 samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, S, flags, seed, 0,
                 chain0, 1.0, st)
        torch.cuda.synchronize()
        pstd = np.sqrt(m) if mass else None
        for i in range(S):
            p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, chain0, D, N, 1.0, pstd)
            u = device_uniform(lib, seed, i, chain0, N)
            _, rej = orc.hmc_iter(op, method, q, p, u, m, h, L, compat=flags)
            assert np.array_equal(to_numpy(reject[i]).astype(bool), rej)
            assert np.array_equal(to_numpy(samples[i]), q) and np.array_equal(to_numpy(momenta[i]), p)


@pytest.mark.parametrize("D,N,mass,compat", [(33, 130, False, True), (48, 500, True, False), (64, 333, False, True),
                                             (100, 130, True, True), (128, 64, False, False), (128, 257, True, True)])
@pytest.mark.parametrize("rng", ["upload", "philox"])
def test_rosenbrock_multilane_reference_order_bitexact(P, lib, D, N, mass, compat, rng):
    """k_rosg_exact_hmc (kernels_rosg.hip): 4 / 8 lanes of one wave per chain, the reference's operation
    order, energy sums passed from part to part in dimension order -> bit-exact with the oracle for
    Rosenbrock at 32 < D <= 128 (the drop-in's default path there); rejections present."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(D * 7 + N)
    pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    h, L, S, seed, chain0 = 0.05, 10, 3, 9, 4321
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    flags = lib.COMPAT_P_FROM_OLDQ if compat else 0
    q = np.ascontiguousarray(1.0 + 0.3 * rs.standard_normal((D, N)))
    n_rej = 0
    if rng == "upload":
        for it in range(S):
            p = rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0)
            u = rs.uniform(size=N)
            u[::5] = 1.5
            qo, po, ratio, rej = gpu_hmc_iter(lib, pot, "Leapfrog", q, p, u, m, h, L, compat=compat)
            q_or, p_or = q.copy(), p.copy()
            r_or, rej_or = orc.hmc_iter(op, "Leapfrog", q_or, p_or, u, m, h, L, compat=flags)
            assert np.array_equal(rej, rej_or)
            assert np.array_equal(qo, q_or) and np.array_equal(po, p_or)
            fin = np.isfinite(r_or) & (r_or > 0)
            assert np.max(np.abs(np.log(ratio[fin]) - np.log(r_or[fin]))) < 1e-9
            n_rej += int(rej.sum())
            q = qo
        assert n_rej >= S * (N // 5)
    else:
        md = as_device(m, 0, np.float64) if mass else None
        qd = as_device(q, 0, np.float64)
        samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
        reject = empty((S, N), np.uint8, 0)
        lib.call("pbbi_hmc_run", pot.handle, 0, qd.data_ptr(), md.data_ptr() if mass else None, samples.data_ptr(),
                 momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, S, flags, seed, 0, chain0, 1.0, stream_ptr(0))
        torch.cuda.synchronize()
        pstd = np.sqrt(m) if mass else None
        for i in range(S):
            p = device_normal(lib, seed, lib.STREAM_MOMENTUM, i, chain0, D, N, 1.0, pstd)
            u = device_uniform(lib, seed, i, chain0, N)
            _, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, m, h, L, compat=flags)
            assert np.array_equal(to_numpy(reject[i]).astype(bool), rej)
            assert np.array_equal(to_numpy(samples[i]), q) and np.array_equal(to_numpy(momenta[i]), p)


def test_rccl_collectives_run_on_hardware_single_rank():
    """The collectives of the sharded path on RCCL itself.  This pool hands out one GPU, so the
    world is one rank: `init_process_group("nccl")` on cuda:0, then the all-gather of sample slabs
    (`distributed.gather_samples`) and the two all-reduces of `distributed.ensemble_weights` on HIP tensors,
    in a child process (the process group must not leak into this one).  World size > 1 is covered on
    gloo (tests/test_host_logic.py) and by the driver's multi-GPU bench."""
    import socket
    import subprocess
    import sys
    import textwrap
    from conftest import ROOT
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        from physicsbasedbayesianinference_amd import distributed as D
        torch.cuda.set_device(0)
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        except Exception as e:
            print("RCCL-INIT-FAILED", repr(e)); sys.exit(0)
        x = torch.arange(2 * 3 * 5, dtype=torch.float64, device="cuda").reshape(2, 3, 5)
        y = D.gather_samples(x, _force_collective=True)
        assert y.shape == x.shape and torch.equal(y, x)
        H = torch.tensor([0.5, 1.5, 0.25, 3.0], dtype=torch.float64, device="cuda")
        t = H.min().clone(); dist.all_reduce(t, op=dist.ReduceOp.MIN); assert float(t) == 0.25
        s = H.sum().clone(); dist.all_reduce(s, op=dist.ReduceOp.SUM); assert float(s) == 5.25
        w, logz = D.ensemble_weights(H, beta=2.0)
        ref = np.exp(-2.0 * (H.cpu().numpy() - 0.25)); ref /= ref.sum()
        assert np.allclose(w.cpu().numpy(), ref, rtol=1e-13)
        dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
        print("RCCL-OK", dist.is_nccl_available())
    """ % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    if "RCCL-INIT-FAILED" in out:
        pytest.skip("RCCL could not initialise on this box: " + out[-300:])
    assert r.returncode == 0 and "RCCL-OK" in out, out[-2000:]


# ---------------------------------------------------------------------------------------------
# Python callables as potentials (trace.py): the reference's lambdas (src/HMC.py:52-60,
# src/tests/test_integrator_harmonic.py:22-24, src/tests/test_HMC.py:27-33,48-49,124-125) traced into
# descriptors / generated kernels -- no C++ string, no CPU path.  Golden fixtures through them.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("route", ["descriptor", "source"])
@pytest.mark.parametrize("name,method", [("G1_leapfrog_harmonic", "Leapfrog"),
                                         ("G2_stormerverlet_harmonic", "Stormer-Verlet")])
def test_traced_harmonic_lambda_reproduces_the_integrator_fixtures(P, name, method, route):
    """harmonicPotential = lambda q: harmonicPotentialND(q, springConsts); harmonicGradient = grad(...)
    (src/tests/test_integrator_harmonic.py:22-24), handed to Leapfrog / StormerVerlet like the reference
    does.  route "descriptor": the trace is recognised as Harmonic(k) (bit-exact with the oracle);
    "source": the same trace pushed through the generated-source kernels."""
    from physicsbasedbayesianinference_amd.trace import grad, trace_potential
    g = load_golden(name)
    springConsts = g["springConsts"]
    harmonicPotential = lambda q: P.harmonicPotentialND(q, springConsts)   # noqa: E731
    harmonicGradient = grad(harmonicPotential)
    cls = P.Leapfrog if method == "Leapfrog" else P.StormerVerlet
    for s in ("_h0.1", "_h0.01"):
        D, N = g["q0" + s].shape
        ens = P.Ensemble(D, N)
        ens.mass = g["mass" + s].copy()
        ens.q[...] = g["q0" + s]
        ens.p[...] = g["p0" + s]
        gradient = harmonicGradient if route == "descriptor" else \
            trace_potential(harmonicPotential, D=D, prefer="source")
        integ = cls(ens, float(g["stepSize" + s]), float(g["finalTime" + s]), gradient)
        assert integ.potential.kind == ("harmonic" if route == "descriptor" else "custom")
        q, p = integ.integrate()
        assert q is ens.q and p is ens.p
        assert scaled_err(q, g["q" + s]) <= RTOL_GOLDEN and scaled_err(p, g["p" + s]) <= RTOL_GOLDEN
        assert scaled_err(integ.v, g["v" + s]) <= RTOL_GOLDEN
        if route == "descriptor":
            qo, po = np.ascontiguousarray(g["q0" + s]), np.ascontiguousarray(g["p0" + s])
            orc.integrate(orc.pot_harmonic(springConsts), method, qo, po, g["mass" + s],
                          float(g["stepSize" + s]), int(g["numSteps" + s]))
            assert np.array_equal(q, qo) and np.array_equal(p, po)
    # grad(f) on numbers: the traced descriptor's gradient kernel
    assert np.array_equal(harmonicGradient(np.array([3.0, 4.0])), springConsts * np.array([3.0, 4.0]))


@pytest.mark.parametrize("route", ["descriptor", "source"])
@pytest.mark.parametrize("name", ["G4_getsamples_dense_d8", "G11_getsamples_test2", "G4b_getsamples_dense_mean_d16"])
def test_traced_gaussian_lambda_reproduces_getsamples_fixtures(P, name, route):
    """potential = lambda q: <closed-form -logpdf on the traceable namespace> through HMC.getSamples on the
    reference's seed: the fixture's samples, momenta and reject masks.  "descriptor": the quadratic form is
    recognised (GaussianDense: MFMA kernel); "source": generated-source kernels."""
    from physicsbasedbayesianinference_amd import trace as jnp
    g = load_golden(name)
    D, N, S = int(g["D"]), int(g["N"]), int(g["S"])
    mean, Pm, const = g["mean"], g["precision"], float(g["const"])
    potentialFunc = lambda q: 0.5 * jnp.dot(q - mean, jnp.dot(Pm, q - mean)) + const   # noqa: E731
    potential = potentialFunc if route == "descriptor" else jnp.trace_potential(potentialFunc, D=D, prefer="source")
    np.random.seed(int(g["seed"]))
    ens = P.Ensemble(D, N)
    ens.mass = g["mass"].copy()
    hmc = P.HMC(ens, float(g["simulTime"]), float(g["stepSize"]), None, potential=potential,
                method=str(g["method"]), verbose=False)
    assert hmc._pot.kind == ("gauss_dense" if route == "descriptor" else "custom")
    samples, momenta = hmc.getSamples(S, float(g["temperature"]), float(g["qStd"]))
    assert np.array_equal(hmc.reject_masks, g["reject_mask"])
    assert scaled_err(samples, g["samples"]) <= RTOL_DENSE and scaled_err(momenta, g["momenta"]) <= RTOL_DENSE


def test_traced_multivariate_normal_and_density_only(P):
    """src/tests/test_HMC.py:124-130 as written (densityFunc / potentialFunc on multivariate_normal, T = 300 K)
    against fixture G11 (recorded from the reference with the same mean and covariance), and the density-only
    construction of src/tests/test_HMC.py:27-33 against fixture G3 (config C1)."""
    from physicsbasedbayesianinference_amd import trace as jnp
    from physicsbasedbayesianinference_amd.trace import multivariate_normal
    g = load_golden("G11_getsamples_test2")
    mean = jnp.ones(2) * 5
    cov = jnp.array([[4, -3], [-3, 4]])
    densityFunc = lambda q: multivariate_normal.pdf(q, mean, cov=cov)          # noqa: E731
    potentialFunc = lambda q: -multivariate_normal.logpdf(q, mean, cov=cov)    # noqa: E731
    np.random.seed(int(g["seed"]))
    ens = P.Ensemble(2, int(g["N"]))
    hmc = P.HMC(ens, float(g["simulTime"]), float(g["stepSize"]), densityFunc, potential=potentialFunc, verbose=False)
    samples, momenta = hmc.getSamples(int(g["S"]), float(g["temperature"]), float(g["qStd"]))
    assert hmc._pot.kind == "gauss_dense"
    assert np.array_equal(hmc.reject_masks, g["reject_mask"])
    assert scaled_err(samples, g["samples"]) <= RTOL_DENSE and scaled_err(momenta, g["momenta"]) <= RTOL_DENSE
    # density only: potential = -log(density)  (src/HMC.py:75-84)
    g = load_golden("G3_getsamples_c1")
    density = lambda x: jnp.exp(-0.50 * jnp.linalg.norm(x) ** 2) / jnp.sqrt(2 * jnp.pi)   # noqa: E731
    np.random.seed(int(g["seed"]))
    ens = P.Ensemble(1, int(g["N"]))
    hmc = P.HMC(ens, float(g["simulTime"]), float(g["stepSize"]), density, verbose=False)
    samples, momenta = hmc.getSamples(int(g["S"]), float(g["temperature"]), float(g["qStd"]))
    assert np.array_equal(hmc.reject_masks, g["reject_mask"])
    assert scaled_err(samples, g["samples"]) <= RTOL_GOLDEN and scaled_err(momenta, g["momenta"]) <= RTOL_GOLDEN
    assert abs(hmc.potential(np.array([0.3])) - (0.5 * 0.09 + 0.5 * np.log(2 * np.pi))) < 1e-15


def test_traced_nonquadratic_callable_vs_oracle(P, lib):
    """A callable with no closed-form descriptor (logistic regression written on the namespace): generated
    source + symbolic gradient in the plugin kernels against the oracle running the SAME generated source on
    the host -- decisions equal, states within 1e-10 -- and against the hand-written C++ of
    custom.LOGISTIC_REGRESSION_SOURCE (values within 1e-12)."""
    from physicsbasedbayesianinference_amd import trace as jnp
    from physicsbasedbayesianinference_amd.custom import complete_source, logistic_regression_posterior
    rs = np.random.RandomState(4)
    M, D, N = 24, 5, 300
    X = rs.standard_normal((M, D))
    y = (rs.uniform(size=M) < 0.5).astype(np.float64)

    def softplus(z):
        return jnp.maximum(z, 0.0) + jnp.log1p(jnp.exp(-jnp.abs(z)))

    fn = lambda w: jnp.sum(softplus(X @ w) - y * (X @ w)) + 0.5 * jnp.dot(w, w)   # noqa: E731
    pot = jnp.trace_potential(fn, D=D)
    assert pot.kind == "custom"
    q = rs.standard_normal((D, N))
    U, gr = pot.value_and_gradient(q)
    hand = logistic_regression_posterior(X, y, 1.0)
    Uh, gh = hand.value_and_gradient(q)
    assert scaled_err(U, Uh) <= 1e-12 and scaled_err(gr, gh) <= 1e-12
    op = orc.pot_custom(complete_source(pot.traced_source), D, pot.params)
    h, L = 0.2, 6
    for it in range(3):
        p, u = rs.standard_normal((D, N)), rs.uniform(size=N)
        qo, po, ratio, rej = gpu_hmc_iter(lib, pot, "Leapfrog", q, p, u, None, h, L)
        q_or, p_or = q.copy(), p.copy()
        _, rej_or = orc.hmc_iter(op, "Leapfrog", q_or, p_or, u, None, h, L)
        assert np.array_equal(rej, rej_or)
        assert scaled_err(qo, q_or) <= 1e-10 and scaled_err(po, p_or) <= 1e-10
        q = qo


def test_untraceable_callable_raises_typeerror_in_the_class_api(P):
    import math
    ens = P.Ensemble(2, 4)
    with pytest.raises(TypeError):
        P.HMC(ens, 1.0, 0.1, None, potential=lambda q: math.exp(q[0]), verbose=False)
    with pytest.raises(TypeError):
        P.Leapfrog(ens, 0.1, 1.0, lambda q: np.tanh(q))


def test_reference_script_runs_through_dropin_with_its_own_imports(tmp_path):
    """The reference's harmonic test set-up (src/tests/test_integrator_harmonic.py:13-24,41-81) with ITS import
    lines -- `from ensemble import Ensemble`, `from integrator import Leapfrog`, `from potential import
    harmonicPotentialND`, `from jax import grad` -- resolved by dropin/ (dropin/jax is the traceable namespace):
    fixture G1 to its tolerance."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = load_golden("G1_leapfrog_harmonic")
    inp, outp = tmp_path / "in.npz", tmp_path / "out.npz"
    np.savez(inp, q0=g["q0_h0.1"], p0=g["p0_h0.1"], mass=g["mass_h0.1"], k=g["springConsts"],
             h=g["stepSize_h0.1"], T=g["finalTime_h0.1"])
    script = f"""
import sys
sys.path[:0] = [{root!r}, {os.path.join(root, "dropin")!r}]
from ensemble import Ensemble
from integrator import Leapfrog, StormerVerlet
from potential import harmonicPotentialND
from jax import grad
import numpy as np
z = np.load({str(inp)!r})
springConsts = z["k"]
harmonicPotential = lambda q: harmonicPotentialND(q, springConsts)
harmonicGradient = grad(harmonicPotential)
ensemble1 = Ensemble(2, z["q0"].shape[1])
ensemble1.mass = z["mass"].copy()
ensemble1.q[...] = z["q0"]
ensemble1.p[...] = z["p0"]
sol_q_p = Leapfrog(ensemble1, float(z["h"]), float(z["T"]), harmonicGradient)
q_num, p_num = sol_q_p.integrate()
np.savez({str(outp)!r}, q=q_num, p=p_num)
"""
    res = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    out = np.load(outp)
    assert scaled_err(out["q"], g["q_h0.1"]) <= RTOL_GOLDEN and scaled_err(out["p"], g["p_h0.1"]) <= RTOL_GOLDEN


# ---------------------------------------------------------------------------------------------
# PBBI_DRAW_F64: the double-precision momentum draw of the RNG contract (include/pbbi.h) -- the
# counterpart of the reference's float64 normals (src/ensemble.py:72-74,88-91).  Built from
# +, -, *, /, sqrt, fma in a fixed order, so the oracle's restatement gives THE SAME BITS and a
# whole pbbi_hmc_run can be compared with a host-only oracle run (no device draws replayed).
# ---------------------------------------------------------------------------------------------
def ulp_distance(a, b):
    ia, ib = np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64)
    ia = np.where(ia < 0, np.int64(-2 ** 63) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2 ** 63) - ib, ib)
    return np.abs(ia - ib)


def test_f64_draw_device_equals_oracle_mirror(lib):
    """pbbi_philox_normal with PBBI_STREAM_DRAW_F64 against oracle_philox_normal with the same bit: at most
    2 ulp apart (the bar); in fact the same bits, scaling included."""
    D, N = 40, 20000
    for seed, it, chain0, scale, per_chain in ((77, 3, 123456789012, 1.5, False), (2 ** 63 + 9, 0, 0, 1.0, True)):
        sn = 0.5 + (np.arange(N) % 7) * 0.25 if per_chain else None
        z = device_normal(lib, seed, lib.STREAM_MOMENTUM | lib.STREAM_DRAW_F64, it, chain0, D, N, scale, sn)
        zo = orc.philox_normal(seed, orc.STREAM_MOMENTUM | orc.STREAM_DRAW_F64, it, chain0, D, N,
                               sn if per_chain else scale)
        d = ulp_distance(z, zo)
        assert d.max() <= 2, d.max()
        assert np.array_equal(z, zo), f"{(d > 0).mean():.2e} of the variates differ by <= {d.max()} ulp"
    # position stream, float32 output: the double variate rounded once
    from physicsbasedbayesianinference_amd._device import empty, stream_ptr, to_numpy
    out = empty((D, 500), np.float32, 0)
    lib.call("pbbi_philox_normal", 5, lib.STREAM_POSITION | lib.STREAM_DRAW_F64, 1, 7, D, 500, 500, 2.0, None,
             lib.F32, 0, out.data_ptr(), stream_ptr(0))
    zo = orc.philox_normal(5, orc.STREAM_POSITION | orc.STREAM_DRAW_F64, 1, 7, D, 500, 2.0)
    assert np.array_equal(to_numpy(out), zo.astype(np.float32))


def test_f64_draw_distribution(lib):
    """N(0,1): moments, KS on 2.6e6 variates (all of them, the grid is 2^-52), independence inside and across
    blocks, no ties, sharding by chain offset."""
    from scipy import stats
    z = device_normal(lib, 5, lib.STREAM_MOMENTUM | lib.STREAM_DRAW_F64, 0, 0, 128, 20000)
    assert abs(z.mean()) < 3e-3 and abs(z.std() - 1) < 2e-3
    assert abs(stats.skew(z.ravel())) < 0.01 and abs(stats.kurtosis(z.ravel())) < 0.02
    assert stats.kstest(z.ravel(), "norm").pvalue > 1e-3
    assert np.abs(z).max() > 4.5
    c = np.corrcoef(z[:32])
    assert np.max(np.abs(c - np.eye(32))) < 0.04
    assert np.unique(z).size == z.size
    zs = device_normal(lib, 5, lib.STREAM_MOMENTUM | lib.STREAM_DRAW_F64, 0, 3000, 128, 500)
    assert np.array_equal(zs, z[:, 3000:3500])
    assert not np.array_equal(z[:, :500], device_normal(lib, 5, lib.STREAM_MOMENTUM, 0, 0, 128, 500))


F64_RUNS = [
    # name, kind, D, N, flags (beyond compat | DRAW_F64), dtype, tolerance (None = bit-exact), mass
    ("lane_diag8", "diag", 8, 300, 0, "float64", None, True),
    ("lane2_ros32_exact", "ros", 32, 200, 0, "float64", None, False),
    ("lane2_ros32_kdk", "ros", 32, 200, "kdk", "float64", 1e-12, False),
    ("lane2_ros24_kdk_mass", "ros", 24, 130, "kdk", "float64", 1e-12, True),
    ("sepx_diag64_exact", "diag", 64, 200, 0, "float64", None, True),
    ("sepn_diag64_kdk", "diag", 64, 200, "kdk", "float64", 1e-12, False),
    ("rosgx_ros64_exact", "ros", 64, 130, 0, "float64", None, False),
    ("rosg_ros64_kdk", "ros", 64, 130, "kdk", "float64", 1e-12, True),
    ("rosn_ros200_kdk", "ros", 200, 70, "kdk", "float64", 1e-12, False),
    ("stream_ros300", "ros", 300, 70, 0, "float64", None, True),
    ("stream_diag12_f32", "diag", 12, 200, 0, "float32", 3e-5, False),
    ("dense128_fused", "dense", 128, 300, 0, "float64", 1e-11, False),
    ("dense128_mass", "dense", 128, 150, 0, "float64", 1e-11, True),
    ("dense100", "dense", 100, 150, 0, "float64", 1e-11, False),
    ("dense96", "dense", 96, 150, 0, "float64", 1e-11, False),
    ("dense80_mass", "dense", 80, 150, 0, "float64", 1e-11, True),
    ("dense24", "dense", 24, 70, 0, "float64", 1e-11, True),
    ("gemm_dense200", "dense", 200, 150, 0, "float64", 1e-10, False),
    ("gemm_dense256_f32", "dense", 256, 128, 0, "float32", 2e-4, False),
    ("custom_quartic9", "quartic", 9, 300, 0, "float64", None, True),
    ("custom_quartic48", "quartic", 48, 100, 0, "float64", None, False),
]


@pytest.mark.parametrize("name,kind,D,N,extra,dtype,tol,mass", F64_RUNS, ids=[r[0] for r in F64_RUNS])
def test_hmc_run_with_f64_draws_vs_host_only_oracle(P, lib, name, kind, D, N, extra, dtype, tol, mass):
    """pbbi_hmc_run with PBBI_DRAW_F64 in every kernel family against oracle_hmc_run_philox with the same flag,
    which draws ITS OWN momenta (the oracle's C restatement of the draw): decisions equal, states bit-exact
    where the kernel keeps the reference's operation order, within the family's tolerance elsewhere."""
    import torch
    from custom_sources import QUARTIC
    from physicsbasedbayesianinference_amd.custom import CustomPotential
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(D)
    S, L, h = 5, 6, 0.1
    if kind == "diag":
        mu, prec = rs.standard_normal(D), rs.uniform(0.5, 2, D)
        pot, op = P.GaussianDiag(mu, prec=prec, const=0.0, dtype=dtype), orc.pot_gauss_diag(mu, prec)
    elif kind == "ros":
        pot, op, h = P.Rosenbrock(D, dtype=dtype), orc.pot_rosenbrock(D), 0.01
    elif kind == "dense":
        A = rs.standard_normal((D, D))
        Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
        Pm = 0.5 * (Pm + Pm.T)
        pot, op = P.GaussianDense(None, precision=Pm, const=0.0, dtype=dtype), orc.pot_gauss_dense(np.zeros(D), Pm)
    else:
        pot, op, h = CustomPotential(D, QUARTIC, [1.0, 0.5]), orc.pot_custom(QUARTIC, D, [1.0, 0.5]), 0.2
    npdt = np.float32 if dtype == "float32" else np.float64
    m = (1.0 + (np.arange(N) % 3) * 0.5) if mass else None
    md = as_device(m, 0, npdt) if mass else None
    seed, iter0, chain0, kT = 99, 3, 2 ** 32 + 11, 1.0
    flags = lib.COMPAT_P_FROM_OLDQ | lib.DRAW_F64 | (lib.KDK_FMA if extra == "kdk" else 0)
    q0 = orc.philox_normal(seed, orc.STREAM_POSITION | orc.STREAM_DRAW_F64, iter0, chain0, D, N,
                           0.1 if kind == "ros" else 1.0) + (1.0 if kind == "ros" else 0.0)
    if dtype == "float32":
        q0 = q0.astype(np.float32).astype(np.float64)
    qd = as_device(q0, 0, npdt)
    samples, momenta = empty((S, D, N), npdt, 0), empty((S, D, N), npdt, 0)
    reject = empty((S, N), np.uint8, 0)
    lib.call("pbbi_hmc_run", pot.handle, lib.LEAPFROG, qd.data_ptr(), md.data_ptr() if mass else None,
             samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(), None, N, N, h, L, S, flags, seed, iter0,
             chain0, kT, stream_ptr(0))
    torch.cuda.synchronize()
    gs, gm, gr = (to_numpy(x) for x in (samples, momenta, reject))
    q = np.ascontiguousarray(q0)
    if dtype == "float64":
        os_, om, orj, _ = orc.hmc_run_philox(op, "Leapfrog", q, m, h, L, S, seed, iter0, chain0, kT,
                                            compat=orc.COMPAT_P_FROM_OLDQ | orc.DRAW_F64)
        assert np.array_equal(gr.astype(bool), orj)
        if tol is None:
            assert np.array_equal(gs, os_) and np.array_equal(gm, om)
        else:
            assert scaled_err(gs, os_) <= tol and scaled_err(gm, om) <= tol
    else:  # float32 kernels against the fp64 oracle restarted from the kernel's own state every iteration
        pstd = np.sqrt((m if mass else np.ones(N)) * kT)
        for i in range(S):
            p = (orc.philox_normal(seed, orc.STREAM_MOMENTUM | orc.STREAM_DRAW_F64, iter0 + i, chain0, D, N, pstd)
                 ).astype(np.float32).astype(np.float64)
            u = orc.philox_uniform(seed, iter0 + i, chain0, N)
            p = np.ascontiguousarray(p)
            ratio, rej = orc.hmc_iter(op, "Leapfrog", q, p, u, m, h, L)
            with np.errstate(divide="ignore"):
                clear = np.abs(np.log(u) - np.minimum(0.0, np.log(ratio))) > 1e-2
            assert np.array_equal(gr[i].astype(bool)[clear], rej[clear])
            same = gr[i].astype(bool) == rej
            assert scaled_err(gs[i][:, same], q[:, same]) <= tol and scaled_err(gm[i][:, same], p[:, same]) <= tol
            q = np.ascontiguousarray(gs[i].astype(np.float64))
    assert 0.0 <= gr.mean() < 0.9


def test_class_api_draw_f64(P, lib):
    """HMC(..., rng="philox", draw_f64=True): positions and momenta from the double-precision draw; equal to a
    host-only oracle run; the default (single precision) gives a different, equally valid run."""
    D, N, S, L, h, seed = 6, 500, 4, 5, 0.2, 21
    rs = np.random.RandomState(0)
    mu, prec = rs.standard_normal(D), rs.uniform(0.5, 2, D)
    pot, op = P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec)
    ens = P.Ensemble(D, N)
    hmc = P.HMC(ens, L * h + 1e-9, h, None, potential=pot, rng="philox", seed=seed, kdk_fma=False, draw_f64=True,
                verbose=False)
    s, mom = hmc.getSamples(S, 1.0 / kB, 0.7)
    q = np.ascontiguousarray(orc.philox_normal(seed, orc.STREAM_POSITION | orc.STREAM_DRAW_F64, 0, 0, D, N, 0.7))
    os_, om, orj, _ = orc.hmc_run_philox(op, "Leapfrog", q, None, h, L, S, seed, 0, 0, 1.0,
                                        compat=orc.COMPAT_P_FROM_OLDQ | orc.DRAW_F64)
    assert np.array_equal(hmc.reject_masks, orj)
    assert np.array_equal(s, os_.transpose(1, 2, 0)) and np.array_equal(mom, om.transpose(1, 2, 0))
    hmc32 = P.HMC(P.Ensemble(D, N), L * h + 1e-9, h, None, potential=pot, rng="philox", seed=seed, kdk_fma=False,
                  verbose=False)
    s32, _ = hmc32.getSamples(S, 1.0 / kB, 0.7)
    assert not np.array_equal(s32, s)


@pytest.mark.parametrize("rng", ["philox", "numpy"])
def test_sample_chunks_gathered_equal_one_run(P, rng):
    """distributed.sample_chunks_sharded / HMC.sampleChunksGathered (collection overlapped with sampling: side
    stream, two send / two receive buffers) on ONE process: the concatenated chunks are getSamples' result bit
    for bit, in both RNG modes, with a short last chunk, momenta included."""
    from physicsbasedbayesianinference_amd.distributed import get_samples_sharded, sample_chunks_sharded
    D, N, S, chunk, seed = 12, 333, 11, 4, 5
    rs = np.random.RandomState(1)
    pot = P.GaussianDiag(rs.standard_normal(D), prec=rs.uniform(0.5, 2.0, D), const=0.0)
    m = 1.0 + (np.arange(N) % 3) * 0.5
    np.random.seed(seed)
    s_ref, m_ref, h_ref = get_samples_sharded(pot, D, N, 0.5, 0.1, S, 1.0 / kB, 1.0, rng=rng, seed=seed, mass=m)
    np.random.seed(seed)
    got_s, got_m, last = [], [], None
    for bs, bm, hmc in sample_chunks_sharded(pot, D, N, 0.5, 0.1, S, chunk, 1.0 / kB, 1.0, rng=rng, seed=seed, mass=m,
                                             momenta=True):
        assert bs.blocks.shape[0] == 1 and bs.sizes == [N] and bs.view4().shape[2:] == (1, N)
        got_s.append(bs.to_sdn().cpu().numpy())
        got_m.append(bm.to_sdn().cpu().numpy())
        last = hmc
    got_s, got_m = np.concatenate(got_s), np.concatenate(got_m)
    assert [len(x) for x in (got_s, got_m)] == [S, S]
    assert np.array_equal(got_s, s_ref.permute(2, 0, 1).cpu().numpy())
    assert np.array_equal(got_m, m_ref.permute(2, 0, 1).cpu().numpy())
    assert abs(last.acceptRate - h_ref.acceptRate) < 1e-12


def test_describe_run_names_the_route_and_the_carry_cliff(P, lib):
    """pbbi_describe_run: the kernel family, the carried-gradient state (with the N = 2^31 / (16 D) cliff of the
    dense path spelled out) and the iterations per launch, without launching anything."""
    import ctypes
    Pm = np.eye(128) + 0.01

    def describe(pot, N, L=10, S=100, flags=1, method=0):
        buf = ctypes.create_string_buffer(1024)
        lib.call("pbbi_describe_run", pot.handle, method, N, N, L, S, flags, buf, len(buf))
        return buf.value.decode()
    dense = P.GaussianDense(None, precision=Pm, const=0.0)
    d = describe(dense, 65536)
    assert "k_dense_hmc" in d and "carried between iterations: yes" in d and "up to 64" in d
    d = describe(dense, 1 << 21)                      # past the cliff: 2 * D * N * 8 bytes would pass 2^32
    assert "carried between iterations: no" in d and "N <= 2096639 chains" in d and "shard" in d
    assert "carried between iterations: yes" in describe(dense, 2096639)
    assert "carried between iterations: yes" in describe(dense, 1000, method=1)   # Stormer-Verlet carries too
    assert "fixed trajectory length" in describe(dense, 1000, flags=1 | lib.PER_CHAIN_STEPS)
    assert "carried between iterations: yes" in describe(P.GaussianDense(None, precision=np.eye(32), const=0.0), 1000)
    assert "a run of one iteration" in describe(dense, 1000, S=1)
    assert "k_ros2_hmc" in describe(P.Rosenbrock(32), 4096, flags=1 | lib.KDK_FMA)
    assert "double precision" in describe(P.Rosenbrock(32), 4096, flags=1 | lib.DRAW_F64)
    assert "k_sep_hmc" in describe(P.GaussianDiag(np.zeros(64), prec=np.ones(64), const=0.0), 4096, flags=1 | lib.KDK_FMA)
    d = describe(P.GaussianDense(None, precision=np.eye(256), const=0.0), 512)   # 128 < D <= 256: P streamed
    assert "streamed P" in d and "carried between iterations: yes" in d and "up to 64" in d
    assert "kernels_big" in describe(P.GaussianDense(None, precision=np.eye(300), const=0.0), 512)
    assert "kernels_big" in describe(P.GaussianDense(None, precision=np.eye(256), const=0.0, dtype="float32"), 512)
    hmc = P.HMC(P.Ensemble(128, 256), 1.0, 0.1, None, potential=dense, rng="philox", verbose=False)
    assert "k_dense_hmc" in hmc.describeRun(10)


def test_numpy_stream_run_that_dies_hands_back_the_right_rng_state(P, monkeypatch):
    """The rng="numpy" producer thread draws ahead of the launches.  When a launch raises, the global NumPy
    stream must be where the reference's would be after the iterations that WERE launched (its loop draws p and u
    of iteration i right before using them, src/HMC.py:154,168), ens.p must not alias a pinned staging buffer, and
    releaseHostBuffers() drops the cached staging buffers."""
    from physicsbasedbayesianinference_amd import _lib
    D, N, S, fail_at, seed = 4, 3000, 8, 3, 17
    pot = P.StandardGaussian(D)
    ens = P.Ensemble(D, N)
    hmc = P.HMC(ens, 0.3, 0.1, None, potential=pot, verbose=False)
    real_call, count = _lib.call, [0]

    def flaky(name, *args):
        if name == "pbbi_hmc_iter_kt":
            count[0] += 1
            if count[0] == fail_at + 1:
                raise _lib.PbbiError(-3, "injected failure")
        return real_call(name, *args)
    monkeypatch.setattr(_lib, "call", flaky)
    np.random.seed(seed)
    with pytest.raises(_lib.PbbiError):
        hmc.getSamples(S, 1.0 / kB, 1.0)
    after = np.random.standard_normal(3)
    # the reference at that point: q0, then fail_at complete iterations (p and u each), then the p of the
    # iteration whose launch failed has NOT been consumed by our loop (it raised before using it)
    rs = np.random.RandomState(seed)
    rs.standard_normal((D, N))
    for _ in range(fail_at):
        rs.standard_normal((D, N))
        rs.uniform(size=N)
    assert np.array_equal(after, rs.standard_normal(3))
    assert isinstance(ens.p, np.ndarray) and ens.p.flags.owndata
    # a healthy run afterwards, then the staging buffers can be dropped and re-made
    monkeypatch.setattr(_lib, "call", real_call)
    np.random.seed(seed)
    s1, _ = hmc.getSamples(3, 1.0 / kB, 1.0)
    assert hmc._host_cache
    hmc.releaseHostBuffers()
    assert not hmc._host_cache
    np.random.seed(seed)
    s2, _ = hmc.getSamples(3, 1.0 / kB, 1.0)
    assert np.array_equal(s1, s2)


# ---------------------------------------------------------------------------------------------
# GIST: the self-tuning no-U-turn sampler (pbbi_hmc_run_gist) -- a reversible per-chain dynamic length.
# ---------------------------------------------------------------------------------------------
def _gist_run(lib, pot, q0, m, h, Lmax, S, flags, seed, iter0, chain0, kT=1.0):
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    D, N = q0.shape
    qd = as_device(q0, 0, np.float64)
    md = as_device(m, 0, np.float64) if m is not None else None
    samples, momenta = empty((S, D, N), np.float64, 0), empty((S, D, N), np.float64, 0)
    reject, ratio, tau = empty((S, N), np.uint8, 0), empty((S, N), np.float64, 0), empty((S, 3, N), np.int32, 0)
    lib.call("pbbi_hmc_run_gist", pot.handle, qd.data_ptr(), md.data_ptr() if m is not None else None,
             samples.data_ptr(), momenta.data_ptr(), reject.data_ptr(), ratio.data_ptr(), tau.data_ptr(), N, N, h, Lmax,
             S, flags, seed, iter0, chain0, kT, stream_ptr(0))
    torch.cuda.synchronize()
    return (to_numpy(samples), to_numpy(momenta), to_numpy(reject).astype(bool), to_numpy(ratio), to_numpy(tau),
            to_numpy(qd))


@pytest.mark.parametrize("case,mass,draw64", [("diag8", True, False), ("ros12", False, True), ("ros32", True, False),
                                              ("harm3", False, False)])
def test_gist_lane_kernels_bitexact_vs_oracle(P, lib, case, mass, draw64):
    """pbbi_hmc_run_gist on the chain-per-lane kernels (reference operation order): U-turn counts tau_f / tau_b,
    drawn lengths, decisions, positions and momenta of every iteration equal to oracle_hmc_iter_gist bit for bit."""
    rs = np.random.RandomState(3)
    if case == "diag8":
        D, h = 8, 0.25
        mu, prec = rs.standard_normal(D), rs.uniform(0.3, 3.0, D)
        pot, op = P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec)
    elif case == "harm3":
        D, h = 3, 0.2
        k = np.array([0.5, 2.0, 9.0])
        pot, op = P.Harmonic(k), orc.pot_harmonic(k)
    else:
        D, h = int(case[3:]), 0.03
        pot, op = P.Rosenbrock(D), orc.pot_rosenbrock(D)
    N, S, Lmax, seed, iter0, chain0, kT = 500, 4, 40, 13, 2, 100, 1.0
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    q0 = (1.0 if case.startswith("ros") else 0.0) + 0.5 * rs.standard_normal((D, N))
    flags = lib.COMPAT_P_FROM_OLDQ | (lib.DRAW_F64 if draw64 else 0)
    gs, gm, grej, gratio, gtau, gq = _gist_run(lib, pot, q0, m, h, Lmax, S, flags, seed, iter0, chain0, kT)
    q = np.ascontiguousarray(q0)
    pstd = np.sqrt((m if mass else np.ones(N)) * kT)
    n_rej = 0
    for i in range(S):
        if draw64:
            p = orc.philox_normal(seed, orc.STREAM_MOMENTUM | orc.STREAM_DRAW_F64, iter0 + i, chain0, D, N, pstd)
        else:
            p = device_normal(lib, seed, lib.STREAM_MOMENTUM, iter0 + i, chain0, D, N, 1.0, pstd)
        p = np.ascontiguousarray(p)
        ua = orc.philox_uniform(seed, iter0 + i, chain0, N)
        ul = orc.philox_steps_uniform(seed, iter0 + i, chain0, N)
        ratio, rej, tau = orc.hmc_iter_gist(op, q, p, ua, ul, m, h, Lmax)
        assert np.array_equal(gtau[i], tau), f"iteration {i}"
        assert np.array_equal(grej[i], rej)
        assert np.array_equal(gs[i], q) and np.array_equal(gm[i], p)
        ok = np.isfinite(ratio) & (ratio > 0)
        assert np.allclose(np.log(gratio[i][ok]), np.log(ratio[ok]), atol=1e-8)
        n_rej += int(rej.sum())
    assert np.array_equal(gq, gs[-1])
    assert 0.01 < n_rej / (S * N) < 0.8
    assert gtau[:, 0].min() >= 1 and (gtau[:, 1] <= gtau[:, 0]).all() and len(np.unique(gtau[:, 0])) > 3


@pytest.mark.parametrize("D,mass", [(128, False), (100, True), (24, False), (160, True), (256, False)])   # D > 128: streamed-P kernel
def test_gist_dense_kernel_vs_oracle(P, lib, D, mass):
    """... and on the dense MFMA kernel (kick-drift-kick values: a U-turn dot product that passes zero within
    rounding may be seen one step apart from the oracle; such chains -- none or very few -- are left out)."""
    rs = np.random.RandomState(D)
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    Pm = 0.5 * (Pm + Pm.T)
    pot, op = P.GaussianDense(None, precision=Pm, const=0.0), orc.pot_gauss_dense(np.zeros(D), Pm)
    N, S, Lmax, h, seed = 333, 3, 60, 0.2, 5
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    q0 = rs.standard_normal((D, N))
    gs, gm, grej, gratio, gtau, _ = _gist_run(lib, pot, q0, m, h, Lmax, S, lib.COMPAT_P_FROM_OLDQ, seed, 0, 0)
    pstd = np.sqrt(m if mass else np.ones(N))
    q_start = q0
    agree = 0
    for i in range(S):
        q = np.ascontiguousarray(q_start)
        p = np.ascontiguousarray(device_normal(lib, seed, lib.STREAM_MOMENTUM, i, 0, D, N, 1.0, pstd))
        ua, ul = orc.philox_uniform(seed, i, 0, N), orc.philox_steps_uniform(seed, i, 0, N)
        ratio, rej, tau = orc.hmc_iter_gist(op, q, p, ua, ul, m, h, Lmax)
        same = (gtau[i] == tau).all(axis=0)
        with np.errstate(divide="ignore"):
            clear = np.abs(np.log(ua) - np.minimum(0.0, np.log(np.maximum(ratio, 1e-300)))) > 1e-6
        ok = same & clear
        agree += int(same.sum())
        assert np.array_equal(grej[i][ok], rej[ok])
        assert scaled_err(gs[i][:, ok], q[:, ok]) <= RTOL_DENSE and scaled_err(gm[i][:, ok], p[:, ok]) <= RTOL_DENSE
        q_start = gs[i]   # continue from the kernel's own state (a chain left out above would drift apart)
    assert agree >= 0.98 * S * N


def test_gist_samples_a_correlated_gaussian_with_a_resonant_mode(P):
    """The point of a dynamic length: a target with widely spread frequencies.  One eigen-direction has
    omega * T = 2 pi for the fixed-length sampler's T -- it comes back to where it started and never mixes (its
    variance stays at the initial 0.01) -- while GIST, whose lengths follow each chain's own U-turn, recovers the
    whole covariance within Monte-Carlo error."""
    D, N = 6, 4096
    rs = np.random.RandomState(1)
    Qm, _ = np.linalg.qr(rs.standard_normal((D, D)))
    h, L = 0.1, 10                                     # fixed-length sampler: T = 1
    # the LEAPFROG frequency of mode 0 is exactly 2 pi / (L h): cos(w h) = 1 - (h omega)^2 / 2 with w L h = 2 pi
    omega = np.array([2 * np.sin(np.pi / L) / h, 0.7, 1.3, 2.1, 3.3, 0.4])
    Pm = (Qm * omega ** 2) @ Qm.T
    Pm = 0.5 * (Pm + Pm.T)
    cov = (Qm / omega ** 2) @ Qm.T
    pot = P.GaussianDense(None, precision=Pm, const=0.0)
    # fixed length: the resonant direction keeps its initial spread
    hmc = P.HMC(P.Ensemble(D, N), L * h + 1e-9, h, None, potential=pot, rng="philox", seed=3, verbose=False)
    s, _ = hmc.getSamples(60, 1.0 / kB, 0.1)
    z = np.tensordot(Qm[:, 0], s[:, :, 30:], axes=(0, 0))
    assert abs(z.var() / 0.01 - 1.0) < 0.1             # exactly where it started (qStd^2 = 0.01), not 1/omega^2 = 0.026
    # GIST
    hmc = P.HMC(P.Ensemble(D, N), L * h + 1e-9, h, None, potential=pot, rng="philox", seed=3, verbose=False)
    s, _ = hmc.getSamplesGIST(80, 1.0 / kB, 0.1, max_steps=200)
    x = s[:, :, 30:].reshape(D, -1)
    emp = np.cov(x)
    assert np.max(np.abs(x.mean(axis=1))) < 0.05
    # in the eigenbasis: every variance within 6 % of 1/omega^2, off-diagonals small
    C = Qm.T @ emp @ Qm * np.outer(omega, omega)
    assert np.max(np.abs(np.diag(C) - 1.0)) < 0.06, np.diag(C)
    assert np.max(np.abs(C - np.diag(np.diag(C)))) < 0.05
    assert 0.3 < hmc.acceptRate < 0.95
    tau = hmc.gist_tau
    assert tau[:, 0].mean() > 5 and tau[:, 0].std() > 1      # lengths really vary chain by chain


# ---------------------------------------------------------------------------------------------
# Tempering that uses the temperature: replica exchange between rungs (pbbi_replica_exchange, tempering.py)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,parity", [(4, 0), (4, 1), (5, 0), (5, 1), (2, 0), (1, 0)])
def test_replica_exchange_kernel_vs_oracle(P, lib, R, parity):
    """The device exchange step against oracle_replica_exchange on the same state: the same pairs swap, the
    state afterwards is the same bit for bit, untouched rungs stay untouched (potentials whose evaluation kernel
    is bit-exact with the oracle: diagonal Gaussian)."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, stream_ptr, to_numpy
    D, Nr, seed, it, chain0 = 5, 777, 31, 9, 2 ** 33
    rs = np.random.RandomState(R * 2 + parity)
    mu, prec = rs.standard_normal(D), rs.uniform(0.5, 2.0, D)
    pot, op = P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec)
    kTs = 1.7 ** np.arange(R)
    q0 = rs.standard_normal((D, R * Nr)) * np.repeat(np.sqrt(kTs), Nr)[None, :]
    qd = as_device(q0, 0, np.float64)
    betas = as_device(1.0 / kTs, 0, np.float64)
    sw = torch.zeros((max(R - 1, 1), Nr), dtype=torch.uint8, device="cuda")
    lib.call("pbbi_replica_exchange", pot.handle, qd.data_ptr(), Nr, R, R * Nr, betas.data_ptr(), parity, seed, it,
             chain0, sw.data_ptr(), stream_ptr(0))
    torch.cuda.synchronize()
    q_or = np.ascontiguousarray(q0)
    sw_or = orc.replica_exchange(op, q_or, Nr, R, 1.0 / kTs, parity, seed, it, chain0)
    assert np.array_equal(to_numpy(qd), q_or)
    if R > 1:
        assert np.array_equal(to_numpy(sw)[:R - 1].astype(bool), sw_or)
        if parity + 1 < R:
            assert 0.2 < sw_or[parity::2].mean() < 0.95      # some pairs swap, some do not
        assert not sw_or[1 - parity::2].any()
    else:
        assert np.array_equal(to_numpy(qd), q0)


def test_tempering_ladder_recovers_both_modes_of_a_bimodal_target(P):
    """A 2-D mixture 0.7 N((-4, 0), 0.6^2) + 0.3 N((+4, 0), 0.6^2), written as a Python callable on the traceable
    namespace: a barrier of ~22 kT between the modes.  Started symmetrically, a single-temperature ensemble keeps
    half its chains in each mode for ever (weight 0.5); the ladder's rung 0 recovers the weights 0.7 / 0.3 within
    3 %, with the right within-mode spread."""
    from physicsbasedbayesianinference_amd import trace as jnp
    from physicsbasedbayesianinference_amd.tempering import TemperingLadder, geometric_ladder
    a, b, sig, wa = np.array([-4.0, 0.0]), np.array([4.0, 0.0]), 0.6, 0.7

    def potential(q):
        la = np.log(wa) - 0.5 * jnp.sum((q - a) ** 2) / sig ** 2
        lb = np.log(1.0 - wa) - 0.5 * jnp.sum((q - b) ** 2) / sig ** 2
        return -jnp.logaddexp(la, lb)
    Nr = 4096
    # one temperature: the modes never exchange chains
    hmc = P.HMC(P.Ensemble(2, Nr), 1.0, 0.1, None, potential=potential, rng="philox", seed=5, verbose=False)
    s, _ = hmc.getSamples(120, 1.0 / kB, 4.0)
    left0 = (s[0, :, 0] < 0).mean()
    left = (s[0, :, 60:] < 0).mean()
    assert abs(left - left0) < 0.02 and abs(left - wa) > 0.1            # stuck at the initial split
    # the ladder
    ladder = TemperingLadder(potential, 2, Nr, geometric_ladder(40.0, 7), simulTime=1.0, stepSize=0.1, seed=5)
    x = ladder.run(numSamples=60, qStd=4.0, swap_every=2, burn_in=500)
    assert x.shape == (2, Nr, 60)
    w_left = (x[0] < 0).mean()
    assert abs(w_left - wa) < 0.03, w_left
    for mode, sel in ((a, x[0] < 0), (b, x[0] >= 0)):
        pts = x[:, sel]
        assert np.max(np.abs(pts.mean(axis=1) - mode)) < 0.05
        assert np.max(np.abs(pts.std(axis=1) - sig)) < 0.05
    assert ladder.swapRates.shape == (6,) and ladder.swapRates.min() > 0.1
    assert ladder.acceptRates.min() > 0.5
    # rungs in between sample exp(-U / kT): the hot rung's spread within a mode is sig * sqrt(kT)
    both = ladder.run(numSamples=20, qStd=4.0, swap_every=2, burn_in=200, record_rungs=(0, 3))
    hot = both[3]
    kT3 = ladder.kTs[3]
    assert abs(hot[1].std() / (sig * np.sqrt(kT3)) - 1.0) < 0.1


def test_eight_schools_hierarchical_model(P):
    """The reference's hierarchical example (samples/NumpyroExamples/eight_schools.py, its data file) as a traced
    potential.  Parity unpinned: NumPyro is not importable here and the reference records no posterior, so the
    check is (a) the two parametrisations -- different potentials, different geometry -- agree on the posterior of
    (mu, tau, theta), and (b) the classic summaries of this data set (Gelman et al., BDA3 5.5: E[mu] ~ 4.4,
    E[tau] ~ 3.6, shrunken school effects between 3 and 7)."""
    from physicsbasedbayesianinference_amd.models import eight_schools_constrain, eight_schools_potential
    N = 8192
    post = {}
    for centered, (T, h, S, B) in ((False, (1.0, 0.05, 60, 150)), (True, (1.5, 0.015, 200, 400))):
        pot = eight_schools_potential(centered=centered)
        hmc = P.HMC(P.Ensemble(10, N), T, h, None, potential=pot, rng="philox", seed=2, verbose=False, kdk_fma=False)
        assert hmc._pot.kind == "custom"
        s, _ = hmc.getSamples(S, 1.0 / kB, 1.0, burn_in=B)
        assert hmc.acceptRate > 0.6
        post[centered] = eight_schools_constrain(s, centered)
    nc = post[False]
    assert abs(nc["mu"].mean() - 4.4) < 0.4 and abs(nc["tau"].mean() - 3.6) < 0.5
    th = nc["theta"].mean(axis=(1, 2))
    # shrunken school effects: every one between 3 and 7, schools A and G (y = 28, 18) on top, E (y = -1) lowest
    assert th.min() > 2.5 and th.max() < 8.0 and set(np.argsort(th)[-2:]) == {0, 6} and th.argmin() == 4
    # the centred run (funnel: slower, only loosely converged) tells the same story
    c = post[True]
    assert abs(c["mu"].mean() - nc["mu"].mean()) < 0.6
    assert abs(np.median(c["tau"]) - np.median(nc["tau"])) < 1.0
    assert np.max(np.abs(c["theta"].mean(axis=(1, 2)) - th)) < 1.0


_TWO_RANK_GPU_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from scipy.constants import k as kB
rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=2)
torch.cuda.set_device(0)                      # both ranks on the one GPU of this box: the collective is gloo's
import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd.distributed import (get_samples_sharded, sample_chunks_sharded, shard_bounds)
D, N, S, chunk, seed = 128, {N}, 7, 3, 11
rs = np.random.RandomState(0); A = rs.standard_normal((D, D)); Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0)
mass = 1.0 + (np.arange(N) % 3) * 0.5
for rng in ("philox", "numpy"):
    # the whole ensemble in ONE process (rank 0 and rank 1 both compute it: it is the reference)
    np.random.seed(seed)
    ens = P.Ensemble(D, N); ens.mass = mass.copy()
    ref_hmc = P.HMC(ens, 0.5, 0.1, None, potential=pot, rng=rng, seed=seed, verbose=False)
    ref, ref_m = ref_hmc.getSamples(S, 1.0 / kB, 1.0)
    # sharded, one all-gather at the end
    np.random.seed(seed)
    s, m, _ = get_samples_sharded(pot, D, N, 0.5, 0.1, S, 1.0 / kB, 1.0, rng=rng, seed=seed, mass=mass)
    assert tuple(s.shape) == (D, N, S)
    assert np.array_equal(s.cpu().numpy(), ref) and np.array_equal(m.cpu().numpy(), ref_m), "sharded run differs: " + rng
    # sharded, collected chunk by chunk while the next chunk samples
    np.random.seed(seed)
    got = []
    for bs, bm, hmc in sample_chunks_sharded(pot, D, N, 0.5, 0.1, S, chunk, 1.0 / kB, 1.0, rng=rng, seed=seed,
                                             mass=mass, momenta=True):
        assert bs.blocks.is_cuda and bs.blocks.shape[0] == 2
        assert bs.sizes == [shard_bounds(N, r, 2)[1] - shard_bounds(N, r, 2)[0] for r in range(2)]
        got.append((bs.to_sdn().cpu().numpy(), bm.to_sdn().cpu().numpy()))
    gs = np.concatenate([g[0] for g in got]); gm = np.concatenate([g[1] for g in got])
    assert np.array_equal(gs.transpose(1, 2, 0), ref) and np.array_equal(gm.transpose(1, 2, 0), ref_m), "chunked: " + rng
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("N", [512, 333])
def test_two_ranks_on_one_gpu_sharded_kernels_and_overlapped_collection(tmp_path, N):
    """Two processes, both on this box's one GPU, gloo between them (RCCL refuses two ranks on one device): the
    SHARDED path end to end with the real kernels -- Philox counters offset by the shard's first chain, the NumPy
    stream's columns, one all-gather at the end, and the chunked collection that overlaps chunk k's gather (side
    stream) with chunk k+1's kernels -- equals the one-process ensemble bit for bit, even and uneven split."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_TWO_RANK_GPU_WORKER.format(root=root, port=port, N=N))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


@pytest.mark.parametrize("kind,D,mass", [("diag", 64, False), ("diag", 200, True), ("diag", 40, True),
                                         ("ros", 64, True), ("ros", 128, False), ("ros", 100, False)])
@pytest.mark.parametrize("rng", ["upload", "philox"])
def test_per_chain_steps_multilane_kdk_kernels(P, lib, kind, D, mass, rng):
    """PBBI_PER_CHAIN_STEPS in the multi-lane kick-drift-kick kernels (round 3: k_sep_hmc / k_rosg_hmc <DYN>, D > 32;
    finished chains frozen by per-lane coefficients): step counts and decisions equal to the oracle's
    leapfrog_chain_dyn, states within the kick-drift-kick tolerance; 0 steps leaves a chain where it was."""
    import torch
    from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy
    rs = np.random.RandomState(D)
    if kind == "diag":
        mu, prec = rs.standard_normal(D), rs.uniform(0.5, 2.0, D)
        pot, op, h = P.GaussianDiag(mu, prec=prec, const=0.0), orc.pot_gauss_diag(mu, prec), 0.15
        q0 = rs.standard_normal((D, 333))
    else:
        pot, op, h = P.Rosenbrock(D), orc.pot_rosenbrock(D), 0.02
        q0 = 1.0 + 0.2 * rs.standard_normal((D, 333))
    N, L, seed = 333, 9, 21
    m = 1.0 + (np.arange(N) % 3) * 0.5 if mass else None
    md = as_device(m, 0, np.float64) if mass else None
    flags = lib.COMPAT_P_FROM_OLDQ | lib.PER_CHAIN_STEPS | lib.KDK_FMA
    qd = as_device(q0, 0, np.float64)
    qo, po = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0)
    ro, rj, so = empty((N,), np.float64, 0), empty((N,), np.uint8, 0), empty((N,), np.int32, 0)
    if rng == "upload":
        p0 = rs.standard_normal((D, N)) * (np.sqrt(m) if mass else 1.0)
        u = rs.uniform(size=N)
        steps_in = rs.randint(0, L + 1, size=N).astype(np.int32)
        steps_in[:70] = 0                                          # a whole wave of chains that do not move
        pd, ud, sd = as_device(p0, 0, np.float64), as_device(u, 0, np.float64), torch.tensor(steps_in, device="cuda")
        lib.call("pbbi_hmc_iter_dyn", pot.handle, 0, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
                 md.data_ptr() if mass else None, sd.data_ptr(), qo.data_ptr(), po.data_ptr(), ro.data_ptr(),
                 rj.data_ptr(), so.data_ptr(), N, N, h, L, flags, 1.0, stream_ptr(0))
    else:
        p0 = device_normal(lib, seed, lib.STREAM_MOMENTUM, 0, 0, D, N, 1.0, np.sqrt(m) if mass else None)
        u = device_uniform(lib, seed, 0, 0, N)
        steps_in = orc.philox_steps(seed, 0, 0, N, L)
        lib.call("pbbi_hmc_run_dyn", pot.handle, 0, qd.data_ptr(), md.data_ptr() if mass else None, qo.data_ptr(),
                 po.data_ptr(), rj.data_ptr(), ro.data_ptr(), so.data_ptr(), N, N, h, L, 1, flags, seed, 0, 0, 1.0,
                 stream_ptr(0))
    torch.cuda.synchronize()
    q_or, p_or = np.ascontiguousarray(q0), np.ascontiguousarray(p0)
    r_or, rej_or, st_or = orc.hmc_iter_dyn(op, q_or, p_or, u, m, h, L, steps_in=steps_in)
    assert np.array_equal(to_numpy(so), st_or)
    assert np.array_equal(to_numpy(rj).astype(bool), rej_or)
    assert scaled_err(to_numpy(qo), q_or) <= 1e-12 and scaled_err(to_numpy(po), p_or) <= 1e-12
    if rng == "upload":   # (the separable kernel stores (q - mu) + mu: the start up to one rounding)
        assert np.max(np.abs(to_numpy(qo)[:, :70] - q0[:, :70])) <= 2e-15
        assert not to_numpy(rj).astype(bool)[:70].any()
    # the U-turn stop stays with the kernels that can form the dot product
    bad = lib.load().pbbi_hmc_iter_dyn(pot.handle, 0, qd.data_ptr(), qd.data_ptr(), ro.data_ptr(), None, None,
                                       qo.data_ptr(), None, None, None, None, N, N, h, L,
                                       flags | lib.UTURN_STOP, 1.0, None)
    assert bad == -2   # PBBI_ERR_UNSUPPORTED


@pytest.mark.parametrize("case", ["diag8", "ros32", "harm3"])
def test_gist_fused_lane_kernel_equals_the_composed_form(P, lib, case, monkeypatch):
    """pbbi_hmc_run_gist serves elementwise potentials with D <= 32 by ONE launch per iteration (k_lane_gist_hmc:
    forward search, length draw, proposal, backward search and accept test of a chain in its lane); every other
    handle -- and these with PBBI_GIST_COMPOSED=1 -- by three masked launches plus the small kernels.  Same
    arithmetic: the two forms agree bit for bit in every output."""
    rs = np.random.RandomState(7)
    if case == "diag8":
        D, h = 8, 0.25
        pot = P.GaussianDiag(rs.standard_normal(D), prec=rs.uniform(0.3, 3.0, D), const=0.0)
    elif case == "harm3":
        D, h = 3, 0.2
        pot = P.Harmonic(np.array([0.5, 2.0, 9.0]))
    else:
        D, h = 32, 0.03
        pot = P.Rosenbrock(D)
    N, S, Lmax, seed = 700, 5, 50, 3
    m = 1.0 + (np.arange(N) % 3) * 0.5
    q0 = (1.0 if case == "ros32" else 0.0) + 0.5 * rs.standard_normal((D, N))
    monkeypatch.delenv("PBBI_GIST_COMPOSED", raising=False)
    fused = _gist_run(lib, pot, q0, m, h, Lmax, S, lib.COMPAT_P_FROM_OLDQ, seed, 1, 9)
    monkeypatch.setenv("PBBI_GIST_COMPOSED", "1")
    composed = _gist_run(lib, pot, q0, m, h, Lmax, S, lib.COMPAT_P_FROM_OLDQ, seed, 1, 9)
    for a, b in zip(fused, composed):
        assert np.array_equal(a, b, equal_nan=True)
    assert 0.0 < fused[2].mean() < 0.9


def test_traced_sum_over_data_is_rolled_into_a_loop(P, lib):
    """A likelihood written as a sum over M = 256 observations traces to ~10 000 operations; as straight-line code
    hipcc does not finish on it.  The tracer groups the terms of the sum by shape and emits ONE loop per shape over
    a table of constants (the plugin's parameter array): the kernel agrees with the hand-written C++ logistic
    regression of custom.py to 1e-12, HMC iterations agree with the oracle running the same generated source."""
    from physicsbasedbayesianinference_amd import trace as jnp
    from physicsbasedbayesianinference_amd.custom import complete_source, logistic_regression_posterior
    rs = np.random.RandomState(0)
    M, D, N = 256, 16, 512
    X = rs.standard_normal((M, D))
    y = (rs.uniform(size=M) < 0.5).astype(np.float64)

    def softplus(z):
        return jnp.maximum(z, 0.0) + jnp.log1p(jnp.exp(-jnp.abs(z)))

    fn = lambda w: jnp.sum(softplus(X @ w) - y * (X @ w)) + 0.5 * jnp.dot(w, w)   # noqa: E731
    pot = jnp.trace_potential(fn, D=D)
    assert pot.kind == "custom" and pot.params.size >= M * D and pot.traced_source.count("for (int i") == 4
    assert len(pot.traced_source) < 64 * 1024
    q = rs.standard_normal((D, N)) * 0.5
    U, gr = pot.value_and_gradient(q)
    Uh, gh = logistic_regression_posterior(X, y, 1.0).value_and_gradient(q)
    assert scaled_err(U, Uh) <= 1e-12 and scaled_err(gr, gh) <= 1e-12
    op = orc.pot_custom(complete_source(pot.traced_source), D, pot.params)
    h, L = 0.05, 5
    for it in range(2):
        p, u = rs.standard_normal((D, N)), rs.uniform(size=N)
        qo, po, ratio, rej = gpu_hmc_iter(lib, pot, "Leapfrog", q, p, u, None, h, L)
        q_or, p_or = q.copy(), p.copy()
        _, rej_or = orc.hmc_iter(op, "Leapfrog", q_or, p_or, u, None, h, L)
        assert np.array_equal(rej, rej_or)
        assert scaled_err(qo, q_or) <= 1e-10 and scaled_err(po, p_or) <= 1e-10
        q = qo
