"""Pins the CPU oracle (oracle/pbbi_oracle.c) against golden vectors produced
by the reference's own unmodified Python (tests/golden/gen_golden.py).

Tolerance: trajectories/energies agree to RTOL = 1e-12 (observed ~1e-15; the
only differences are NumPy's BLAS/pairwise summation order inside np.dot / @ /
np.sum versus the oracle's sequential loops); reject masks must be EQUAL.
"""
import numpy as np
import pytest
from scipy.constants import k as kB

from conftest import load_golden
from oracle import oracle as orc

RTOL = 1e-12


def close(a, b, rtol=RTOL, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), what + ": NaN pattern differs"
    inf = np.isinf(a) | np.isinf(b)
    assert np.array_equal(a[inf], b[inf]), what + ": inf pattern differs"
    ok = ~(nan_a | inf)
    scale = max(1.0, float(np.max(np.abs(b[ok])))) if ok.any() else 1.0
    err = float(np.max(np.abs(a[ok] - b[ok]))) / scale if ok.any() else 0.0
    assert err <= rtol, f"{what}: scaled max error {err:.3e} > {rtol:.1e}"


def pot_of(g, kind):
    if kind == "dense":
        return orc.pot_gauss_dense(g["mean"], g["precision"], float(g["const"]))
    if kind == "std":
        D = int(g["D"])
        return orc.pot_gauss_diag(np.zeros(D), np.ones(D))
    if kind == "rosenbrock":
        return orc.pot_rosenbrock(int(g["D"]), float(g["a"]), float(g["b"]), float(g["s"]))
    raise KeyError(kind)


GETSAMPLES = [
    ("G3_getsamples_c1", "std"),
    ("G4_getsamples_dense_d8", "dense"),
    ("G4_getsamples_dense_d128", "dense"),
    ("G4b_getsamples_dense_mean_d16", "dense"),
    ("G5_getsamples_rosenbrock_d32", "rosenbrock"),
    ("G6_getsamples_rejects", "std"),
    ("G7_getsamples_nan", "std"),
    ("G8_getsamples_mass", "dense"),
    ("G11_getsamples_test2", "dense"),
    ("G12_getsamples_stormerverlet", "dense"),
]


@pytest.mark.parametrize("name,kind", GETSAMPLES)
def test_getsamples_iterations(name, kind):
    """Feed the recorded q-state / p-draw / u of every iteration to oracle_hmc_iter."""
    g = load_golden(name)
    pot = pot_of(g, kind)
    D, N, S = int(g["D"]), int(g["N"]), int(g["S"])
    L = int(g["numSteps"])
    method = str(g["method"])
    q = np.ascontiguousarray(g["q0"])
    for i in range(S):
        p = np.ascontiguousarray(g["p_draw"][i])
        ratio, rej = orc.hmc_iter(pot, method, q, p, g["u"][i], g["mass"], float(g["stepSize"]), L)
        # decisions: identical reject mask
        assert np.array_equal(rej, g["reject_mask"][i]), f"{name} it {i}: reject mask differs"
        close(q, g["samples"][:, :, i], what=f"{name} samples[{i}]")
        close(p, g["momenta"][:, :, i], what=f"{name} momenta[{i}]")
        # ratio = exp(dH): relative error of exp(x) is |delta x|, and |x| can be large
        fin = np.isfinite(g["ratio"][i]) & (g["ratio"][i] > 0) & np.isfinite(ratio) & (ratio > 0)
        assert np.array_equal(np.isnan(ratio), np.isnan(g["ratio"][i]))
        if fin.any():
            dlog = np.abs(np.log(ratio[fin]) - np.log(g["ratio"][i][fin]))
            assert dlog.max() < 1e-9, (name, i, dlog.max())


@pytest.mark.parametrize("name,kind", GETSAMPLES)
def test_getsamples_full_stream(name, kind):
    """Whole getSamples from the seed alone: RNG order + loop restated."""
    g = load_golden(name)
    pot = pot_of(g, kind)
    r = orc.get_samples_numpy_stream(pot, str(g["method"]), int(g["D"]), int(g["N"]), int(g["S"]),
                                     float(g["simulTime"]), float(g["stepSize"]),
                                     float(g["temperature"]), float(g["qStd"]), int(g["seed"]),
                                     mass=g["mass"])
    assert r["numSteps"] == int(g["numSteps"])
    assert np.array_equal(r["reject_mask"], g["reject_mask"])
    close(r["samples"], g["samples"], what=name + " samples")
    close(r["momenta"], g["momenta"], what=name + " momenta")


def test_reject_quirks_visible():
    """G6 pins SURVEY appendix A items 1, 2, 11 on the reference's own output."""
    g = load_golden("G6_getsamples_rejects")
    S = int(g["S"])
    prev = g["q0"]
    n_rej = 0
    for i in range(S):
        rej = g["reject_mask"][i]
        n_rej += int(rej.sum())
        # rejected: sample repeats old point AND momentum_hmc holds old POSITION (HMC.py:176)
        assert np.array_equal(g["samples"][:, rej, i], prev[:, rej])
        assert np.array_equal(g["momenta"][:, rej, i], prev[:, rej])
        # accepted: un-negated final momentum
        assert np.array_equal(g["momenta"][:, ~rej, i], g["p_prop"][i][:, ~rej])
        prev = g["samples"][:, :, i]
    assert n_rej > 100


def test_nan_ratio_is_accepted():
    g = load_golden("G7_getsamples_nan")
    nan = np.isnan(g["ratio"])
    assert nan.sum() > 0
    assert not g["reject_mask"][nan].any()


INTEG = [
    ("G1_leapfrog_harmonic", "Leapfrog", ("_h0.1", "_h0.01")),
    ("G2_stormerverlet_harmonic", "Stormer-Verlet", ("_h0.1", "_h0.01")),
    ("G8b_leapfrog_mass", "Leapfrog", ("",)),
    ("G8c_stormerverlet_mass", "Stormer-Verlet", ("",)),
]


@pytest.mark.parametrize("name,method,sfxs", INTEG)
def test_integrators_harmonic(name, method, sfxs):
    g = load_golden(name)
    pot = orc.pot_harmonic(g["springConsts"])
    for s in sfxs:
        q, p = np.ascontiguousarray(g["q0" + s]), np.ascontiguousarray(g["p0" + s])
        v = orc.integrate(pot, method, q, p, g["mass" + s], float(g["stepSize" + s]),
                          int(g["numSteps" + s]))
        close(q, g["q" + s], what=name + s + " q")
        close(p, g["p" + s], what=name + s + " p")
        close(v, g["v" + s], what=name + s + " v")


def test_leapfrog_second_order_vs_analytic():
    """Reference's only integrator known-answer generator: the analytic
    oscillator (src/tests/test_integrator_harmonic.py:27-38), evaluated at the
    time the integrator really reaches, numSteps*h (numSteps truncates)."""
    g = load_golden("G1_leapfrog_harmonic")
    k = g["springConsts"]
    errs = []
    for s in ("_h0.1", "_h0.01"):
        t = int(g["numSteps" + s]) * float(g["stepSize" + s])
        om = np.sqrt(np.outer(k, 1.0 / g["mass" + s]))
        q0, v0 = g["q0" + s], g["p0" + s] / g["mass" + s]
        qa = q0 * np.cos(om * t) + v0 / om * np.sin(om * t)
        errs.append(np.max(np.abs(g["q" + s] - qa)))
    assert errs[1] < errs[0] / 50  # 2nd order: 10x smaller h -> ~100x smaller error
    assert errs[1] < 1e-2


def test_leapfrog_rosenbrock():
    g = load_golden("G5b_leapfrog_rosenbrock_d32")
    pot = orc.pot_rosenbrock(32, float(g["a"]), float(g["b"]), float(g["s"]))
    q, p = np.ascontiguousarray(g["q0"]), np.ascontiguousarray(g["p0"])
    orc.integrate(pot, "Leapfrog", q, p, g["mass"], float(g["stepSize"]), int(g["numSteps"]))
    close(q, g["q"], what="rosenbrock q")
    close(p, g["p"], what="rosenbrock p")


def test_weights_ratio_and_weights():
    g = load_golden("G4_getsamples_dense_d8")
    pot = pot_of(g, "dense")
    newQ = np.ascontiguousarray(g["q_prop"][0])
    newP = np.ascontiguousarray(-g["p_prop"][0])  # reference passes the negated momentum
    oldQ = np.ascontiguousarray(g["q0"])
    oldP = np.ascontiguousarray(g["p_draw"][0])
    r = orc.weights_ratio(pot, newQ, newP, oldQ, oldP, g["mass"])
    close(np.log(r), np.log(g["ratio"][0]), rtol=1e-10, what="log ratio")
    w, H = orc.weights(pot, oldQ, oldP, g["mass"])
    close(w, np.exp(-H), what="weights")


def test_known_answers():
    g = load_golden("G10_known_answers")
    U = orc.potential(orc.pot_harmonic(g["harmonic_k"].astype(float)), g["harmonic_q"])
    assert U[0] == 33.0 == float(g["harmonic_U"][0])  # src/tests/test_potential.py:25
    assert np.array_equal(U, g["harmonic_U"])


def test_numsteps_table():
    g = load_golden("G9_numsteps")
    for T, h, n in zip(g["finalTime"], g["stepSize"], g["numSteps"]):
        assert int(T / h) == int(n)
    # documented examples: int(1/0.1)=10, int(0.5/0.05)=10, int(0.3/0.1)=2
    assert list(g["numSteps"][:3]) == [10, 10, 2]


def test_step_size_square_matches_pow():
    """The oracle/kernels use h*h where the reference writes stepSize**2."""
    for name, _ in GETSAMPLES:
        h = float(load_golden(name)["stepSize"])
        assert h * h == h ** 2


# ------------------------------------------------------------------ Philox
def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert orc.philox_raw([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    f = 0xFFFFFFFF
    assert orc.philox_raw([f, f, f, f], [f, f]) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox_raw([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344],
                          [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420,
                                                        0x24126EA1]


def test_philox_normal_statistics_and_sharding():
    z = orc.philox_normal(123, orc.STREAM_MOMENTUM, 5, 0, 16, 20000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.corrcoef(z[0], z[4])[0, 1]) < 0.03  # the two Box-Muller branches
    # chain offset: a shard reproduces the matching columns of the full draw
    zs = orc.philox_normal(123, orc.STREAM_MOMENTUM, 5, 12000, 16, 500)
    assert np.array_equal(zs, z[:, 12000:12500])
    u = orc.philox_uniform(123, 5, 0, 20000)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    # different iteration / stream -> different numbers
    assert not np.array_equal(z, orc.philox_normal(123, orc.STREAM_MOMENTUM, 6, 0, 16, 20000))
    assert not np.array_equal(z, orc.philox_normal(123, orc.STREAM_POSITION, 5, 0, 16, 20000))


def test_philox_run_matches_iter_composition():
    D, N, S, L, h = 8, 33, 3, 10, 0.1
    g = load_golden("G4_getsamples_dense_d8")
    pot = pot_of(g, "dense")
    q0 = orc.philox_normal(9, orc.STREAM_POSITION, 0, 0, D, N)
    q = q0.copy()
    samples, momenta, rej, ratio = orc.hmc_run_philox(pot, "Leapfrog", q, None, h, L, S, seed=9)
    q2 = q0.copy()
    for i in range(S):
        p = orc.philox_normal(9, orc.STREAM_MOMENTUM, i, 0, D, N)
        u = orc.philox_uniform(9, i, 0, N)
        r2, rej2 = orc.hmc_iter(pot, "Leapfrog", q2, p, u, None, h, L)
        assert np.array_equal(samples[i], q2) and np.array_equal(momenta[i], p)
        assert np.array_equal(rej[i], rej2)
    assert np.array_equal(q, q2)


# ---- the double-precision draw of the RNG contract (PBBI_DRAW_F64, include/pbbi.h) -------------------
def test_f64_draw_matches_the_contract_step_by_step():
    """oracle_philox_normal with the PBBI_STREAM_DRAW_F64 bit against a step-by-step Python statement of the
    contract (tools/gen_draw_coeffs.py: exact rational fma, the same operation order): the same bits."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "gen_draw_coeffs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                        "gen_draw_coeffs.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    seed, it, D = 0x1234567890ABCDEF, 7, 40
    for chain0, N in ((0, 6), (2 ** 33 + 5, 3)):
        z = orc.philox_normal(seed, orc.STREAM_MOMENTUM | orc.STREAM_DRAW_F64, it, chain0, D, N)
        for n in range(N):
            chain = chain0 + n
            for d in range(D):
                slot = (d >> 2) & 3
                blk = (((d >> 4) << 2) | (d & 3)) | (0x80000000 if slot >= 2 else 0)
                x = orc.philox_raw([chain & 0xFFFFFFFF, blk, it, orc.STREAM_MOMENTUM | ((chain >> 32) << 8)],
                                   [seed & 0xFFFFFFFF, seed >> 32])
                assert G.box_muller(x)[slot & 1] == z[d, n], (d, n)


def test_f64_draw_is_standard_normal_with_long_tails():
    from scipy import stats
    z = orc.philox_normal(11, orc.STREAM_MOMENTUM | orc.STREAM_DRAW_F64, 0, 0, 64, 32768).ravel()   # 2.1e6 variates
    assert abs(z.mean()) < 4e-3 and abs(z.std() - 1.0) < 3e-3
    assert stats.kstest(z, "norm").pvalue > 1e-3
    assert abs(stats.skew(z)) < 0.01 and abs(stats.kurtosis(z)) < 0.02
    # pairs from one block are uncorrelated; dims of different blocks too
    zz = z.reshape(64, 32768)
    assert abs(np.corrcoef(zz[0], zz[4])[0, 1]) < 0.02 and abs(np.corrcoef(zz[0], zz[8])[0, 1]) < 0.02
    # not the single-precision draw, and finer than its 2^-24 grid
    zf = orc.philox_normal(11, orc.STREAM_MOMENTUM, 0, 0, 64, 32768).ravel()
    assert abs(np.corrcoef(z, zf)[0, 1]) < 0.01
    assert np.unique(z).size == z.size


def test_f64_draw_tail_reach():
    """The smallest radius argument (w1 >> 12 == 0, u1 = 2^-53) gives r = sqrt(106 ln 2) = 8.57 sigma: the tails
    the single-precision draw cuts at 6.7 sigma are there."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "gen_draw_coeffs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                        "gen_draw_coeffs.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    zc, zs = G.box_muller([0, 0, 0, 0])
    assert abs(zc - np.sqrt(106 * np.log(2.0))) < 1e-14 and zs == 0.0
    zc, zs = G.box_muller([0xFFF, 0, 0, 0x40000000])      # still u1 = 2^-53; angle pi/2
    assert abs(zs - np.sqrt(106 * np.log(2.0))) < 1e-14 and abs(zc) < 1e-15


def test_oracle_under_address_and_ub_sanitizers():
    """`make -C oracle asan_check`: every oracle entry point driven over exactly-sized heap buffers with
    -fsanitize=address,undefined (GPU sanitizers are unavailable on the pool; the CPU build is where they run)."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    res = subprocess.run(["make", "-C", here, "-B", "asan_check"], capture_output=True, text=True,
                         env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "asan ok" in res.stdout
