#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Build-container only.  This script imports the reference's unmodified
`/root/reference/src/{ensemble,integrator,HMC,potential}.py` (with the local
`_jaxshim` standing in for the absent third-party `jax`) and records inputs and
outputs of its hot path as small `.npz` files.  Only those `.npz` files travel
(to git and to the GPU box); neither the reference nor this script's imports
are needed to *run* the tests.

Run:  python tests/golden/gen_golden.py      (from anywhere)

Every potential/gradient handed to the reference is an explicit NumPy closed
form (the reference's tests use jax autodiff, which is unavailable offline):
  std / diagonal Gaussian   U = 0.5*sum(prec*(q-mu)^2) + c      dU = prec*(q-mu)
  dense Gaussian            U = 0.5*(q-mu).P.(q-mu) + c         dU = P.(q-mu)
  harmonic                  U = reference harmonicPotentialND   dU = k*q
  Rosenbrock                U = sum_i [b(q_{i+1}-q_i^2)^2 + (a-q_i)^2] * (1/s)
Fixture list mirrors SURVEY.md section 8c (G1..G10) plus two extras (G11, G12).
"""
import io
import os
import sys
import contextlib

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PBBI_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "_jaxshim"), os.path.join(REF, "src")]

import numpy as np  # noqa: E402
from scipy.constants import k as kB  # noqa: E402

from ensemble import Ensemble  # noqa: E402  (reference)
from HMC import HMC  # noqa: E402  (reference)
from integrator import Integrator, Leapfrog, StormerVerlet  # noqa: E402
from potential import harmonicPotentialND  # noqa: E402


# ----------------------------------------------------------------- potentials
def gauss_diag(mu, prec, const=0.0):
    mu = np.asarray(mu, float)
    prec = np.asarray(prec, float)

    def U(q):
        x = q - mu
        return 0.5 * np.dot(prec * x, x) + const

    def dU(q):
        return prec * (q - mu)

    return U, dU


def gauss_dense(mu, P, const=0.0):
    mu = np.asarray(mu, float)
    P = np.asarray(P, float)

    def U(q):
        x = q - mu
        return 0.5 * np.dot(x, P @ x) + const

    def dU(q):
        return P @ (q - mu)

    return U, dU


def harmonic(k):
    k = np.asarray(k, float)
    return (lambda q: harmonicPotentialND(q, k)), (lambda q: k * q)


def rosenbrock(a=1.0, b=100.0, s=20.0):
    inv_s = 1.0 / s  # the build's definition scales by the pre-computed reciprocal

    def U(q):
        t = q[1:] - q[:-1] ** 2
        return (np.sum(b * t * t) + np.sum((a - q[:-1]) ** 2)) * inv_s

    def dU(q):
        g = np.zeros_like(q)
        t = q[1:] - q[:-1] ** 2
        g[:-1] += (-4.0 * b * q[:-1] * t - 2.0 * (a - q[:-1])) * inv_s
        g[1:] += (2.0 * b * t) * inv_s
        return g

    return U, dU


def dense_precision(D, seed=0):
    """SURVEY 8d recipe: Sigma = A A^T / D + I, A = RandomState(seed) normals,
    P = sym(inv Sigma)."""
    A = np.random.RandomState(seed).standard_normal((D, D))
    Sigma = A @ A.T / D + np.eye(D)
    P = np.linalg.inv(Sigma)
    return 0.5 * (P + P.T), Sigma


# ----------------------------------------------------------- reference runners
def run_getsamples(D, N, S, simulTime, stepSize, temperature, qStd, seed, U, dU,
                   method="Leapfrog", mass=None):
    """Run the reference HMC.getSamples and capture every intermediate."""
    rec = {"p_draw": [], "u": [], "ratio": []}
    np.random.seed(seed)
    ens = Ensemble(D, N)
    if mass is not None:
        ens.mass = np.asarray(mass, float)
    with contextlib.redirect_stdout(io.StringIO()):
        hmc = HMC(ens, simulTime, stepSize, None, potential=U, gradient=dU,
                  method=method)
    orig_setpos, orig_setmom = ens.setPosition, ens.setMomentum
    orig_ratio, orig_uniform = hmc.getWeightsRatio, np.random.uniform

    def setpos(qStd_):
        q = orig_setpos(qStd_)
        rec["q0"] = q.copy()
        return q

    def setmom(T_):
        p = orig_setmom(T_)
        rec["p_draw"].append(p.copy())
        return p

    def ratio(nq, np_, oq, op):
        with np.errstate(all="ignore"):
            r = orig_ratio(nq, np_, oq, op)
        rec["ratio"].append(r.copy())
        rec.setdefault("q_prop", []).append(nq.copy())
        rec.setdefault("p_prop", []).append((-np_).copy())  # un-negated
        return r

    def uniform(*a, **k):
        u = orig_uniform(*a, **k)
        rec["u"].append(np.array(u, copy=True))
        return u

    ens.setPosition, ens.setMomentum = setpos, setmom
    hmc.getWeightsRatio = ratio
    np.random.uniform = uniform
    try:
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            samples, momenta = hmc.getSamples(S, temperature, qStd)
    finally:
        np.random.uniform = orig_uniform
    u = np.stack(rec["u"])
    ratio_ = np.stack(rec["ratio"])
    with np.errstate(all="ignore"):
        mask = u > np.minimum(1, ratio_)
    return dict(
        D=D, N=N, S=S, simulTime=simulTime, stepSize=stepSize,
        temperature=temperature, qStd=qStd, seed=seed,
        numSteps=hmc.integrator.numSteps, method=method,
        mass=np.asarray(ens.mass, float),
        q0=rec["q0"], p_draw=np.stack(rec["p_draw"]), u=u, ratio=ratio_,
        reject_mask=mask, q_prop=np.stack(rec["q_prop"]),
        p_prop=np.stack(rec["p_prop"]),
        samples=samples, momenta=momenta,
    )


def run_integrate(cls, D, N, stepSize, finalTime, qStd, temperature, seed, dU,
                  mass=None, q_mean=0.0):
    np.random.seed(seed)
    ens = Ensemble(D, N)
    if mass is not None:
        ens.mass = np.asarray(mass, float)
    ens.setPosition(qStd)
    ens.q += q_mean
    ens.setMomentum(temperature)
    q0, p0 = ens.q.copy(), ens.p.copy()
    integ = cls(ens, stepSize, finalTime, dU)
    q, p = integ.integrate()
    assert q is ens.q and p is ens.p  # in place, aliased
    return dict(D=D, N=N, stepSize=stepSize, finalTime=finalTime, seed=seed,
                numSteps=integ.numSteps, mass=np.asarray(ens.mass, float),
                q0=q0, p0=p0, q=q.copy(), p=p.copy(), v=integ.v.copy())


def harmonic_analytic(q0, p0, mass, t, k):
    omega = np.sqrt(np.outer(k, 1.0 / mass))
    v0 = p0 / mass
    q = q0 * np.cos(omega * t) + v0 / omega * np.sin(omega * t)
    v = -omega * q0 * np.sin(omega * t) + v0 * np.cos(omega * t)
    return q, v * mass


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name:28s} {os.path.getsize(path)/1024:8.1f} KiB")


def main():
    T1 = 1.0 / kB  # kB*T == 1.0 exactly -> pStd = sqrt(mass)

    # G1 / G2: integrators on the 2-D harmonic oscillator, one period of dim 0
    k = np.array((2.0, 3.0))
    _, dUh = harmonic(k)
    period = 2 * np.pi / np.sqrt(k[0])
    for tag, cls in (("G1_leapfrog_harmonic", Leapfrog),
                     ("G2_stormerverlet_harmonic", StormerVerlet)):
        out = {"springConsts": k, "period": period}
        for h in (1e-1, 1e-2):
            r = run_integrate(cls, 2, 5, h, period, 10.0, T1, 7, dUh)
            qa, pa = harmonic_analytic(r["q0"], r["p0"], r["mass"], period, k)
            sfx = "_h%g" % h
            for key, val in r.items():
                out[key + sfx] = val
            out["q_analytic" + sfx], out["p_analytic" + sfx] = qa, pa
        save(tag, **out)

    # G3: config C1 exactly (1-D standard Gaussian, ensemble 32, 10 steps)
    U, dU = gauss_diag([0.0], [1.0])
    save("G3_getsamples_c1", **run_getsamples(1, 32, 20, 1.0, 0.1, T1, 1.0, 1234, U, dU))

    # G4: dense Gaussian, D = 8 and D = 128 (config C2's potential, small N)
    for D, N in ((8, 64), (128, 40)):  # N = 40: ragged w.r.t. 16-chain tiles
        P, Sigma = dense_precision(D)
        mu = np.zeros(D)
        const = 0.5 * np.linalg.slogdet(2 * np.pi * Sigma)[1]
        U, dU = gauss_dense(mu, P, const)
        r = run_getsamples(D, N, 3, 1.0, 0.1, T1, 1.0, 42, U, dU)
        save("G4_getsamples_dense_d%d" % D, precision=P, mean=mu, const=const, **r)

    # G4b: dense Gaussian with a non-zero mean (D = 16)
    P, Sigma = dense_precision(16, seed=3)
    mu = np.random.RandomState(5).standard_normal(16) * 2.0
    const = 0.5 * np.linalg.slogdet(2 * np.pi * Sigma)[1]
    U, dU = gauss_dense(mu, P, const)
    r = run_getsamples(16, 48, 3, 1.0, 0.1, T1, 1.5, 43, U, dU)
    save("G4b_getsamples_dense_mean_d16", precision=P, mean=mu, const=const, **r)

    # G5: Rosenbrock D = 32 (config C3's potential, small N)
    U, dU = rosenbrock()
    r = run_getsamples(32, 64, 3, 0.1, 0.01, T1, 0.5, 44, U, dU)
    save("G5_getsamples_rosenbrock_d32", a=1.0, b=100.0, s=20.0, **r)
    r = run_integrate(Leapfrog, 32, 64, 0.01, 0.1, 0.1, T1, 45, dU, q_mean=1.0)
    save("G5b_leapfrog_rosenbrock_d32", a=1.0, b=100.0, s=20.0, **r)

    # G6: rejection-heavy (large step): pins mask + momentum_hmc quirks
    U, dU = gauss_diag([0.0, 0.0], [1.0, 1.0])
    r = run_getsamples(2, 2000, 4, 3.0, 1.5, T1, 1.0, 46, U, dU)
    print("   G6 reject fraction:", r["reject_mask"].mean())
    save("G6_getsamples_rejects", **r)

    # G7: overflowing energies -> inf - inf = NaN ratio -> proposal accepted
    U, dU = gauss_diag([0.0], [1.0])
    r = run_getsamples(1, 64, 2, 1.0, 0.1, T1, 1e154, 47, U, dU)
    print("   G7 NaN ratios:", int(np.isnan(r["ratio"]).sum()), "of", r["ratio"].size,
          " rejects:", int(r["reject_mask"].sum()))
    save("G7_getsamples_nan", **r)

    # G8: non-unit masses (mass = 1..5 cycled)
    D, N = 3, 10
    mass = 1.0 + (np.arange(N) % 5)
    P, Sigma = dense_precision(D, seed=1)
    U, dU = gauss_dense(np.zeros(D), P, 0.0)
    r = run_getsamples(D, N, 4, 1.0, 0.1, T1, 1.0, 48, U, dU, mass=mass)
    save("G8_getsamples_mass", precision=P, mean=np.zeros(D), const=0.0, **r)
    _, dUh = harmonic(np.array((2.0, 3.0, 0.5)))
    for tag, cls in (("G8b_leapfrog_mass", Leapfrog), ("G8c_stormerverlet_mass", StormerVerlet)):
        r = run_integrate(cls, D, N, 0.05, 1.0, 2.0, T1, 49, dUh, mass=mass)
        save(tag, springConsts=np.array((2.0, 3.0, 0.5)), **r)

    # G9: numSteps = int(finalTime / stepSize) truncation table
    pairs = [(1.0, 0.1), (0.5, 0.05), (0.3, 0.1), (1, 0.1), (3.0, 1.5), (0.1, 0.01),
             (2 * np.pi / np.sqrt(2.0), 1e-2), (1.0, 0.3), (0.7, 0.1), (0.05, 0.1),
             (10.0, 0.001), (1.0, 1.0 / 3.0)]
    ens = Ensemble(1, 1)
    with contextlib.redirect_stdout(io.StringIO()):
        steps = [Integrator(ens, h, T, lambda q: q).numSteps for (T, h) in pairs]
    save("G9_numsteps", finalTime=np.array([p[0] for p in pairs], float),
         stepSize=np.array([p[1] for p in pairs], float), numSteps=np.array(steps))

    # G10: known answers held by the reference's own asserting tests
    ens = Ensemble(2, 10)
    ens.q[:, 0] = np.array([3.0, 4.0])
    pot33 = harmonicPotentialND(ens.q, np.array([2, 3]))
    e = Ensemble(4, 100)
    q1, p1, m1, w1 = e.particle(10)
    try:
        e.particle(101)
        idx_err = ""
    except IndexError as err:
        idx_err = str(err)
    save("G10_known_answers", harmonic_q=ens.q, harmonic_k=np.array([2, 3]),
         harmonic_U=pot33, particle_q=q1, particle_p=p1, particle_m=m1,
         particle_w=w1, index_error=np.array(idx_err))

    # G11: the reference's test_HMC.py::test2 set-up (T = 300 K, mean (5,5))
    mu = np.ones(2) * 5
    cov = np.array([[4.0, -3.0], [-3.0, 4.0]])
    P = np.linalg.inv(cov)
    P = 0.5 * (P + P.T)
    const = 0.5 * np.linalg.slogdet(2 * np.pi * cov)[1]
    U, dU = gauss_dense(mu, P, const)
    r = run_getsamples(2, 50, 4, 0.5, 0.05, 300, 1, 50, U, dU)
    save("G11_getsamples_test2", precision=P, mean=mu, const=const, **r)

    # G12: getSamples with method="Stormer-Verlet"
    P, Sigma = dense_precision(4, seed=2)
    U, dU = gauss_dense(np.zeros(4), P, 0.0)
    r = run_getsamples(4, 32, 3, 1.0, 0.1, T1, 1.0, 51, U, dU, method="Stormer-Verlet")
    save("G12_getsamples_stormerverlet", precision=P, mean=np.zeros(4), const=0.0, **r)


if __name__ == "__main__":
    main()
