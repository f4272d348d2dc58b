"""ctypes front-end of the CPU oracle (oracle/pbbi_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of pbbi_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.  Parity pinning: PINNED against the reference's own
outputs (tests/golden/*.npz, tests/test_oracle_golden.py).

Potentials are described by plain dicts (`pot_*` helpers below) so that the
tests can hand the *same* parameters to the oracle and to the product's
descriptors.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

HARMONIC, GAUSS_DIAG, GAUSS_DENSE, ROSENBROCK, CUSTOM = 0, 1, 2, 3, 4
LEAPFROG, STORMER_VERLET = 0, 1
COMPAT_P_FROM_OLDQ = 1
BETA_ACCEPT = 4
DRAW_F64 = 32          # PBBI_DRAW_F64: double-precision Box-Muller (include/pbbi.h)
STREAM_DRAW_F64 = 0x100  # the same as a bit of the stream argument of philox_normal
STREAM_MOMENTUM, STREAM_POSITION, STREAM_UNIFORM = 0, 1, 2
METHODS = {"Leapfrog": LEAPFROG, "Stormer-Verlet": STORMER_VERLET}


class _Pot(C.Structure):
    _fields_ = [("kind", C.c_int), ("D", C.c_int), ("mean", C.c_void_p), ("prec", C.c_void_p),
                ("cst", C.c_double), ("a", C.c_double), ("b", C.c_double), ("s", C.c_double),
                ("user_U", C.c_void_p), ("user_grad", C.c_void_p)]


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc, a few seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "pbbi_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.oracle_get_max_threads.restype = C.c_int
    return _LIB


def set_threads(n):
    lib().oracle_set_threads(C.c_int(int(n)))


def max_threads():
    return int(lib().oracle_get_max_threads())


# ------------------------------------------------------------------ potentials
def pot_harmonic(springConsts):
    return dict(kind=HARMONIC, prec=np.ascontiguousarray(springConsts, dtype=np.float64))


def pot_gauss_diag(mean, prec, const=0.0):
    return dict(kind=GAUSS_DIAG, mean=np.ascontiguousarray(mean, dtype=np.float64),
                prec=np.ascontiguousarray(prec, dtype=np.float64), cst=float(const))


def pot_gauss_dense(mean, precision, const=0.0):
    P = np.ascontiguousarray(precision, dtype=np.float64)
    assert P.ndim == 2 and P.shape[0] == P.shape[1]
    return dict(kind=GAUSS_DENSE, mean=np.ascontiguousarray(mean, dtype=np.float64), prec=P,
                cst=float(const), D=P.shape[0])


def pot_rosenbrock(D, a=1.0, b=100.0, s=20.0):
    return dict(kind=ROSENBROCK, D=int(D), a=float(a), b=float(b), s=float(s))


_CUSTOM_LIBS = {}


def pot_custom(source, D, params=()):
    """The user-potential source of physicsbasedbayesianinference_amd/custom.py compiled for the
    HOST (g++ -O2 -ffp-contract=off, T = double, q / g plain pointers) and called chain by chain
    by the oracle's integrators -- the CPU side of the parity tests of CustomPotential."""
    import hashlib
    # `source` is the text that goes inside namespace user: the tests hand over
    # custom.complete_source(user_source), the same text the device plugin is built from
    tu = ('#include <cmath>\n#include <cstdint>\n#include <type_traits>\n#include <utility>\n'
          'using namespace std;\nusing T = double;\n'
          '#define PBBI_FN static inline\nnamespace user {\n' + source + '\n}\n'
          'extern "C" double pbbi_user_U(const double* q, int D, const double* prm) '
          '{ return user::potential(q, D, prm); }\n'
          'extern "C" void pbbi_user_grad(const double* q, double* g, int D, const double* prm) '
          '{ user::gradient(q, g, D, prm); }\n')
    key = hashlib.sha256(tu.encode()).hexdigest()[:20]
    if key not in _CUSTOM_LIBS:
        d = os.path.join(_HERE, "_custom")
        os.makedirs(d, exist_ok=True)
        so, src = os.path.join(d, f"user_{key}.so"), os.path.join(d, f"user_{key}.cpp")
        if not os.path.exists(so):
            with open(src, "w") as f:
                f.write(tu)
            tmp = so + f".tmp{os.getpid()}"
            subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17",
                                   src, "-o", tmp])
            os.replace(tmp, so)
        _CUSTOM_LIBS[key] = C.CDLL(so)
    L = _CUSTOM_LIBS[key]
    return dict(kind=CUSTOM, D=int(D), prec=np.ascontiguousarray(params, dtype=np.float64).ravel(),
                user_U=C.cast(L.pbbi_user_U, C.c_void_p).value,
                user_grad=C.cast(L.pbbi_user_grad, C.c_void_p).value)


def _cpot(pot):
    D = pot.get("D")
    if D is None:
        D = int(pot["prec"].shape[0])
    keep = []
    mean = pot.get("mean")
    prec = pot.get("prec")
    for arr in (mean, prec):
        if arr is not None:
            assert arr.dtype == np.float64 and arr.flags.c_contiguous
            keep.append(arr)
    st = _Pot(pot["kind"], int(D),
              mean.ctypes.data if mean is not None else None,
              prec.ctypes.data if prec is not None else None,
              pot.get("cst", 0.0), pot.get("a", 1.0), pot.get("b", 100.0), pot.get("s", 20.0),
              pot.get("user_U"), pot.get("user_grad"))
    return st, keep, int(D)


def _dn(arr):
    assert arr.dtype == np.float64 and arr.ndim == 2 and arr.flags.c_contiguous, \
        "oracle arrays are (D, N) C-order float64"
    return arr.shape


def _ptr(arr):
    return None if arr is None else C.c_void_p(arr.ctypes.data)


def _mass(mass, N):
    if mass is None:
        return None
    m = np.ascontiguousarray(mass, dtype=np.float64)
    assert m.shape == (N,)
    return m


# ------------------------------------------------------------------ entry points
def potential(pot, q, want_grad=False):
    st, keep, D = _cpot(pot)
    q = np.ascontiguousarray(q, dtype=np.float64)
    single = q.ndim == 1
    q2 = q.reshape(D, -1) if single else q
    assert q2.shape[0] == D
    N = q2.shape[1]
    q2 = np.ascontiguousarray(q2)
    U = np.empty(N)
    g = np.empty_like(q2) if want_grad else None
    rc = lib().oracle_potential(C.byref(st), _ptr(q2), C.c_int64(N), C.c_int64(N), _ptr(U), _ptr(g))
    assert rc == 0
    if single:
        return (U[0], g[:, 0]) if want_grad else U[0]
    return (U, g) if want_grad else U


def integrate(pot, method, q, p, mass, h, L):
    """In-place integrate() on (D, N) arrays; returns v (Integrator.v)."""
    st, keep, D = _cpot(pot)
    Dq, N = _dn(q)
    assert (Dq, N) == _dn(p) and Dq == D
    m = _mass(mass, N)
    v = np.empty_like(q)
    rc = lib().oracle_integrate(C.byref(st), C.c_int(METHODS.get(method, method)), _ptr(q), _ptr(p),
                                _ptr(m), C.c_int64(N), C.c_int64(N), C.c_double(h), C.c_int(L),
                                _ptr(v))
    assert rc == 0
    return v


def weights(pot, q, p, mass=None):
    st, keep, D = _cpot(pot)
    _, N = _dn(q)
    m = _mass(mass, N)
    w, H = np.empty(N), np.empty(N)
    rc = lib().oracle_weights(C.byref(st), _ptr(q), _ptr(p), _ptr(m), C.c_int64(N), C.c_int64(N),
                              _ptr(w), _ptr(H))
    assert rc == 0
    return w, H


def weights_ratio(pot, newQ, newP, oldQ, oldP, mass=None):
    st, keep, D = _cpot(pot)
    _, N = _dn(newQ)
    m = _mass(mass, N)
    r = np.empty(N)
    with np.errstate(all="ignore"):
        rc = lib().oracle_weights_ratio(C.byref(st), _ptr(newQ), _ptr(newP), _ptr(oldQ), _ptr(oldP),
                                        _ptr(m), C.c_int64(N), C.c_int64(N), _ptr(r))
    assert rc == 0
    return r


def hmc_iter_dyn(pot, q, p, u, mass, h, L, steps_in=None, uturn=False, compat=COMPAT_P_FROM_OLDQ, beta=1.0):
    """hmc_iter (Leapfrog) with per-chain trajectory lengths: steps_in (N int32, None = L for every
    chain) and / or stop at the first U-turn.  Returns (ratio, reject_mask, steps_taken)."""
    st, keep, D = _cpot(pot)
    Dq, N = _dn(q)
    assert _dn(p) == (Dq, N) and Dq == D
    u = np.ascontiguousarray(u, dtype=np.float64)
    m = _mass(mass, N)
    si = None if steps_in is None else np.ascontiguousarray(steps_in, dtype=np.int32)
    ratio, rej, steps = np.empty(N), np.empty(N, dtype=np.uint8), np.empty(N, dtype=np.int32)
    rc = lib().oracle_hmc_iter_dyn(C.byref(st), C.c_int(LEAPFROG), _ptr(q), _ptr(p), _ptr(u), _ptr(m),
                                   C.c_int64(N), C.c_int64(N), C.c_double(h), C.c_int(L), C.c_int(compat),
                                   C.c_double(beta), _ptr(si), C.c_int(1 if uturn else 0), _ptr(steps),
                                   _ptr(ratio), _ptr(rej))
    assert rc == 0
    return ratio, rej.astype(bool), steps


def hmc_iter_gist(pot, q, p, u_acc, u_len, mass, h, Lmax, compat=COMPAT_P_FROM_OLDQ, beta=1.0):
    """One GIST (self-tuning no-U-turn) iteration in place on q, p (oracle_hmc_iter_gist).  Returns
    (ratio, reject_mask, tau) with tau (3, N) = forward U-turn count, drawn length, backward U-turn count."""
    st, keep, D = _cpot(pot)
    Dq, N = _dn(q)
    assert Dq == D and p.shape == q.shape
    m = _mass(mass, N)
    u_acc = np.ascontiguousarray(u_acc, dtype=np.float64)
    u_len = np.ascontiguousarray(u_len, dtype=np.float64)
    ratio, rej, tau = np.empty(N), np.empty(N, dtype=np.uint8), np.empty((3, N), dtype=np.int32)
    rc = lib().oracle_hmc_iter_gist(C.byref(st), _ptr(q), _ptr(p), _ptr(u_acc), _ptr(u_len), _ptr(m),
                                    C.c_int64(N), C.c_int64(N), C.c_double(h), C.c_int(Lmax), C.c_int(compat),
                                    C.c_double(beta), _ptr(ratio), _ptr(rej), tau.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return ratio, rej.astype(bool), tau


def replica_exchange(pot, q, Nr, R, betas, parity, seed, it, chain0=0):
    """oracle_replica_exchange in place on q (D, R*Nr); returns the ((R-1), Nr) bool array of swaps (rows of the
    other parity False)."""
    st, keep, D = _cpot(pot)
    Dq, N = _dn(q)
    assert Dq == D and N == R * Nr
    betas = np.ascontiguousarray(betas, dtype=np.float64)
    sw = np.zeros((max(R - 1, 0), Nr), dtype=np.uint8)
    rc = lib().oracle_replica_exchange(C.byref(st), _ptr(q), C.c_int64(Nr), C.c_int(R), C.c_int64(N), _ptr(betas),
                                       C.c_int(parity), C.c_uint64(seed), C.c_uint64(it), C.c_uint64(chain0),
                                       sw.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return sw.astype(bool)


def philox_steps_uniform(seed, it, chain0, N):
    out = np.empty(N)
    lib().oracle_philox_steps_uniform(C.c_uint64(seed), C.c_uint64(it), C.c_uint64(chain0), C.c_int64(N), _ptr(out))
    return out


def philox_steps(seed, it, chain0, N, L):
    out = np.empty(N, dtype=np.int32)
    lib().oracle_philox_steps(C.c_uint64(seed), C.c_uint64(it), C.c_uint64(chain0), C.c_int64(N), C.c_int(L),
                              _ptr(out))
    return out


def hmc_iter(pot, method, q, p, u, mass, h, L, compat=COMPAT_P_FROM_OLDQ, beta=1.0):
    """One getSamples iteration, in place on q (state) and p (drawn momentum).
    Returns (ratio, reject_mask).  beta != 1: the build's PBBI_BETA_ACCEPT accept test."""
    st, keep, D = _cpot(pot)
    Dq, N = _dn(q)
    assert _dn(p) == (Dq, N) and Dq == D
    u = np.ascontiguousarray(u, dtype=np.float64)
    assert u.shape == (N,)
    m = _mass(mass, N)
    ratio = np.empty(N)
    rej = np.empty(N, dtype=np.uint8)
    rc = lib().oracle_hmc_iter_beta(C.byref(st), C.c_int(METHODS.get(method, method)), _ptr(q), _ptr(p),
                                    _ptr(u), _ptr(m), C.c_int64(N), C.c_int64(N), C.c_double(h),
                                    C.c_int(L), C.c_int(compat), C.c_double(beta), _ptr(ratio), _ptr(rej))
    assert rc == 0
    return ratio, rej.astype(bool)


def philox_raw(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oracle_philox_raw(c, k, o)
    return [int(x) for x in o]


def philox_normal(seed, stream, it, chain0, D, N, scale=1.0):
    out = np.empty((D, N))
    sc = None
    s0 = 1.0
    if np.ndim(scale) == 0:
        s0 = float(scale)
    else:
        sc = np.ascontiguousarray(scale, dtype=np.float64)
        assert sc.shape == (N,)
    lib().oracle_philox_normal(C.c_uint64(seed), C.c_int(stream), C.c_uint64(it), C.c_uint64(chain0),
                               C.c_int(D), C.c_int64(N), C.c_int64(N), C.c_double(s0), _ptr(sc),
                               _ptr(out))
    return out


def philox_uniform(seed, it, chain0, N):
    out = np.empty(N)
    lib().oracle_philox_uniform(C.c_uint64(seed), C.c_uint64(it), C.c_uint64(chain0), C.c_int64(N),
                                _ptr(out))
    return out


def hmc_run_philox(pot, method, q, mass, h, L, S, seed, iter0=0, chain0=0, kT=1.0,
                   compat=COMPAT_P_FROM_OLDQ, want_momenta=True):
    """S iterations with the Philox contract.  q (D,N) is the state, updated in
    place.  Returns samples (S,D,N), momenta (S,D,N)|None, reject (S,N) bool,
    ratio (S,N)."""
    st, keep, D = _cpot(pot)
    Dq, N = _dn(q)
    assert Dq == D
    m = _mass(mass, N)
    samples = np.empty((S, D, N))
    momenta = np.empty((S, D, N)) if want_momenta else None
    rej = np.empty((S, N), dtype=np.uint8)
    ratio = np.empty((S, N))
    rc = lib().oracle_hmc_run_philox(C.byref(st), C.c_int(METHODS.get(method, method)), _ptr(q),
                                     _ptr(m), C.c_int64(N), C.c_int64(N), C.c_double(h), C.c_int(L),
                                     C.c_int(S), C.c_int(compat), C.c_uint64(seed),
                                     C.c_uint64(iter0), C.c_uint64(chain0), C.c_double(kT),
                                     _ptr(samples), _ptr(momenta), _ptr(rej), _ptr(ratio))
    assert rc == 0
    return samples, momenta, rej.astype(bool), ratio


def get_samples_numpy_stream(pot, method, D, N, S, simulTime, stepSize, temperature, qStd, seed,
                             mass=None, compat=COMPAT_P_FROM_OLDQ):
    """Restatement of HMC.getSamples (src/HMC.py:123-183) on the legacy NumPy
    RandomState stream: q0 = normals*qStd, then per iteration D*N normals times
    sqrt(mass*kB*T) and N uniforms (src/ensemble.py:72-74,88-91).
    Returns dict with samples/momenta as (D,N,S) like the reference."""
    from scipy.constants import k as kB
    rs = np.random.RandomState(seed)
    L = int(simulTime / stepSize)  # src/integrator.py:51
    m = np.ones(N) if mass is None else np.asarray(mass, float)
    q = np.ascontiguousarray(rs.standard_normal((D, N)) * qStd)
    pStd = np.sqrt(m * kB * temperature)
    samples = np.zeros((D, N, S))
    momenta = np.zeros((D, N, S))
    ratios, masks = [], []
    for i in range(S):
        p = np.ascontiguousarray(rs.standard_normal((D, N)) * pStd)
        u = rs.uniform(size=N)
        r, rej = hmc_iter(pot, method, q, p, u, m, stepSize, L, compat)
        samples[:, :, i] = q
        momenta[:, :, i] = p
        ratios.append(r)
        masks.append(rej)
    return dict(samples=samples, momenta=momenta, ratio=np.stack(ratios),
                reject_mask=np.stack(masks), numSteps=L)
