"""Randomised cross-check (GPU): every potential kind at random D, N, L, masses -- the PBBI_KDK_FMA
kernels against the reference-operation-order kernels on the same inputs (pbbi_hmc_iter with
host-supplied draws): masks equal, q / p within 1e-11.  python tools/fuzz_kdk_vs_exact.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicsbasedbayesianinference_amd as P  # noqa: E402
from physicsbasedbayesianinference_amd import _lib as lib  # noqa: E402
from physicsbasedbayesianinference_amd._device import as_device, empty, stream_ptr, to_numpy  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad = 0.0, 0
for it in range(cases):
    kind = rs.choice(["diag", "harmonic", "rosenbrock"])
    D = int(rs.choice([rs.randint(1, 17), rs.randint(17, 33), rs.randint(33, 65), rs.randint(65, 129), rs.randint(129, 257)]))
    N = int(rs.choice([1, rs.randint(2, 64), rs.randint(64, 700)]))
    L = int(rs.choice([0, 1, 2, 7, 20]))
    mass = bool(rs.randint(2))
    method = int(rs.randint(2))  # 0 Leapfrog, 1 Stormer-Verlet
    if kind == "diag":
        pot = P.GaussianDiag(rs.standard_normal(D), prec=rs.uniform(0.5, 2, D), const=0.1)
        h, sc = 0.3, 1.0
    elif kind == "harmonic":
        pot = P.Harmonic(rs.uniform(0.5, 2, D))
        h, sc = 0.3, 1.0
    else:
        pot = P.Rosenbrock(D)
        h, sc = 0.02, 0.3
    q, p, u = rs.standard_normal((D, N)) * sc, rs.standard_normal((D, N)), rs.uniform(size=N)
    u[::3] = 1.5
    m = as_device(1.0 + (np.arange(N) % 3) * 0.5, 0, np.float64) if mass else None
    qd, pd, ud = (as_device(x, 0, np.float64) for x in (q, p, u))
    res = []
    for flags in (lib.COMPAT_P_FROM_OLDQ, lib.COMPAT_P_FROM_OLDQ | lib.KDK_FMA):
        qo, po = empty((D, N), np.float64, 0), empty((D, N), np.float64, 0)
        ro, rj = empty((N,), np.float64, 0), empty((N,), np.uint8, 0)
        lib.call("pbbi_hmc_iter", pot.handle, method, qd.data_ptr(), pd.data_ptr(), ud.data_ptr(),
                 m.data_ptr() if mass else None, qo.data_ptr(), po.data_ptr(), ro.data_ptr(), rj.data_ptr(),
                 N, N, h, L, flags, stream_ptr(0))
        torch.cuda.synchronize()
        res.append((to_numpy(qo), to_numpy(po), to_numpy(ro), to_numpy(rj)))
    (q0, p0, r0, j0), (q1, p1, r1, j1) = res
    scale = max(1.0, np.abs(q0).max(), np.abs(p0).max())
    err = max(np.abs(q0 - q1).max(), np.abs(p0 - p1).max()) / scale
    lr = np.abs(np.log(r0) - np.log(r1)).max() if N else 0.0
    ok = np.array_equal(j0, j1) and err < 1e-11 and lr < 1e-8
    worst = max(worst, err)
    if not ok:
        bad += 1
        print("MISMATCH", kind, "method", method, "D", D, "N", N, "L", L, "mass", mass, "err", err, "log-ratio", lr,
              "masks differ", int((j0 != j1).sum()), flush=True)
print(f"{cases} cases, {bad} mismatches, worst scaled error {worst:.2e}")
sys.exit(1 if bad else 0)
