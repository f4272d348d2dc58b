"""Throughput sweep over dimensions for each kernel family (in-kernel draws, L = 10): finds
dimension-dependent cliffs (padding, guards).  python tools/sweep_dims.py [kdk|exact]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicsbasedbayesianinference_amd as P  # noqa: E402
from physicsbasedbayesianinference_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "kdk"
flags = _lib.COMPAT_P_FROM_OLDQ | (_lib.KDK_FMA if mode == "kdk" else 0)


def run(name, pot, D, N, h, K=60, W=200, L=10):
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.empty((D, N), dtype=torch.float64, device="cuda")
    _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, D, N, N, 0.1, None, _lib.F64, 0,
              q.data_ptr(), stream)
    q += 1.0
    S = max(K, W)
    samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
    mom = torch.empty((S, D, N), dtype=torch.float64, device="cuda")

    def go(s, it0):
        _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                  mom.data_ptr(), None, None, N, N, h, L, s, flags, 7, it0, 0, 1.0, stream)
    go(W, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(K, W)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    bytes_it = (4 * D * 8 + 9) * N
    print(json.dumps({"kernel": name, "mode": mode, "D": D, "chains": N, "rate": K * L * N / t,
                      "us_per_iter": t / K * 1e6, "hbm_frac": bytes_it / (t / K) / 8e12,
                      "elem_steps_per_s": K * L * N * D / t}), flush=True)


rs = np.random.RandomState(0)
for D in (8, 12, 16, 24, 32, 48, 64):
    run("diag", P.GaussianDiag(rs.standard_normal(D), prec=rs.uniform(0.5, 2, D), const=0.0), D, 131072, 0.1)
for D in (20, 24, 32):
    run("rosenbrock", P.Rosenbrock(D), D, 131072, 0.01)
for D in (24, 32, 48, 64, 100, 128):
    A = rs.standard_normal((D, D))
    Pm = np.linalg.inv(A @ A.T / D + np.eye(D))
    run("dense", P.GaussianDense(None, precision=0.5 * (Pm + Pm.T), const=0.0), D, 65536, 0.1, K=30, W=60)
