#!/usr/bin/env python3
"""C2 (d=128 dense Gaussian, 65 536 chains, fp64): microseconds per iteration of pbbi_hmc_run for the
trajectory lengths given on the command line (default 10), after a settling run.  A/B tool:
PBBI_NO_CARRY=1 python tools/c2_time.py 9 10 11   (PBBI_TIME_D / _N / _S: another dimension, ensemble, run length)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib

METHOD = int(os.environ.get("PBBI_TIME_METHOD", 0))  # 0 Leapfrog, 1 Stormer-Verlet
D, N, S = (int(os.environ.get(k, v)) for k, v in (("PBBI_TIME_D", 128), ("PBBI_TIME_N", 65536), ("PBBI_TIME_S", 100)))
A = np.random.RandomState(0).standard_normal((D, D))
Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0)
q = torch.randn((D, N), dtype=torch.float64, device="cuda")
samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
mom = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")


def run(L, it0):
    _lib.call("pbbi_hmc_run", pot.handle, METHOD, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
              rej.data_ptr(), None, N, N, 0.1, L, S, 1, 1, it0, 0, 1.0, None)


for rep in range(3):
    run(10, rep * S)
torch.cuda.synchronize()
for L in [int(x) for x in sys.argv[1:]] or [10]:
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(L, 1000 + rep * S); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / S)
    print(f"L={L}: {best:.1f} us per iteration, accept {1 - float(rej.float().mean()):.3f}", flush=True)
