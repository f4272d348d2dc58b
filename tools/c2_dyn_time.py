#!/usr/bin/env python3
"""C2's shape (d=128 dense Gaussian, 65 536 chains, L=10) with per-chain trajectory lengths on the dense
MFMA kernel: microseconds per iteration of pbbi_hmc_run_dyn for PBBI_PER_CHAIN_STEPS, PBBI_UTURN_STOP
(L = 40) and both, with the mean number of steps taken."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib

D, N, S = 128, 65536, 20
A = np.random.RandomState(0).standard_normal((D, D))
Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0)
q = torch.randn((D, N), dtype=torch.float64, device="cuda")
samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
steps = torch.empty((S, N), dtype=torch.int32, device="cuda")
for name, flags, L in (("fixed length", 1, 10), ("per-chain steps", 1 | _lib.PER_CHAIN_STEPS, 10),
                       ("u-turn stop", 1 | _lib.UTURN_STOP, 40), ("both", 1 | _lib.PER_CHAIN_STEPS | _lib.UTURN_STOP, 40)):
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("pbbi_hmc_run_dyn", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), None, rej.data_ptr(),
                  None, steps.data_ptr(), N, N, 0.1, L, S, flags, 1, rep * S, 0, 1.0, None)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / S)
    print(f"{name}: {best:.1f} us per iteration, mean steps {float(steps.float().mean()):.2f}, "
          f"accept {1 - float(rej.float().mean()):.3f}", flush=True)
