#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 PMC passes (tools/run_profiles.sh):
  1. CALIBRATION: k_dense_eval over the C2-size state reads q exactly once (D*N*8 bytes,
     the same 8 B/lane, 4 x 128 B-segment access pattern as the HMC kernel) and writes N*8.
  2. the C2 HMC iterations themselves (k_dense_hmc): one run of 9 = a first launch + one fused launch of 8.
FETCH_SIZE / WRITE_SIZE of (1) against its known byte count calibrate the counters for (2)
(MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for 16 B/lane streams)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib

D, N, L, S = 128, 65536, 10, 9  # (tools/summarize_profiles_r02.py: C2_ITERS)
A = np.random.RandomState(0).standard_normal((D, D))
Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0)
q = torch.randn((D, N), dtype=torch.float64, device="cuda")
U = torch.empty((N,), dtype=torch.float64, device="cuda")
for _ in range(4):
    _lib.call("pbbi_potential_eval", pot.handle, q.data_ptr(), N, N, U.data_ptr(), None, None)
samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
mom = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
_lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
          rej.data_ptr(), None, N, N, 0.1, L, S, 1, 1, 0, 0, 1.0, None)
torch.cuda.synchronize()
print("ok", float(U.sum()), float(rej.float().mean()))
