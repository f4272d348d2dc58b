#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 passes over the streamed dense kernel (tools/run_profiles_dstream.sh):
one pbbi_hmc_run of 8 iterations at D = PBBI_TIME_D (default 256), 65 536 chains, L = 10 -- one fused launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib

D, N, L, S = int(os.environ.get("PBBI_TIME_D", 256)), 65536, 10, 8
A = np.random.RandomState(0).standard_normal((D, D))
Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0)
q = torch.randn((D, N), dtype=torch.float64, device="cuda")
samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
mom = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
for rep in range(2):
    _lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
              rej.data_ptr(), None, N, N, 0.1, L, S, 1, 1, rep * S, 0, 1.0, None)
torch.cuda.synchronize()
print("ok", float(rej.float().mean()))
