#!/usr/bin/env python3
"""Config C5 (d=4096 dense precision, fp32, 8192 chains), 2 HMC iterations = 22 k_big_gemm launches:
a small fixed workload for rocprofv3 PMC passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib

D, N, L, S = 4096, 8192, 10, int(os.environ.get("C5_ITERS", "2"))
A = np.random.RandomState(0).standard_normal((D, D))
Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0, dtype="float32")
q = torch.randn((D, N), dtype=torch.float32, device="cuda")
samples = torch.empty((S, D, N), dtype=torch.float32, device="cuda")
mom = torch.empty((S, D, N), dtype=torch.float32, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
_lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
          rej.data_ptr(), None, N, N, 0.05, L, S, 1, 7, 0, 0, 1.0, None)
torch.cuda.synchronize()
print("ok", float(rej.float().mean()))
