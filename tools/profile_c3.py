#!/usr/bin/env python3
"""Config C3 (Rosenbrock d=32, 262144 chains), 32 HMC iterations = two fused launches of k_ros2_hmc
(16 iterations each, the chain in registers in between): a small fixed workload for rocprofv3 PMC
passes.  C3_EXACT=1: the reference-order form."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib

D, N, L, S = int(os.environ.get("C3_D", 32)), 262144, 10, 32  # (C3_D: the multi-lane kernels at D = 64 / 128, same workload)
FLAGS = _lib.COMPAT_P_FROM_OLDQ | (0 if os.environ.get('C3_EXACT') == '1' else _lib.KDK_FMA)
pot = P.Rosenbrock(D)
q = 1.0 + 0.1 * torch.randn((D, N), dtype=torch.float64, device="cuda")
samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
mom = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
_lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
          rej.data_ptr(), None, N, N, 0.01, L, S, FLAGS, 1, 0, 0, 1.0, None)
torch.cuda.synchronize()
print("ok", float(rej.float().mean()))
