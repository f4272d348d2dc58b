#!/usr/bin/env python3
"""Small chains on the chain-per-lane kernel (k_lane_hmc, D <= 16): microseconds per iteration of one
pbbi_hmc_run of 200 iterations, reference operation order and kick-drift-kick flag alike (the kernel has one
form).  A/B: PBBI_NO_LANE_FUSE=1 python tools/lane_fuse_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib

S, L, h = 200, 10, 0.1
for name, pot, D, N in (("standard Gaussian D=1, 32 chains (C1)", P.StandardGaussian(1), 1, 32),
                        ("harmonic D=3, 65536 chains", P.Harmonic(np.array([2.0, 3.0, 0.5])), 3, 65536),
                        ("diagonal Gaussian D=8, 262144 chains", P.GaussianDiag(np.zeros(8), prec=np.linspace(0.5, 2, 8), const=0.0), 8, 262144),
                        ("diagonal Gaussian D=16, 262144 chains", P.GaussianDiag(np.zeros(16), prec=np.linspace(0.5, 2, 16), const=0.0), 16, 262144),
                        ("Rosenbrock D=12, 262144 chains", P.Rosenbrock(12), 12, 262144)):
    q = torch.ones((D, N), dtype=torch.float64, device="cuda")
    samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
    mom = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
    rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
    hh = 0.02 if "Rosenbrock" in name else h
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
                  rej.data_ptr(), None, N, N, hh, L, S, 1, 1, rep * S, 0, 1.0, None)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / S)
    print(f"{name}: {best:.2f} us per iteration, {L * N / best * 1e6:.3g} step*chain/s, accept {1 - float(rej.float().mean()):.3f}",
          flush=True)
