#!/usr/bin/env python3
"""Where do the first milliseconds of C2 go?  Times consecutive 5-iteration pbbi_hmc_run launches from a cold
start with HIP events, (a) writing sample slabs that were never touched, (b) slabs zeroed beforehand,
(c) re-using ONE 5-iteration slab set.  usage: tools/ramp_probe.py [a|b|c]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import physicsbasedbayesianinference_amd as P
from physicsbasedbayesianinference_amd import _lib

mode = sys.argv[1] if len(sys.argv) > 1 else "a"
D, N, L, C, NL = 128, 65536, 10, 5, 24
A = np.random.RandomState(0).standard_normal((D, D))
Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0)
q = torch.empty((D, N), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
_lib.call("pbbi_philox_normal", 42, _lib.STREAM_POSITION, 0, 0, D, N, N, 1.0, None, _lib.F64, 0, q.data_ptr(), st)
S = C if mode == "c" else C * NL
samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
momenta = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
reject = torch.empty((S, N), dtype=torch.uint8, device="cuda")
if mode == "b":
    samples.zero_(); momenta.zero_()
torch.cuda.synchronize()
time.sleep(0.5)  # idle before the first launch, like a fresh process
ev = [torch.cuda.Event(enable_timing=True) for _ in range(NL + 1)]
ev[0].record()
for k in range(NL):
    o = 0 if mode == "c" else k * C
    _lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples[o].data_ptr(), momenta[o].data_ptr(),
              reject[o].data_ptr(), None, N, N, 0.1, L, C, 1, 42, k * C, 0, 1.0, st)
    ev[k + 1].record()
torch.cuda.synchronize()
print(mode, " ".join(f"{ev[k].elapsed_time(ev[k + 1]) / C * 1e3:.0f}" for k in range(NL)), "us per iteration per launch")
