"""Throughput of user-defined potentials (csrc/pbbi_custom.h) on one GPU: the quartic chain and
Bayesian logistic regression of tests/custom_sources.py, in-kernel draws, L = 10."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from custom_sources import LOGISTIC, QUARTIC, logistic_problem  # noqa: E402

import physicsbasedbayesianinference_amd as P  # noqa: E402
from physicsbasedbayesianinference_amd import _lib  # noqa: E402
from physicsbasedbayesianinference_amd.custom import CustomPotential  # noqa: E402


def run(name, pot, D, N, h, L=10, K=32, W=16):
    stream = torch.cuda.current_stream().cuda_stream
    q = torch.zeros((D, N), dtype=torch.float64, device="cuda")
    S = max(K, W)
    samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
    rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")

    def go(s, it0):
        _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                  None, rej.data_ptr(), None, N, N, h, L, s, _lib.KDK_FMA, 7, it0, 0, 1.0, stream)
    go(W, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(K, W)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(json.dumps({"potential": name, "D": D, "chains": N, "L": L,
                      "leapfrog_steps_x_chains_per_s": K * L * N / t, "ms_per_iteration": t / K * 1e3,
                      "accept_rate": 1.0 - float(rej[:K].float().mean().item())}))


if __name__ == "__main__":
    run("quartic chain (registers)", CustomPotential(16, QUARTIC, [1.0, 0.5]), 16, 262144, 0.05)
    run("quartic chain (registers, kick-drift-kick form)", CustomPotential(32, QUARTIC, [1.0, 0.5]), 32, 262144, 0.05)
    run("quartic chain (workspace)", CustomPotential(48, QUARTIC, [1.0, 0.5]), 48, 131072, 0.05)
    run("quartic chain", CustomPotential(128, QUARTIC, [1.0, 0.5]), 128, 65536, 0.05)
    X, y, lam, prm = logistic_problem(M=256, D=16)
    run("logistic regression M=256", CustomPotential(16, LOGISTIC, prm), 16, 65536, 0.02)
