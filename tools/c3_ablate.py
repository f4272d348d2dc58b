#!/usr/bin/env python3
"""C3 (Rosenbrock d=32, 262144 chains) launch time against the number of leapfrog steps and with /
without the momentum slab: separates the fixed part (loads, draw, energies, stores) from the
per-step vector work, to see whether they add or overlap.  usage: tools/c3_ablate.py [--exact]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import physicsbasedbayesianinference_amd as P  # noqa: E402
from physicsbasedbayesianinference_amd import _lib  # noqa: E402

d, N, h = 32, 262144, 0.01
pot = P.Rosenbrock(d)
stream = torch.cuda.current_stream().cuda_stream
K = 200
samples = torch.empty((K, d, N), dtype=torch.float64, device="cuda")
momenta = torch.empty((K, d, N), dtype=torch.float64, device="cuda")
reject = torch.empty((K, N), dtype=torch.uint8, device="cuda")
res = []
for exact in (False, True):
    flags = _lib.COMPAT_P_FROM_OLDQ | (0 if exact else _lib.KDK_FMA)
    for with_p in (True, False):
        for L in (0, 1, 2, 5, 10, 20):
            q = torch.empty((d, N), dtype=torch.float64, device="cuda")
            _lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 0.1, None, _lib.F64, 0,
                      q.data_ptr(), stream)
            q += 1.0

            def run(S, it0):
                _lib.call("pbbi_hmc_run", pot.handle, _lib.LEAPFROG, q.data_ptr(), None, samples.data_ptr(),
                          momenta.data_ptr() if with_p else None, reject.data_ptr(), None, N, N, h, L, S,
                          flags, 7, it0, 0, 1.0, stream)
            for r in range(3):
                run(K, r * K)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(K, 3 * K); e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / K
            res.append(dict(exact=exact, with_p=with_p, L=L, us=us))
            print(f"exact={exact} p_out={with_p} L={L:2d}: {us:7.2f} us/launch", flush=True)
print(json.dumps(res))
