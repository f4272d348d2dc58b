#!/usr/bin/env python3
"""Wall time of one C2-size momentum draw (128 x 65 536 normals) through libpbbi_host.so, bit-checked against
np.random; PBBI_HOST_LIB=<path> times another build of the library (A/B), argv = thread counts."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from physicsbasedbayesianinference_amd import _hoststream as hs

if os.environ.get("PBBI_HOST_LIB"):
    hs.LIB_PATH = os.environ["PBBI_HOST_LIB"]
n = int(os.environ.get("PBBI_HOST_N", 128 * 65536))
np.random.seed(5)
ref = np.random.standard_normal(n)
if os.environ.get("PBBI_HOST_GEN"):   # generator threads (MT19937 jump-ahead); 1 = the sequential generator
    import ctypes
    hs._load().pbbi_host_debug_set_gen.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
    hs._load().pbbi_host_debug_set_gen(int(os.environ["PBBI_HOST_GEN"]), 8192, 1024)
for thr in [int(a) for a in sys.argv[1:]] or [1, 16]:
    hs._load().pbbi_host_set_threads(thr)
    times = []
    for rep in range(7):
        np.random.seed(5)
        t = time.perf_counter()
        x = hs.standard_normal(n)
        times.append(time.perf_counter() - t)
    print(f"{hs.LIB_PATH.split('/')[-1]} n {n} threads {thr}: {min(times) * 1e9 / n:.3f} ns per normal, best {min(times) * 1e3:.2f} ms, median {sorted(times)[3] * 1e3:.2f} ms, "
          f"bit-exact {np.array_equal(x, ref)}", flush=True)
