#!/usr/bin/env python3
"""Why is the FIRST timed run of a second bench_c2 call in one process slow on the host side?  Times the phases."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from physicsbasedbayesianinference_amd import _lib
a = argparse.Namespace(chains=65536, steps=100, warmup=100, no_cpu_baseline=True, draw="f32")
for k in range(3):
    a.draw = "f64" if k == 1 else "f32"
    t0 = time.perf_counter()
    o = bench.bench_c2(a, 0, 1, 0)
    torch.cuda.synchronize()
    print(k, a.draw, "value %.4g steady %.4g wall of call %.2f s" % (o["value"], o["value_steady"], time.perf_counter() - t0), flush=True)
    if k == 1:
        torch.cuda.empty_cache()
