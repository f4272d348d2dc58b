#!/usr/bin/env python3
"""Where does the host-side stall of the first timed run of an EMBEDDED bench line come from?  Replays bench.py's
default sequence (headline, CPU baseline, the f64-draw line) with the host duration of every pbbi_hmc_run call
and of every synchronize printed.  variants: argv[1] in {full, nocpu, sleep}"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from physicsbasedbayesianinference_amd import _lib

variant = sys.argv[1] if len(sys.argv) > 1 else "full"
log = []
real_call = _lib.call


def timed_call(name, *a):
    t0 = time.perf_counter()
    r = real_call(name, *a)
    if name == "pbbi_hmc_run":
        log.append(("run S=%d" % a[12], (time.perf_counter() - t0) * 1e3))
    return r


_lib.call = timed_call
real_sync = torch.cuda.synchronize


def timed_sync(*a, **k):
    t0 = time.perf_counter()
    real_sync(*a, **k)
    log.append(("sync", (time.perf_counter() - t0) * 1e3))


torch.cuda.synchronize = timed_sync
a = argparse.Namespace(chains=65536, steps=100, warmup=100, no_cpu_baseline=(variant == "nocpu"), draw="f32")
o = bench.bench_c2(a, 0, 1, 0)
print("headline value %.4g steady %.4g" % (o["value"], o["value_steady"]), flush=True)
if variant == "sleep":
    time.sleep(2.0)
log.clear()
a.draw, a.no_cpu_baseline = "f64", True
o = bench.bench_c2(a, 0, 1, 0)
print("f64 value %.4g steady %.4g" % (o["value"], o["value_steady"]))
print(" ".join("%s=%.1fms" % x for x in log))
