#!/usr/bin/env python3
"""Phase shares of k_dense_leapfrog2 from in-kernel s_memtime stamps (diagnostic build).

    make -C physicsbasedbayesianinference_amd/csrc -B EXTRA=-DPBBI_STAMPS OUT=/tmp/libpbbi_stamps.so
    PBBI_LIB=/tmp/... python tools/stamp_probe.py
The stamped build forbids overlaps the real kernel has: read shares, never its run time.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from physicsbasedbayesianinference_amd import _lib
import physicsbasedbayesianinference_amd as P

D, N, L = 128, 65536, 10
LIGHT = os.environ.get("STAMPS_LIGHT") == "1"  # PBBI_STAMPS=2 build: entry/exit, 100 MHz clock
lib = _lib.load()
A = np.random.RandomState(0).standard_normal((D, D))
Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0)
nblk = N // 128
stamps = torch.zeros((nblk * 8, 64), dtype=torch.int64, device="cuda")
lib.pbbi_debug_set_stamp_buffer.argtypes = [C.c_void_p]
lib.pbbi_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
q = torch.randn((D, N), dtype=torch.float64, device="cuda")
S = int(os.environ.get("STAMP_ITERS", "3"))
samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
mom = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
_lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
          rej.data_ptr(), None, N, N, 0.1, L, S, 1, 1, 0, 0, 1.0, None)
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.int64)  # last iteration's stamps
if not LIGHT:
    names = {0: "start", 1: "LDS staged", 2: "RNG done", 3: "q,p loaded", 4: "g(q0)+kick", 40: "steps done",
             41: "stored"}
    for j in range(L):
        names[5 + 2 * j] = f"step{j} begin"; names[6 + 2 * j] = f"step{j} pass0+kick"
    order = [0, 1, 2, 3, 4] + [k for j in range(L) for k in (5 + 2 * j, 6 + 2 * j)] + [40, 41]
    for label, rows in (("waves 0-3 (prio 1)", [b * 8 + w for b in (0, 100, 300) for w in range(4)]),
                        ("waves 4-7 (prio 0)", [b * 8 + w for b in (0, 100, 300) for w in range(4, 8)])):
        print(label)
        T = st[rows][:, order]
        d = np.diff(T, axis=1)
        tot = (T[:, -1] - T[:, 0]).mean()
        for i, k in enumerate(order[1:]):
            print(f"  {names[k]:22s} {d[:, i].mean():10.0f} cyc  {100 * d[:, i].mean() / tot:5.1f} %")
        print(f"  total {tot:.0f} cycles (memtime ticks)")
# ---- workgroup timeline (ticks = shader cycles)
W = st.reshape(nblk, 8, 64)
t0 = W[:, :, 0].min()
wg_start = W[:, :, 0].min(axis=1) - t0
wg_end = W[:, :, 41].max(axis=1) - t0
dur = wg_end - wg_start
order_ = np.argsort(wg_start)
unit = "x10ns" if LIGHT else "cyc"
print(f"[{unit}] kernel span {wg_end.max():.0f}; WG duration mean {dur.mean():.0f} min {dur.min():.0f} max {dur.max():.0f}")
first = order_[:256]; second = order_[256:]
print(f"round 1: starts {wg_start[first].min():.0f}..{wg_start[first].max():.0f}, ends {wg_end[first].min():.0f}..{wg_end[first].max():.0f}")
if len(second):
    print(f"round 2: starts {wg_start[second].min():.0f}..{wg_start[second].max():.0f}, ends {wg_end[second].min():.0f}..{wg_end[second].max():.0f}")
    # gap between a round-1 end and the next start on (presumably) the same CU: compare sorted lists
    e1 = np.sort(wg_end[first]); s2 = np.sort(wg_start[second])
    print(f"median (round-2 start - round-1 end), rank-matched: {np.median(s2 - e1[:len(s2)]):.0f} cyc")
w0 = W[:, 0, :]; w4 = W[:, 4, :]
print(f"wave0 busy span mean {(w0[:,41]-w0[:,0]).mean():.0f}; wave4 busy span mean {(w4[:,41]-w4[:,0]).mean():.0f}")
