#!/usr/bin/env python3
"""Phase shares of the streamed dense kernel (kernels_dstream.hip, D = 256) from in-kernel s_memtime stamps.

    bash tools/build_stamps_dstream.sh && PBBI_LIB=build/stamps/libpbbi_stamps_dstream.so python tools/stamp_probe_dstream.py
The stamped build forbids overlaps the real kernel has: read shares, never its run time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from physicsbasedbayesianinference_amd import _lib
import physicsbasedbayesianinference_amd as P

D, N, L = int(os.environ.get("PBBI_TIME_D", 256)), int(os.environ.get("PBBI_TIME_N", 16384)), 10
lib = _lib.load()
A = np.random.RandomState(0).standard_normal((D, D))
Pm = np.linalg.inv(A @ A.T / D + np.eye(D)); Pm = 0.5 * (Pm + Pm.T)
pot = P.GaussianDense(None, precision=Pm, const=0.0)
WAVES = 8 if D <= 128 else 4          # (D <= 128: the resident-P kernel of kernels_dense.hip, tools/build_stamps.sh)
nblk = N // (16 * WAVES)
stamps = torch.zeros((nblk * 8, 64), dtype=torch.int64, device="cuda")
lib.pbbi_debug_set_stamp_buffer.argtypes = [C.c_void_p]
lib.pbbi_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
q = torch.randn((D, N), dtype=torch.float64, device="cuda")
S = int(os.environ.get("STAMP_ITERS", 4))
samples = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
mom = torch.empty((S, D, N), dtype=torch.float64, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
_lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
          rej.data_ptr(), None, N, N, 0.1, L, S, 1, 1, 0, 0, 1.0, None)
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.int64)  # the launch's last iteration
names = {42: "iteration top", 2: "draw done", 3: "pp, v = p/m", 4: "carried g: H_old, half kick", 40: "steps done", 41: "stored"}
for j in range(L):
    names[5 + 2 * j] = f"step{j} begin"; names[6 + 2 * j] = f"step{j} pass0+kick"
order = [42, 2, 3, 4] + [k for j in range(L) for k in (5 + 2 * j, 6 + 2 * j)] + [40, 41]
for grp in ([range(4)] if WAVES == 4 else [range(4), range(4, 8)]):
    rows = [b * 8 + w for b in (0, 17, 100, nblk - 1) for w in grp]
    T = st[rows][:, order]
    d = np.diff(T, axis=1)
    tot = (T[:, -1] - T[:, 0]).mean()
    print(f"waves {list(grp)}")
    for i, k in enumerate(order[1:]):
        print(f"  {names[k]:28s} {d[:, i].mean():10.0f} cyc  {100 * d[:, i].mean() / tot:5.1f} %   (min {d[:, i].min()}, max {d[:, i].max()})")
    print(f"  one iteration: {tot:.0f} shader cycles")
