#!/usr/bin/env python3
"""Per-wave timeline of k_ros2_hmc on config C3 from a diagnostic build (tools/build_stamps_ros2.sh):
    PBBI_LIB=build/stamps/libpbbi_stamps_ros2.so python tools/ros2_timeline.py [--exact]
Stamps are s_memrealtime ticks (10 ns).  Read the timeline, never the run time of this build.
Run with PBBI_FUSE_ITERS=1 for the timeline of ONE iteration per launch (what DESIGN.md 4.2 analyses);
with fused launches stamps 2-4 are those of a wave's last iteration."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from physicsbasedbayesianinference_amd import _lib
import physicsbasedbayesianinference_amd as P

d, N, h, L = 32, 262144, 0.01, int(os.environ.get("ROS2_L", "10"))
exact = "--exact" in sys.argv
lib = _lib.load()
pot = P.Rosenbrock(d)
nblk = N // 32
stamps = torch.zeros((nblk, 8), dtype=torch.int64, device="cuda")
lib.pbbi_debug_set_ros2_stamp_buffer.argtypes = [C.c_void_p]
lib.pbbi_debug_set_ros2_stamp_buffer(C.c_void_p(stamps.data_ptr()))
q = torch.empty((d, N), dtype=torch.float64, device="cuda")
_lib.call("pbbi_philox_normal", 7, _lib.STREAM_POSITION, 0, 0, d, N, N, 0.1, None, _lib.F64, 0, q.data_ptr(), None)
q += 1.0
S = 50
samples = torch.empty((S, d, N), dtype=torch.float64, device="cuda")
mom = torch.empty((S, d, N), dtype=torch.float64, device="cuda")
rej = torch.empty((S, N), dtype=torch.uint8, device="cuda")
flags = 1 | (0 if exact else 2)
for rep in range(4):
    _lib.call("pbbi_hmc_run", pot.handle, 0, q.data_ptr(), None, samples.data_ptr(), mom.data_ptr(),
              rej.data_ptr(), None, N, N, h, L, S, flags, 7, rep * S, 0, 1.0, None)
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.int64)  # the last launch's stamps
T = st[:, :6] - st[:, 0].min()
hw, xcc = st[:, 7], st[:, 6] & 0xF
wave_id, simd, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
simd_key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
print(f"L={L} exact={exact}: kernel span {T[:, 5].max() / 100:.2f} us (first start -> last store done)")
print(f"distinct SIMDs seen: {len(np.unique(simd_key))}, wave slots used: {np.unique(wave_id)}")
names = ["tile begins", "inputs waited for", "p drawn, H_old", "trajectory done", "stores issued", "tile ends"]
d_ = np.diff(T, axis=1)
for i in range(5):
    x = d_[:, i] / 100
    print(f"  {names[i]:28s} -> {names[i+1]:28s}: mean {x.mean():6.2f} us  p10 {np.percentile(x,10):6.2f}  p90 {np.percentile(x,90):6.2f}")
life = (T[:, 5] - T[:, 0]) / 100
print(f"  wave lifetime mean {life.mean():.2f} us; start times: p1 {np.percentile(T[:,0],1)/100:.2f} p25 {np.percentile(T[:,0],25)/100:.2f} "
      f"p50 {np.percentile(T[:,0],50)/100:.2f} p75 {np.percentile(T[:,0],75)/100:.2f} p99 {np.percentile(T[:,0],99)/100:.2f} us")
# chip-wide phase census over time: how many waves are resident / in the vector phases / waiting on stores
grid = np.arange(0, T[:, 5].max(), 100)  # every 1 us
print("  t(us) resident  loading+draw  trajectory  storing")
for t in grid:
    res = ((T[:, 0] <= t) & (T[:, 5] > t)).sum()
    ld = ((T[:, 1] <= t) & (T[:, 2] > t)).sum()
    tr = ((T[:, 2] <= t) & (T[:, 3] > t)).sum()
    stw = ((T[:, 3] <= t) & (T[:, 5] > t)).sum()
    print(f"  {t/100:5.0f} {res:8d} {ld:12d} {tr:11d} {stw:8d}")
# per-SIMD: fraction of the kernel span during which at least one wave is in its trajectory
span = T[:, 5].max()
fr, cnt = [], []
for k in np.unique(simd_key):
    m = simd_key == k
    ev = sorted([(a, 1) for a in T[m, 2]] + [(b, -1) for b in T[m, 3]])
    cur, last, busy = 0, 0, 0
    for t, dlt in ev:
        if cur > 0:
            busy += t - last
        cur += dlt
        last = t
    fr.append(busy / span)
    cnt.append(m.sum())
print(f"per SIMD: waves {np.mean(cnt):.2f} (min {np.min(cnt)} max {np.max(cnt)}); fraction of the span with >=1 wave in its trajectory: "
      f"mean {np.mean(fr):.3f} min {np.min(fr):.3f} max {np.max(fr):.3f}")
