#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/run_profiles.sh) -> profiles/<tag>_kernel_stats.csv and
profiles/<tag>_pmc.json (per-launch means per kernel, HBM traffic calibrated as described in
tools/profile_workload.py)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(root, "gpurun_out", "prof_" + tag)
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
ks = newest(os.path.join(R, "kt", "*", "*_kernel_stats.csv"))
shutil.copy(ks, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
out = {}
for d in ("pmc_sq", "pmc_fetch", "pmc_write"):
    f = newest(os.path.join(R, d, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "k_dense_hmc" if "k_dense_hmc" in k else "k_dense_eval" if "k_dense_eval" in k else None
        if name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in agg.items():
        for c, v in cs.items():
            out.setdefault(name, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
D, N = 128, 65536
known_read = D * N * 8           # k_dense_eval reads q once (P comes from L2 after first touch)
ev, hm = out.get("k_dense_eval", {}), out.get("k_dense_hmc", {})
cal = {}
if "FETCH_SIZE" in ev:
    cal["fetch_kb_to_bytes"] = known_read / ev["FETCH_SIZE"]["mean_per_launch"]
    cal["note"] = ("FETCH_SIZE is in KiB-like units; factor = known bytes read by k_dense_eval "
                   "(D*N*8) / its FETCH_SIZE, i.e. 1024 x the gfx950 under-count correction")
if "FETCH_SIZE" in hm and "WRITE_SIZE" in hm:
    rd = hm["FETCH_SIZE"]["mean_per_launch"] * cal.get("fetch_kb_to_bytes", 1024.0)
    wr = hm["WRITE_SIZE"]["mean_per_launch"] * 1024.0
    cal["k_dense_hmc_read_bytes_per_launch"] = rd
    cal["k_dense_hmc_write_bytes_per_launch"] = wr
    cal["k_dense_hmc_hbm_bytes_per_launch"] = rd + wr
    cal["algorithmic_bytes_per_launch"] = (4 * D * 8 + 9) * N
if "GRBM_GUI_ACTIVE" in hm:
    cyc = hm["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8
    cal["k_dense_hmc_cycles_per_launch"] = cyc
    cal["k_dense_hmc_mfma_busy_frac"] = hm["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (1024 * cyc)
out["derived"] = cal
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(cal, indent=1))
print(open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv")).read()[:700])
