#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/run_profiles_r03.sh) -> profiles/<tag>_kernel_stats.csv,
profiles/<tag>_pmc.json (C2: k_dense_hmc, with the FETCH_SIZE calibration on k_dense_eval described in
tools/profile_workload.py), profiles/<tag>_pmc_c3.json (k_ros2_hmc: one launch = 16 fused iterations),
profiles/<tag>_pmc_c3_exact.json (the same kernel in the reference's operation order), profiles/<tag>_pmc_c5.json (k_big_gemm)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(root, "gpurun_out", "prof_" + tag)
P = os.path.join(root, "profiles")
os.makedirs(P, exist_ok=True)
newest = lambda pat: max(glob.glob(pat, recursive=True), key=os.path.getmtime)
shutil.copy(newest(os.path.join(R, "kt", "**", "*_kernel_stats.csv")), os.path.join(P, f"{tag}_kernel_stats.csv"))


def counters(w, kernels):
    out = {}
    for kind in ("sq", "fetch", "write"):
        f = newest(os.path.join(R, f"pmc_{kind}_{w}", "**", "*_counter_collection.csv"))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for name in kernels:
                if name in r["Kernel_Name"]:
                    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, cs in agg.items():
            for c, v in cs.items():
                out.setdefault(name, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    return out


# ---- C2
out = counters("workload", ["k_dense_hmc", "k_dense_eval"])
D, N = 128, 65536
ev, hm = out.get("k_dense_eval", {}), out.get("k_dense_hmc", {})
cal = {}
if "FETCH_SIZE" in ev:
    cal["fetch_kb_to_bytes"] = D * N * 8 / ev["FETCH_SIZE"]["mean_per_launch"]
    cal["note"] = ("FETCH_SIZE factor = known bytes read by k_dense_eval (D*N*8) / its FETCH_SIZE: 1024 x the "
                   "gfx950 under-count correction for this 8-B-per-lane access pattern")
# tools/profile_workload.py runs C2_ITERS iterations of one pbbi_hmc_run in ONE fused launch (its first
# iteration forms and stores g(q_0), the others read the carried gradient) -- counters are summed over the
# launches and divided by the iterations
C2_ITERS = 9
tot = lambda c: hm[c]["mean_per_launch"] * hm[c]["launches"]
if "FETCH_SIZE" in hm and "WRITE_SIZE" in hm:
    rd = tot("FETCH_SIZE") * cal.get("fetch_kb_to_bytes", 2048.0) / C2_ITERS
    wr = tot("WRITE_SIZE") * 1024.0 / C2_ITERS
    cal.update(k_dense_hmc_read_bytes_per_iteration=rd, k_dense_hmc_write_bytes_per_iteration=wr,
               k_dense_hmc_hbm_bytes_per_iteration=rd + wr, algorithmic_bytes_per_iteration=(4 * D * 8 + 9) * N,
               executed_bytes_per_iteration=(6 * D * 8 + 9) * N, iterations=C2_ITERS,
               launches=hm["FETCH_SIZE"]["launches"])
if "GRBM_GUI_ACTIVE" in hm:
    cyc = tot("GRBM_GUI_ACTIVE") / 8 / C2_ITERS
    cal["k_dense_hmc_cycles_per_iteration"] = cyc
    # (SQ_VALU_MFMA_BUSY_CYCLES saturates at 2^32 on the fused launch: busy from the instruction count --
    #  11 mat-vecs in the run's first iteration, 10 in each later one, (D/16)(D/4) v_mfma_f64_16x16x4 of 64 cycles
    #  per mat-vec and 16-chain tile)
    mfma = (11 + 10 * (C2_ITERS - 1)) * (D // 16) * (D // 4) * (N // 16)
    cal["k_dense_hmc_mfma_busy_frac"] = mfma * 64 / (1024 * cyc * C2_ITERS)
out["derived"] = cal
json.dump(out, open(os.path.join(P, f"{tag}_pmc.json"), "w"), indent=1)
print("C2", json.dumps(cal, indent=1))

# ---- C3: one launch = 16 fused iterations; KDK/FMA form and the reference's operation order
for w, name in (("c3", "pmc_c3"), ("c3x", "pmc_c3_exact")):
    if not glob.glob(os.path.join(R, f"pmc_sq_{w}")):
        continue
    out = counters(w, ["k_ros2_hmc"])
    k = out.get("k_ros2_hmc", {})
    der = {"iterations_per_launch": 16,
           "form": "reference operation order (bit-exact)" if w == "c3x" else "kick-drift-kick with FMA"}
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        # q is read once per LAUNCH (8 B per lane: the factor calibrated above), samples are written every iteration
        rd = k["FETCH_SIZE"]["mean_per_launch"] * cal.get("fetch_kb_to_bytes", 2048.0)
        wr = k["WRITE_SIZE"]["mean_per_launch"] * 1024.0
        der.update(read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=(rd + wr) / 16,
                   hbm_bytes_per_fused_launch=rd + wr, algorithmic_bytes_per_iteration=(4 * 32 * 8 + 9) * 262144)
        der["note"] = "hbm_bytes_per_launch is per ITERATION (what bench.py's roofline.traffic is compared with)"
    if "SQ_WAIT_INST_ANY" in k:
        der["wait_inst_any_over_active_valu"] = (k["SQ_WAIT_INST_ANY"]["mean_per_launch"] /
                                                 k["SQ_ACTIVE_INST_VALU"]["mean_per_launch"])
        der["valu_instructions_per_wave_iteration"] = k["SQ_INSTS_VALU"]["mean_per_launch"] / 8192 / 16
        # a lane holds 16 dims of its chain and takes L = 10 steps per iteration: ALL vector instructions of
        # an iteration (draw, energies, decision included) per element-step of a lane
        der["valu_instructions_per_lane_element_step"] = der["valu_instructions_per_wave_iteration"] / (16 * 10)
        der["cycles_per_launch"] = k["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8
    out["derived"] = der
    json.dump(out, open(os.path.join(P, f"{tag}_{name}.json"), "w"), indent=1)
    print(w, json.dumps(der, indent=1))

# ---- C5
out = counters("c5", ["k_big_gemm"])
k = out.get("k_big_gemm", {})
der = {}
if "GRBM_GUI_ACTIVE" in k:
    cyc = k["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8
    mfma = 2.0 * 4096 * 4096 * 8192 / 4096        # v_mfma_f32_32x32x2 = 4096 flop, 64 cycles on its SIMD
    der.update(cycles_per_launch=cyc, mfma_instructions_per_launch=mfma,
               mfma_busy_frac_from_count=mfma * 64 / (1024 * cyc),
               note="SQ_VALU_MFMA_BUSY_CYCLES saturates at 2^32 on this kernel: busy = MFMA count x 64 cycles / (1024 SIMDs x cycles)")
if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
    der["fetch_bytes_per_launch_x2"] = k["FETCH_SIZE"]["mean_per_launch"] * 2048.0
    der["write_bytes_per_launch"] = k["WRITE_SIZE"]["mean_per_launch"] * 1024.0
out["derived"] = der
json.dump(out, open(os.path.join(P, f"{tag}_pmc_c5.json"), "w"), indent=1)
print("C5", json.dumps(der, indent=1))
print(open(os.path.join(P, f"{tag}_kernel_stats.csv")).read()[:1500])
