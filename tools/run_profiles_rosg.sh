#!/bin/bash
# SQ counters of the Rosenbrock kick-drift-kick kernels at D = 32 (k_ros2_hmc, config C3), 64 and 128 (k_rosg_hmc):
# VALU instructions per element-step and the share of cycles the SIMDs issue -- are the multi-lane kernels bound
# by the same thing as C3?  Output: gpurun_out/prof_rosg/summary.json
set -e
export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_rosg
rm -rf $R; mkdir -p $R
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVES"
for d in 32 64 128; do
  C3_D=$d rocprofv3 --pmc $SQ --output-format csv -d $R/sq_$d -- python3 tools/profile_c3.py > $R/sq_$d.log 2>&1
  cp $(find $R/sq_$d -name "*counter_collection.csv" | head -1) $R/sq_$d.csv
done
python3 - <<PY
import csv, json, collections
R = "$R"
out = {}
for d in (32, 64, 128):
    acc = collections.defaultdict(float); launches = set()
    for row in csv.DictReader(open(f"{R}/sq_{d}.csv")):
        if "_hmc" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); launches.add(row["Dispatch_Id"]); name = row["Kernel_Name"][:90]
    iters, N, L = 32, 262144, 10
    elem_steps = iters * N * L * d
    cyc = acc["GRBM_GUI_ACTIVE"] / 8
    out[d] = dict(kernel=name, launches=len(launches), valu_instructions_per_element_step=acc["SQ_INSTS_VALU"] * 64 / elem_steps,
                  cycles_per_iteration=cyc / iters, us_per_iteration_at_2p4GHz=cyc / iters / 2400.0,
                  element_steps_per_second_at_2p4GHz=elem_steps / (cyc / 2.4e9),
                  issue_floor_fraction=acc["SQ_INSTS_VALU"] * 4.5 / (1024 * cyc),
                  wait_inst_any_over_active_valu=acc["SQ_WAIT_INST_ANY"] / acc["SQ_ACTIVE_INST_VALU"])
out["note"] = ("SQ_INSTS_VALU counts wave-instructions: x 64 lanes / element-steps = vector instructions per element-step "
               "(a chain's element occupies one lane); issue_floor_fraction = instructions x 4.5 cycles per instruction and "
               "SIMD (tools/ubench/valu_f64_clock.hip) / (1024 SIMDs x cycles): the share of the launch the SIMDs spend "
               "issuing fp64 vector instructions")
json.dump(out, open(f"{R}/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
