#!/bin/bash
# rocprofv3 passes over the streamed dense kernel (D = 256): kernel-trace stats of the bench line, then separate
# PMC passes (SQ set, FETCH_SIZE, WRITE_SIZE) over tools/profile_dstream.py.  Flat copies land in gpurun_out/prof_dstream/.
set -e
export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_dstream
rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -- python3 bench.py --workload dense --dim 256 --steps 40 --warmup 40 --no-cpu-baseline > $R/bench_d256.json 2> $R/kt.err
cp $(find $R/kt -name "*kernel_stats.csv" | head -1) $R/kernel_stats_dense_d256.csv
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY"
rocprofv3 --pmc $SQ --output-format csv -d $R/pmc_sq -- python3 tools/profile_dstream.py > $R/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -- python3 tools/profile_dstream.py > $R/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -- python3 tools/profile_dstream.py > $R/pmc_write.log 2>&1
for k in sq fetch write; do cp $(find $R/pmc_$k -name "*counter_collection.csv" | head -1) $R/pmc_$k.csv; done
python3 - <<PY
import csv, json, collections
R = "$R"
out = {}
for k in ("sq", "fetch", "write"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f"{R}/pmc_{k}.csv")):
        name = row["Kernel_Name"]
        if "k_dense_hmc" not in name: continue
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
    for name, c in acc.items():
        out.setdefault(name[:120], {}).update(c)
for name, c in out.items():
    if c.get("SQ_BUSY_CYCLES"): c["mfma_busy_over_sq_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
json.dump(out, open(f"{R}/pmc_dense_d256.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
echo profiles done
