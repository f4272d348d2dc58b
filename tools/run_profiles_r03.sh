#!/bin/bash
# Round-3 profiles.  Run on the GPU box (via gpurun):  tools/run_profiles_r03.sh [tag] [bench args...]
#   kt        rocprofv3 --kernel-trace --stats of the DEFAULT bench.py run (C2 headline + C3 / C5 extras)
#   pmc_*     separate --pmc passes (SQ set, FETCH_SIZE, WRITE_SIZE) over small fixed workloads:
#             workload (C2), c3, c3x (C3 in the reference's operation order), c5
# Summarise with tools/summarize_profiles_r03.py <tag> -> profiles/<tag>_*
set -e
tag=${1:-r03}
export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_$tag
rm -rf $R
mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -- python3 bench.py --no-cpu-baseline > $R/kt.log 2>&1
tail -c 400 $R/kt.log | head -c 300 || true; echo
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY"
for w in workload c3 c3x c5; do
  script=tools/profile_$w.py
  unset C3_EXACT
  if [ $w = c3x ]; then script=tools/profile_c3.py; export C3_EXACT=1; fi
  rocprofv3 --pmc $SQ --output-format csv -d $R/pmc_sq_$w -- python3 $script > $R/pmc_sq_$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch_$w -- python3 $script > $R/pmc_fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc_write_$w -- python3 $script > $R/pmc_write_$w.log 2>&1
  echo "pmc $w done"
done
echo profiles done
