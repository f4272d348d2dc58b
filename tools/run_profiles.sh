#!/bin/bash
# Run on the GPU box (via gpurun):  tools/run_profiles.sh <tag>
# Produces gpurun_out/prof_<tag>/{kt,pmc_sq,pmc_fetch,pmc_write}; summarise with
# tools/summarize_profiles.py <tag> -> profiles/<tag>_*.{csv,json}
set -e
tag=${1:-r01}
export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_$tag
rm -rf $R
mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -- python3 bench.py --no-cpu-baseline > $R/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $R/pmc_sq -- python3 tools/profile_workload.py > $R/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -- python3 tools/profile_workload.py > $R/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -- python3 tools/profile_workload.py > $R/pmc_write.log 2>&1
echo profiles done
