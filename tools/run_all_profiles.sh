#!/bin/bash
# On the GPU box (gpurun): every profile and bench line behind the numbers quoted in DESIGN.md / README.md.
#   tools/run_all_profiles.sh <tag>   ->  gpurun_out/prof_<tag>/..., gpurun_out/bench_<tag>_*.json
set -e
tag=${1:-r01}
export TMPDIR=/tmp
O=$PWD/gpurun_out
bash tools/run_profiles.sh $tag
R=$O/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_c3 -- python3 bench.py --workload c3 > $R/kt_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_c3x -- python3 bench.py --workload c3 --exact-order > $R/kt_c3x.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt_c5 -- python3 bench.py --workload c5 > $R/kt_c5.log 2>&1
python3 bench.py > $O/bench_${tag}.json 2> $O/bench_${tag}.err
python3 bench.py --workload c3 > $O/bench_${tag}_c3.json 2>> $O/bench_${tag}.err
python3 bench.py --workload c3 --exact-order > $O/bench_${tag}_c3_exact.json 2>> $O/bench_${tag}.err
python3 bench.py --workload c5 > $O/bench_${tag}_c5.json 2>> $O/bench_${tag}.err
python3 bench.py --workload stream > $O/bench_${tag}_stream.json 2>> $O/bench_${tag}.err
echo all profiles done
