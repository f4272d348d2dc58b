set -o pipefail
mkdir -p gpurun_out/c5c
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "big or c5 or gemm or dense or getsamples" > gpurun_out/c5c/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c5c/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 20 --warmup 10 > gpurun_out/c5c/bench.json 2> gpurun_out/c5c/bench.err || exit 1
PBBI_NO_CARRY=1 timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 20 --warmup 10 > gpurun_out/c5c/bench_nocarry.json 2> gpurun_out/c5c/bench_nocarry.err || exit 1
python - <<PY
import json
for f in ("bench","bench_nocarry"):
    d=json.loads(open("gpurun_out/c5c/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
