set -o pipefail
mkdir -p gpurun_out/c5w
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "big or c5" > gpurun_out/c5w/tests.log 2>&1; rc=$?; tail -3 gpurun_out/c5w/tests.log
[ $rc -ne 0 ] && exit $rc
for v in wide2 wide3; do
  PBBI_LIB=build/$v/libpbbi.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide" > gpurun_out/c5w/tests_$v.log 2>&1; rc=$?; tail -1 gpurun_out/c5w/tests_$v.log
  [ $rc -ne 0 ] && exit $rc
done
for v in default wide2 wide3 narrow; do
  lib=physicsbasedbayesianinference_amd/libpbbi.so; tile=0
  [ $v = wide2 ] && lib=build/wide2/libpbbi.so
  [ $v = wide3 ] && lib=build/wide3/libpbbi.so
  [ $v = narrow ] && tile=128
  PBBI_LIB=$lib PBBI_BIG_TILE=$tile timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 20 --warmup 10 > gpurun_out/c5w/bench_$v.json 2> gpurun_out/c5w/bench_$v.err; rc=$?
  [ $rc -ne 0 ] && { tail -5 gpurun_out/c5w/bench_$v.err; exit $rc; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/c5w/bench_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_steady"))
PY
done
