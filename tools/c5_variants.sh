#!/bin/bash
# Build-variant A/B of the C5 GEMM on the GPU box: tools/c5_variants.sh <variant>...  where a variant is
# "default", "narrow" (PBBI_BIG_TILE=128) or the name of a build/<name>/libpbbi.so made by
# tools/build_variant.sh.  Runs the wide-tile parity test, then bench.py --workload c5, per variant.
set -o pipefail
mkdir -p gpurun_out/c5w
for v in "$@"; do
  lib=physicsbasedbayesianinference_amd/libpbbi.so; tile=0
  [ $v != default ] && [ $v != narrow ] && lib=build/$v/libpbbi.so
  [ $v = narrow ] && tile=128
  rc=0; [ -z "$NOTEST" ] && { PBBI_LIB=$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide" > gpurun_out/c5w/tests_$v.log 2>&1; rc=$?; }
  [ $rc -ne 0 ] && { tail -5 gpurun_out/c5w/tests_$v.log; exit $rc; }
  PBBI_LIB=$lib PBBI_BIG_TILE=$tile timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 20 --warmup 10 > gpurun_out/c5w/bench_$v.json 2> gpurun_out/c5w/bench_$v.err; rc=$?
  [ $rc -ne 0 ] && { tail -5 gpurun_out/c5w/bench_$v.err; exit $rc; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/c5w/bench_$v.json").read().strip().splitlines()[-1])
print("$v", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done
