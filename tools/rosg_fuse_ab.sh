#!/bin/bash
# A/B of the fused multi-lane Rosenbrock kernel on the GPU box (PBBI_NO_SEP_FUSE=1: one launch per iteration)
set -o pipefail
mkdir -p gpurun_out/rosg
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "rosenbrock or ros" > gpurun_out/rosg/tests.log 2>&1; rc=$?; tail -3 gpurun_out/rosg/tests.log
[ $rc -ne 0 ] && exit $rc
for d in 64 128; do
  timeout -k 10 200 python bench.py --workload stream --potential rosenbrock --dim $d --chains 131072 --no-cpu-baseline > gpurun_out/rosg/ros_$d.json 2>gpurun_out/rosg/ros_$d.err || exit 1
  PBBI_NO_SEP_FUSE=1 timeout -k 10 200 python bench.py --workload stream --potential rosenbrock --dim $d --chains 131072 --no-cpu-baseline > gpurun_out/rosg/ros_${d}_nofuse.json 2>/dev/null || exit 1
done
python - <<PY
import json
for d in (64,128):
    for t in ("","_nofuse"):
        x=json.loads(open(f"gpurun_out/rosg/ros_{d}{t}.json").read().strip().splitlines()[-1]); print(d,t,x["value"],x["roofline"]["frac"])
PY
