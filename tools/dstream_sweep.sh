#!/bin/bash
# the streamed dense kernel (128 < D <= 256) against the GEMM path it replaces: bench.py --workload dense
tag=${1:-dstream}
mkdir -p gpurun_out/$tag
for d in 160 192 200 256; do
  for v in stream gemm; do
    if [ $v = gemm ]; then export PBBI_NO_DENSE_STREAM=1; else unset PBBI_NO_DENSE_STREAM; fi
    timeout -k 10 300 python bench.py --workload dense --dim $d --chains 65536 --steps 20 --warmup 20 > gpurun_out/$tag/dense_d${d}_$v.json 2> gpurun_out/$tag/dense_d${d}_$v.err || { echo "D=$d $v FAILED"; tail -3 gpurun_out/$tag/dense_d${d}_$v.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/dense_d${d}_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("D=$d $v value %.3g steady %.3g ms/iter %.4f frac %.3f steady %.3f | %s" % (d["value"], d["value_steady"], r["iteration_ms_steady"], r["frac"], r["frac_steady"], d["config"]["route"][:90]))
PY
  done
done
