#!/bin/bash
# dense-precision Gaussians across D: bench.py --workload dense lines into gpurun_out/<tag>/dense_d*.json
tag=${1:-dense}
mkdir -p gpurun_out/$tag
for d in 32 64 80 96 100 128 256 512; do
  n=65536
  python bench.py --workload dense --dim $d --chains $n --steps 50 --warmup 50 > gpurun_out/$tag/dense_d$d.json 2> gpurun_out/$tag/dense_d$d.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/dense_d$d.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("D=$d value %.3g steady %.3g ms/iter %.4f frac %.3f steady %.3f | %s" % (d["value"], d["value_steady"], r["iteration_ms_steady"], r["frac"], r["frac_steady"], d["config"]["route"][:110]))
PY
done
