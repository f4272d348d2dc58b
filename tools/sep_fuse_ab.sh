set -o pipefail
mkdir -p gpurun_out/sep
rc=0

for d in 32 64 128; do
  timeout -k 10 200 python bench.py --workload stream --potential diag --dim $d --chains 262144 --no-cpu-baseline > gpurun_out/sep/diag_$d.json 2>gpurun_out/sep/diag_$d.err || exit 1
  PBBI_NO_SEP_FUSE=1 timeout -k 10 200 python bench.py --workload stream --potential diag --dim $d --chains 262144 --no-cpu-baseline > gpurun_out/sep/diag_${d}_nofuse.json 2>/dev/null || exit 1
done
python - <<PY
import json
for d in (32,64,128):
    for t in ("","_nofuse"):
        x=json.loads(open(f"gpurun_out/sep/diag_{d}{t}.json").read().strip().splitlines()[-1]); print(d,t,x["value"],x["roofline"]["frac"],x["roofline"].get("frac_steady"))
PY
