#!/usr/bin/env python3
"""Print hipcc's -Rpass-analysis=kernel-resource-usage as one line per kernel.

usage: tools/kernel_resources.py physicsbasedbayesianinference_amd/csrc/kernels_dense.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    txt = m.group(1)
    if txt.startswith("Function Name:"):
        cur = {"name": txt.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in txt:
        k, v = txt.split(":", 1)
        cur[k.strip()] = v.strip()
keys = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]",
        "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"]
print("%-70s %6s %6s %6s %8s %4s %6s %6s %8s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "sSpill", "vSpill", "LDS"))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print("%-70s %6s %6s %6s %8s %4s %6s %6s %8s" % tuple([name] + [r.get(k, "?") for k in keys]))
