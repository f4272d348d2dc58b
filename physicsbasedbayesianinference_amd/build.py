"""Build libpbbi.so in-tree:  python -m physicsbasedbayesianinference_amd.build [-B]

hipcc cross-compiles gfx950 code objects without a GPU; the resulting .so is
git-ignored but travels to the GPU box with the working tree.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False, jobs=None, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(jobs or min(4, os.cpu_count() or 1))]
    if force:
        cmd.append("-B")
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("libpbbi.so build failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return os.path.join(_HERE, "libpbbi.so")


if __name__ == "__main__":
    print(build(force="-B" in sys.argv, verbose=True))
