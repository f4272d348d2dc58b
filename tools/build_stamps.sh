#!/bin/bash
# Diagnostic build of libpbbi with in-kernel s_memtime stamps (see tools/stamp_probe.py).
set -e
cd "$(dirname "$0")/.."
C=physicsbasedbayesianinference_amd/csrc
make -C $C -j3 >/dev/null
mkdir -p build/stamps
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -DPBBI_STAMPS=${STAMPS:-1} \
    -c $C/kernels_dense.hip -o build/stamps/kernels_dense.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libpbbi_stamps.so \
    $C/pbbi_api.o $C/kernels_lane.o $C/kernels_lane2.o $C/kernels_sepn.o $C/kernels_rosn.o $C/kernels_rosg.o $C/kernels_stream.o $C/kernels_big.o $C/kernels_dstream.o build/stamps/kernels_dense.o -ldl
echo build/stamps/libpbbi_stamps.so
