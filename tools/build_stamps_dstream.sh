#!/bin/bash
# Diagnostic build of libpbbi with in-kernel s_memtime stamps in the streamed dense kernel (tools/stamp_probe_dstream.py).
set -e
cd "$(dirname "$0")/.."
C=physicsbasedbayesianinference_amd/csrc
make -C $C -j3 >/dev/null
mkdir -p build/stamps
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -DPBBI_STAMPS=1 \
    -mllvm -pragma-unroll-threshold=100000 -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills -mllvm -amdgpu-use-amdgpu-trackers=1 \
    -c $C/kernels_dstream.hip -o build/stamps/kernels_dstream.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libpbbi_stamps_dstream.so \
    $C/pbbi_api.o $C/kernels_lane.o $C/kernels_lane2.o $C/kernels_sepn.o $C/kernels_rosn.o $C/kernels_rosg.o $C/kernels_stream.o $C/kernels_big.o $C/kernels_dense.o build/stamps/kernels_dstream.o -ldl
echo build/stamps/libpbbi_stamps_dstream.so
