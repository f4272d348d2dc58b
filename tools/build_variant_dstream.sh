#!/bin/bash
# A/B builds of the streamed dense kernel: tools/build_variant_dstream.sh <name> <extra hipcc flags...>  ->  build/variants/libpbbi_<name>.so
set -e
cd "$(dirname "$0")/.."
C=physicsbasedbayesianinference_amd/csrc
name=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" \
    -mllvm -pragma-unroll-threshold=100000 -mllvm -disable-machine-licm -mllvm -sink-insts-to-avoid-spills -mllvm -amdgpu-use-amdgpu-trackers=1 \
    -c $C/kernels_dstream.hip -o build/variants/kernels_dstream_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libpbbi_$name.so \
    $C/pbbi_api.o $C/kernels_lane.o $C/kernels_lane2.o $C/kernels_sepn.o $C/kernels_rosn.o $C/kernels_rosg.o $C/kernels_stream.o $C/kernels_big.o $C/kernels_dense.o build/variants/kernels_dstream_$name.o -ldl
echo build/variants/libpbbi_$name.so
