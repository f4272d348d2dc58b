#!/bin/bash
# Build a variant of libpbbi in which ONE translation unit is replaced / compiled with extra flags:
#   tools/build_variant.sh <name> <unit> [source.hip] -- <flags...>   ->  build/<name>/libpbbi.so
# <unit> is kernels_dense, kernels_lane2, ...; [source.hip] defaults to csrc/<unit>.hip.
# Use the result with PBBI_LIB=build/<name>/libpbbi.so.
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
C=physicsbasedbayesianinference_amd/csrc
src=$C/$unit.hip
if [ "$1" != "--" ]; then src=$1; shift; fi
shift  # the "--"
make -C $C -j3 >/dev/null
mkdir -p build/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$C "$@" \
    -c $src -o build/$name/$unit.o
objs=""
for f in pbbi_api kernels_lane kernels_lane2 kernels_sepn kernels_rosn kernels_rosg kernels_stream kernels_dense kernels_big; do
  if [ $f = $unit ]; then objs="$objs build/$name/$unit.o"; else objs="$objs $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/$name/libpbbi.so $objs -ldl
echo build/$name/libpbbi.so
