#!/bin/bash
# Diagnostic build of libpbbi with per-wave timeline stamps in k_ros2_hmc (tools/ros2_timeline.py).
set -e
cd "$(dirname "$0")/.."
C=physicsbasedbayesianinference_amd/csrc
make -C $C -j3 >/dev/null
mkdir -p build/stamps
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -DPBBI_STAMPS_ROS2=1 $EXTRA \
    -c $C/kernels_lane2.hip -o build/stamps/kernels_lane2.o
objs=""
for f in pbbi_api kernels_lane kernels_sepn kernels_rosn kernels_rosg kernels_stream kernels_dense kernels_big; do objs="$objs $C/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libpbbi_stamps_ros2.so $objs build/stamps/kernels_lane2.o -ldl
echo build/stamps/libpbbi_stamps_ros2.so
