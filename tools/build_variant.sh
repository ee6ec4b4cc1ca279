#!/bin/bash
# usage: tools/build_variant.sh NAME "-DMACRO=..."  -- builds variants/libmp_NAME.so from the tree's kernels.hip with extra
# compile flags (A/B runs through tools/ab.sh / MINIPATH_HIP_SO); the host objects of the normal build are reused.
set -e
cd "$(dirname "$0")/../minipath_amd/csrc"
make -s
mkdir -p ../../variants
hipcc --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -O3 -std=c++17 -fPIC \
  -Wall -Wno-unused-function -Wno-inline-asm -ffp-contract=off -fno-fast-math $2 -c kernels.hip -o /tmp/kernels_$1.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libmp_$1.so /tmp/kernels_$1.o scene_build.o device_tree.o host_camera.o host_api.o -lpthread
echo variants/libmp_$1.so
