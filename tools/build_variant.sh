#!/bin/bash
# Builds build/librt_<name>.so with extra -D flags for rt_kernels.hip (development aid for parameter sweeps).
# usage: tools/build_variant.sh name -DFOO=1 -DBAR=2
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
name=$1; shift
mkdir -p ../../build
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -fgpu-flush-denormals-to-zero -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -c rt_kernels.hip -o /tmp/rt_kernels_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/librt_$name.so /tmp/rt_kernels_$name.o rt_scene_records.o rt_api.o rt_host_build.o rt_device_build.o rt_camera.o rt_overlay.o rt_raster.o rt_dist.o -pthread -ldl
