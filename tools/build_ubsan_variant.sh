#!/bin/bash
# Development aid: build/librt_ubsan.so -- the library with the HOST side of rt_api.cpp, rt_dist.cpp, rt_host_build.cpp and rt_camera.cpp
# under UndefinedBehaviorSanitizer in trap mode (no runtime library needed: undefined behaviour ends the process with SIGILL); the
# device code is the product's.  Run the tests against it with RT_HOTPATH_LIB=$PWD/build/librt_ubsan.so python -m pytest tests ...
set -e
cd "$(dirname "$0")/../raytracer_amd/csrc"
make rt_kernels.o rt_scene_records.o rt_device_build.o rt_overlay.o rt_raster.o > /dev/null
mkdir -p ../../build
H=/opt/rocm/bin/hipcc
C="-O1 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math"
ID=$(cat rt_kernels.hip rt_internal.h rt_api.cpp | sha256sum | cut -c1-16)
$H $C -Xarch_host -fsanitize=undefined -Xarch_host -fsanitize-trap=undefined -x hip --offload-arch=gfx950 -DRT_BUILD_ID=\"$ID\" -c rt_api.cpp -o /tmp/u_api.o
$H $C -Xarch_host -fsanitize=undefined -Xarch_host -fsanitize-trap=undefined -x hip --offload-arch=gfx950 -c rt_dist.cpp -o /tmp/u_dist.o
$H $C -fsanitize=undefined -fsanitize-trap=undefined -x c++ -c rt_host_build.cpp -o /tmp/u_host.o -pthread
$H $C -fsanitize=undefined -fsanitize-trap=undefined -x c++ -c rt_camera.cpp -o /tmp/u_cam.o
$H -shared -fPIC --offload-arch=gfx950 -o ../../build/librt_ubsan.so rt_kernels.o rt_scene_records.o /tmp/u_api.o /tmp/u_host.o rt_device_build.o /tmp/u_cam.o rt_overlay.o rt_raster.o /tmp/u_dist.o -pthread -ldl
