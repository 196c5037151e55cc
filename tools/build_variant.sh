#!/bin/bash
# Builds a variant of the library with extra compiler flags for A/B runs in one GPU call:
#   tools/build_variant.sh NAME -DFLAG ...   ->  raytracer_amd/variants/libNAME.so   (use with RAYTRACER_AMD_LIB)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$R/raytracer_amd/variants"
cd "$R/raytracer_amd/csrc"
HOST="-std=c++17 -O2 -fPIC -ffp-contract=off -I../../include -I. -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
g++ $HOST "$@" -c bvh_builder.cpp -o /tmp/bvh_builder_$NAME.o
g++ $HOST -DRTA_EXPERIMENTS "$@" -c rt_api.cpp -o /tmp/rt_api_$NAME.o
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c rt_kernels.hip -o /tmp/rt_kernels_$NAME.o
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -fno-slp-vectorize "$@" -c rt_wavefront.hip -o /tmp/rt_wavefront_$NAME.o
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -fno-slp-vectorize "$@" -c rt_cuts.hip -o /tmp/rt_cuts_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/raytracer_amd/variants/lib$NAME.so" /tmp/bvh_builder_$NAME.o rt_host.o /tmp/rt_api_$NAME.o /tmp/rt_kernels_$NAME.o /tmp/rt_wavefront_$NAME.o /tmp/rt_cuts_$NAME.o rt_build.o rt_build_sah.o
echo built "$R/raytracer_amd/variants/lib$NAME.so"
