#!/bin/bash
# like build_variants.sh but also recompiles rm_api.cpp with the flags (for macros the host side sees)
set -e
cd "$(dirname "$0")/../raymarching_amd/csrc"
mkdir -p ../variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c rm_kernels.hip -o ../variants/k_$tag.o &&
    /opt/rocm/bin/hipcc $FLAGS $defs -c rm_api.cpp -o ../variants/a_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/librmhip_$tag.so ../variants/k_$tag.o ../variants/a_$tag.o && echo built $tag ) &
done
wait
