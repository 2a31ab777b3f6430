#!/bin/bash
# Developer tool: A/B builds of librmhip that differ in ONE translation unit's -D flags; the other objects are the main build's.
#   scripts/build_tu_variant.sh stream tagA "-DRM_S_SHADE_MIN=16" tagB "-DRM_S_SHADE_MIN=32" ...   ->  raymarching_amd/variants/librmhip_<tag>.so
set -e
cd "$(dirname "$0")/../raymarching_amd/csrc"
mkdir -p ../variants
TU=$1; shift
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $RM_EXTRA_FLAGS"
OTHERS=$(ls rm_k_*.o rm_scratch.o rm_api.o | grep -v "^rm_k_$TU.o$" | tr '\n' ' ')
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c rm_k_$TU.hip -o ../variants/k_${tag}.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/librmhip_$tag.so ../variants/k_${tag}.o $OTHERS &&
    rm -f ../variants/k_${tag}.o && echo built $tag ) &
done
wait
