#!/bin/bash
# Developer tool: builds A/B variants of librmhip (raymarching_amd/variants/librmhip_<tag>.so) from -D flag sets.
#   scripts/build_variants_full.sh tag1 "-DRM_BWX=1 -DRM_BWY=1" tag2 "-DRM_Q_MIN_WAVES=3" ...
# (this one also recompiles rm_api.cpp with the flags, for macros the host side sees)
set -e
cd "$(dirname "$0")/../raymarching_amd/csrc"
mkdir -p ../variants
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
[ -f rm_api.o ] || /opt/rocm/bin/hipcc $FLAGS -c rm_api.cpp -o rm_api.o
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  objs=""
  for tu in tile tile_fx queue stream; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c rm_k_$tu.hip -o ../variants/k_${tag}_$tu.o &
    objs="$objs ../variants/k_${tag}_$tu.o"
  done
  wait
  /opt/rocm/bin/hipcc $FLAGS $defs -c rm_api.cpp -o ../variants/a_$tag.o; api=../variants/a_$tag.o
  # the other translation units come from the main build as they are
  others=$(ls rm_k_*.o rm_scratch.o | grep -v -e "^rm_k_tile.o$" -e "^rm_k_tile_fx.o$" -e "^rm_k_queue.o$" -e "^rm_k_stream.o$" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/librmhip_$tag.so $objs $others $api && echo built $tag
  rm -f $objs
done
