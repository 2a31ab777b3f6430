#!/bin/bash
# Developer tool: scripts/dev_history.py (what the tile-queue ordering is worth under camera motion) over several library
# builds under raymarching_amd/, at 3840x2160 and 1920x1080.  Edit the list of builds below.
for lib in librmhip.so variants/librmhip_dil1.so variants/librmhip_dil2.so variants/librmhip_dil4.so; do
  for sz in "3840 2160" "1920 1080"; do
    set -- $sz
    echo "== $lib $1x$2"
    RMHIP_LIB=$PWD/raymarching_amd/$lib RM_W=$1 RM_H=$2 timeout -k 10 120 python -u scripts/dev_history.py 2>&1 | grep "^(" || exit 1
  done
done
