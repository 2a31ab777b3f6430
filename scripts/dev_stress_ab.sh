#!/bin/bash
# Developer tool: stress scene (tile kernel) with the in-tree library and variant libraries.
cd "$GRAFT_REPO_ROOT"
for tag in base "$@" base "$@"; do
  if [ "$tag" = base ]; then lib=""; else lib="$GRAFT_REPO_ROOT/raymarching_amd/variants/librmhip_$tag.so"; fi
  echo -n "[$tag] "; RMHIP_LIB=$lib timeout -k 10 120 python scripts/dev_stress.py | grep "stress256 kernel 1" || exit 1
done
