#!/bin/bash
# Developer tool: times a list of cases at 1080p with the in-tree library and with variant libraries.
#   scripts/dev_cases_ab.sh "case1 case2" tag1 tag2 ...
cd "$GRAFT_REPO_ROOT"
CASES=$1; shift
for tag in base "$@"; do
  if [ "$tag" = base ]; then lib=""; else lib="$GRAFT_REPO_ROOT/raymarching_amd/variants/librmhip_$tag.so"; fi
  for c in $CASES; do
    echo -n "[$tag] "; RMHIP_LIB=$lib timeout -k 10 120 python scripts/dev_case.py $c 1920 1080 | grep "kernel 2" || exit 1
  done
done
