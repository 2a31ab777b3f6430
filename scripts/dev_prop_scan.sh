#!/bin/bash
# Developer tool: one float property of the material scanned on the static default pose.
#   scripts/dev_prop_scan.sh W H _RmStaticFraction "-1 0.3 0.5 0.7"
W=$1; H=$2; P=$3
for v in $4; do
  echo -n "$P $v: "
  timeout -k 10 120 python -u scripts/dev_ab_lib.py $W $H raymarching_amd/librmhip.so $P=$v steps=200 2>&1 | grep "round 2" || exit 1
done
