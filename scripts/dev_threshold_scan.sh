#!/bin/bash
# Developer tool: the heavy-tile threshold scanned over library builds on the static default pose.
#   scripts/dev_threshold_scan.sh W H lib1.so lib2.so ...
W=$1; H=$2; shift 2
for thr in 8 12 16 24 32 48; do
  echo "== threshold $thr"
  timeout -k 10 120 python -u scripts/dev_ab_lib.py $W $H "$@" _RmHeavyTileThreshold=$thr steps=200 2>&1 | grep "round 2" || exit 1
done
