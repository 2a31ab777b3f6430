#!/bin/bash
# Kernel trace of the 1080p frame rendered from one context and from two contexts on two streams (scripts/dev_two_in_flight.py).
set -eo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_two
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o two -- python3 $GRAFT_REPO_ROOT/scripts/dev_two_in_flight.py 200 > "$OUT/trace.log" 2>&1
tail -2 "$OUT/trace.log"
timeout -k 10 120 python3 $GRAFT_REPO_ROOT/scripts/dev_two_in_flight.py 200 > "$OUT/plain.log" 2>&1
tail -1 "$OUT/plain.log"
