#!/bin/bash
# Developer tool: runs bench.py once per variant library (interleaved, 2 rounds) and prints kernel time.
cd "$GRAFT_REPO_ROOT"
ARGS=${AB_ARGS:-"--no-cpu-baseline --steps 300 --warmup 30"}
for round in 1 2; do
  for tag in base "$@"; do
    if [ "$tag" = base ]; then lib=""; else lib="$GRAFT_REPO_ROOT/raymarching_amd/variants/librmhip_$tag.so"; fi
    RMHIP_LIB=$lib timeout -k 10 120 python bench.py $ARGS | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s round $round  kernel %.1f us  %.1f Mrays/s  frac %.3f' % ('$tag', j['roofline']['kernel_ms']*1e3, j['value'], j['roofline']['frac']))
" || exit 1
  done
done
