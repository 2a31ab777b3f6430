#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash scripts/regen_profiles.sh TAG PART): every rocprofv3 pass the kept profiles come from, under
# gpurun_out/prof_<TAG>_*.  PART a: the headline frame at 4K / 1080p and the tolerance kernel (trace + FETCH / WRITE / SQ / GRBM);
# PART b: stress scene (trace + SQ + LDS passes), menger and mandelbulb close-ups (trace + SQ + per-type instruction counts),
# the shadow-map lookup frames and two 4K frames in flight (traces).  scripts/collect_profiles.sh TAG turns them into profiles/.
set -eo pipefail
TAG=${1:-r03}; PART=${2:-a}
S=$GRAFT_REPO_ROOT/scripts
if [ "$PART" = a ]; then
  bash $S/profile_round.sh $TAG
  bash $S/pmc_run.sh ${TAG}_4k_sq2 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --headline-only
else
  RM_KERNEL=0 bash $S/pmc_run.sh ${TAG}_stress256_skip scripts/dev_stress_once.py
  RM_BODY=1 bash $S/pmc_ops_run.sh ${TAG}_menger scripts/dev_body_once.py
  RM_BODY=3 bash $S/pmc_ops_run.sh ${TAG}_mandelbulb scripts/dev_body_once.py
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_lookup; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
  RM_MODE=2 timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d $OUT/trace_directional -o bench -- python3 $S/dev_lookup_once.py > $OUT/directional.log 2>&1
  RM_MODE=1 timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d $OUT/trace_perspective -o bench -- python3 $S/dev_lookup_once.py > $OUT/perspective.log 2>&1
  echo "lookup done"
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_two4k; mkdir -p $OUT
  timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d $OUT/trace -o two -- python3 $S/dev_two_in_flight.py 100 3840 2160 > $OUT/trace.log 2>&1
  echo "two in flight done"
fi
