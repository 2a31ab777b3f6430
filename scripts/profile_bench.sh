#!/bin/bash
# Profiles `bench.py` on the GPU box: one kernel-trace pass for per-kernel time, then
# separate PMC passes (counters never combined with traces).  Results land under
# gpurun_out/prof_<tag>/; summarise them with scripts/summarise_profile.py.
set -eo pipefail
TAG=${1:-r01}
shift || true
EXTRA="$@"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --headline-only $EXTRA"
timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "fetch done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "write done"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq.log" 2>&1
echo "sq done"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d "$OUT/pmc_grbm" -o bench -- $BENCH > "$OUT/pmc_grbm.log" 2>&1
echo "grbm done"
find "$OUT" -type f | head -50
