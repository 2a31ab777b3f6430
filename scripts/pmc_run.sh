#!/bin/bash
# Developer tool: one rocprofv3 kernel-trace pass and two SQ counter passes over a python script.
set -eo pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/$@"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $CMD > "$OUT/trace.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o bench -- $CMD > "$OUT/pmc_sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH -d "$OUT/pmc_sq2" -o bench -- $CMD > "$OUT/pmc_sq2.log" 2>&1
echo done
