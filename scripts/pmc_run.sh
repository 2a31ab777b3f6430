#!/bin/bash
# Developer tool: one rocprofv3 kernel-trace pass and three counter passes (never combined with a trace) over a python script.
#   scripts/pmc_run.sh TAG scripts/dev_stress_once.py   ->  gpurun_out/prof_TAG/{trace,pmc_sq,pmc_sq2,pmc_lds}
set -eo pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/$@"
timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $CMD > "$OUT/trace.log" 2>&1
echo "trace done"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o bench -- $CMD > "$OUT/pmc_sq.log" 2>&1
echo "sq done"
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH -d "$OUT/pmc_sq2" -o bench -- $CMD > "$OUT/pmc_sq2.log" 2>&1
echo "sq2 done"
timeout -s KILL 170 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 -d "$OUT/pmc_lds" -o bench -- $CMD > "$OUT/pmc_lds.log" 2>&1 || echo "lds pass failed (a counter name this build of rocprofv3 does not know?)"
echo done
