#!/bin/bash
# Developer tool: one kernel-trace pass, the SQ pass of pmc_run.sh and a pass that counts vector instructions by operation and type
# (fp32 / fp64 add, mul, fma, transcendental) over a python script; every pass a run of its own, never combined with a trace.
#   RM_BODY=3 scripts/pmc_ops_run.sh TAG scripts/dev_body_once.py   ->  gpurun_out/prof_TAG/{trace,pmc_sq,pmc_ops}
set -eo pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/$@"
timeout -s KILL 170 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $CMD > "$OUT/trace.log" 2>&1
echo "trace done"
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o bench -- $CMD > "$OUT/pmc_sq.log" 2>&1
echo "sq done"
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d "$OUT/pmc_ops" -o bench -- $CMD > "$OUT/pmc_ops.log" 2>&1 || echo "ops pass failed (a counter name this build of rocprofv3 does not know?)"
echo done
