#!/bin/bash
# The GPU suite as the round end runs it, output under gpurun_out/.
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 ${2:-900} python3 -u -m pytest tests -x -q -m gpu --timeout 180 --timeout-method thread ${1:+-k "$1"} > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -6 gpurun_out/gpu_tests.log
exit $rc
