#!/bin/bash
# GPU check of the frame-update path: its tests, rank 0's cost with whole expansions and with updates, the N > 1 bench path on one rank.
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "update_frames or compact_transport" --timeout 120 --timeout-method thread > gpurun_out/upd.log 2>&1
rc=$?; tail -5 gpurun_out/upd.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 scripts/dev_root_overlap.py 0 > gpurun_out/overlap2.log 2>&1 || { tail -5 gpurun_out/overlap2.log; exit 1; }
cat gpurun_out/overlap2.log
timeout -k 10 200 python3 bench.py --force-multi --steps 10 --warmup 2 > gpurun_out/bm3.json 2> gpurun_out/bm3.err || { tail -5 gpurun_out/bm3.err; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/bm3.json').read().strip().splitlines()[-1]);assert open('gpurun_out/bm3.json').read().count(chr(10))==1;print(d['ms_per_step'],d['transport_only']['ms_per_step'],d['no_gather']['ms_per_step'])"
