#!/bin/bash
# Developer tool: HBM write traffic of the 4K frame with and without the hit-mask plane (two counter passes).
set -eo pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/planes; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/with -o t -- python3 $GRAFT_REPO_ROOT/scripts/dev_planes_traffic.py > $OUT/with.log 2>&1
export RM_NO_MASK=1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/without -o t -- python3 $GRAFT_REPO_ROOT/scripts/dev_planes_traffic.py > $OUT/without.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import sqlite3, glob
for tag in ("with", "without"):
    db = glob.glob(f"gpurun_out/planes/{tag}/*results.db")[0]
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = con.execute(f"select s.kernel_name, sum(e.value), count(distinct d.id) from {pmc} e join {info} i on e.pmc_id=i.id join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id where i.name='WRITE_SIZE' group by s.kernel_name order by 2 desc").fetchall()
    name, total, n = rows[0]
    print(tag, name[:60], "launches", n, "WRITE_SIZE per launch: %.1f KB = %.2f MB" % (total / n, total / n * 1024 / 1e6))
P
