#!/bin/bash
# Runs in the repository after scripts/regen_profiles.sh (both parts): reduces gpurun_out/prof_<TAG>_* to the files kept under
# profiles/ (summaries, the march loops' disassembly, HBM traffic per frame size with the hash of the kernel sources).
set -eo pipefail
TAG=${1:-r03}
cd "$(dirname "$0")/.."
G=gpurun_out
python3 scripts/summarise_profile.py $G/prof_${TAG}_4k profiles/${TAG}_4k 3840 2160 > /dev/null
python3 scripts/summarise_profile.py $G/prof_${TAG}_1080p profiles/${TAG}_1080p 1920 1080 > /dev/null
python3 scripts/summarise_profile.py $G/prof_${TAG}_4k_fast profiles/${TAG}_4k_fast_tmp 3840 2160 > /dev/null
# the tolerance kernel's traffic is not the exact kernel's: keep the 4K entry of traffic_by_size.json from the strict pass
python3 scripts/summarise_profile.py $G/prof_${TAG}_4k profiles/${TAG}_4k 3840 2160 > /dev/null
mv profiles/${TAG}_4k_fast_tmp_summary.txt profiles/${TAG}_4k_fast_summary.txt; mv profiles/${TAG}_4k_fast_tmp_summary.json profiles/${TAG}_4k_fast_summary.json
python3 scripts/summarise_profile.py $G/prof_${TAG}_4k_sq2 profiles/${TAG}_4k_wait_counters 3840 2160 > /dev/null
python3 scripts/summarise_profile.py $G/prof_${TAG}_stress256_skip profiles/${TAG}_stress256_skip_1080p 1920 1080 > /dev/null
python3 scripts/summarise_profile.py $G/prof_${TAG}_menger profiles/${TAG}_menger_1080p 1920 1080 > /dev/null
python3 scripts/summarise_profile.py $G/prof_${TAG}_mandelbulb profiles/${TAG}_mandelbulb_1080p 1920 1080 > /dev/null
python3 - "$TAG" <<'PY'
import sqlite3, sys
tag = sys.argv[1]
out = []
for mode in ("directional", "perspective"):
    db = f"gpurun_out/prof_{tag}_lookup/trace_{mode}/bench_results.db"
    rows = sqlite3.connect(db).execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    out.append(f"== {mode} shadow map, 3840x2160: depth pass from the light + frame with the lookup (rocprofv3 --kernel-trace --stats, us) ==")
    out.append(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel")
    out += [f"{c:7d} {t:12.1f} {a:10.3f} {p:7.2f}  {n}" for n, c, t, a, p in rows]
    out.append("")
open(f"profiles/{tag}_lookup_4k_summary.txt", "w").write("\n".join(out))
PY
python3 scripts/summarise_overlap.py $G/prof_${TAG}_two4k/trace/two_results.db profiles/${TAG}_4k_two_in_flight.txt > /dev/null
python3 scripts/loop_isa.py 'rm_queue_kernel4_plain<rot=0, scl=0, 1>' --listing --min-valu 20 > profiles/${TAG}_4k_loop_isa.txt
ls -la profiles | grep ${TAG}
