#!/usr/bin/env python3
"""Reduces the rocprofv3 databases written by scripts/profile_bench.sh to the small
text/JSON summaries kept under profiles/.

    python scripts/summarise_profile.py gpurun_out/prof_r01 profiles/r01 [width height]

HBM traffic follows the gfx950 corrections: FETCH_SIZE (in KB) under-reports wide
coalesced reads by 2x for 16-B/lane accesses; this kernel's reads are at most 4 B/lane
(optional source image) so the raw value is reported next to the doubled one.
"""
import json
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
    h = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    lines, js = [], {"width": w, "height": h}
    tdb = os.path.join(src, "trace", "bench_results.db")
    rows = q(tdb, "select name,total_calls,total_duration,average,percentage from top_kernels")
    lines.append("== rocprofv3 --kernel-trace --stats (durations in us) ==")
    lines.append(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel")
    for n, c, t, a, p in rows:
        lines.append(f"{c:7d} {t:12.1f} {a:10.3f} {p:7.2f}  {n}")
    main_kernel = rows[0][0]
    js["kernel"] = main_kernel
    js["kernel_avg_us"] = rows[0][3]
    js["kernel_calls"] = rows[0][1]
    kr = q(tdb, f"select vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, grid_y, grid_z, workgroup_x "
                f"from kernels where name = '{main_kernel}' limit 1")
    if kr:
        js["resources"] = dict(zip(["vgpr", "agpr", "sgpr", "lds", "scratch", "grid_x", "grid_y", "grid_z", "wg_x"], kr[0]))
        lines.append(f"resources as rocprofv3 records them: {js['resources']}")
        # rocprofv3's vgpr column is the allocation (the metadata's .vgpr_count rounded up to the granule of 8) divided by 2:
        # 116 -> 120 -> 60.  Occupancy follows the code object's own metadata (512 VGPRs per SIMD lane: 4 waves need <= 128).
        try:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import kernel_resources as KR
            lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raymarching_amd", "librmhip.so")
            short = KR.short_name_from_demangled(main_kernel)
            for r in KR.kernel_resources(lib):
                if r["demangled"] == short:
                    js["resources_elf"] = {"vgpr_count": int(r.get("vgpr_count", -1)), "sgpr_count": int(r.get("sgpr_count", -1)),
                                           "scratch": int(r.get("private_segment_fixed_size", -1))}
                    lines.append(f"resources in the code object's metadata (what occupancy follows): {js['resources_elf']}")
        except Exception as e:
            lines.append(f"(code-object metadata not read: {e})")
    counters = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_grbm", "pmc_sq2", "pmc_tcp", "pmc_lds", "pmc_ops"):
        db = os.path.join(src, sub, "bench_results.db")
        if not os.path.exists(db):
            continue
        r = q(db, "select counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                  f"where kernel_name = '{main_kernel}' group by counter_name")
        for name, tot, nd in r:
            counters[name] = tot / max(nd, 1)
    js["counters_per_launch"] = counters
    lines.append("")
    lines.append("== PMC counters, average per launch of the main kernel (summed over XCDs/SEs) ==")
    for k in sorted(counters):
        lines.append(f"{k:28s} {counters[k]:18.1f}")
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        fetch_b, write_b = counters["FETCH_SIZE"] * 1024.0, counters["WRITE_SIZE"] * 1024.0
        js["hbm_fetch_bytes_raw"] = fetch_b
        js["hbm_write_bytes"] = write_b
        js["hbm_bytes_per_launch"] = fetch_b + write_b
        js["algorithmic_bytes_per_launch"] = 9.0 * w * h
        lines.append("")
        lines.append(f"HBM traffic per launch: fetch {fetch_b / 1e6:.2f} MB (raw; x2 = {2 * fetch_b / 1e6:.2f} MB if reads were 16 B/lane), "
                     f"write {write_b / 1e6:.2f} MB; algorithmic framebuffer bytes {9.0 * w * h / 1e6:.2f} MB (RGBA8 + fp32 depth + hit mask)")
    if "SQ_INSTS_VALU" in counters and "SQ_WAVES" in counters:
        lines.append(f"VALU instructions per wave: {counters['SQ_INSTS_VALU'] / counters['SQ_WAVES']:.1f}")
    if "SQ_THREAD_CYCLES_VALU" in counters and "SQ_ACTIVE_INST_VALU" in counters and counters["SQ_ACTIVE_INST_VALU"]:
        util = counters["SQ_THREAD_CYCLES_VALU"] / (counters["SQ_ACTIVE_INST_VALU"] * 64.0)
        js["valu_lane_utilisation"] = util
        lines.append(f"VALU lane utilisation (active lanes per issued VALU cycle): {util * 100:.1f} %")
    if "GRBM_GUI_ACTIVE" in counters:
        clk = counters["GRBM_GUI_ACTIVE"] / 8.0 / (js["kernel_avg_us"] * 1e-6) / 1e9
        js["effective_clock_ghz_estimate"] = clk
        lines.append(f"effective clock estimate GRBM_GUI_ACTIVE/8/t: {clk:.2f} GHz (reads high on dispatches < 0.3 ms)")
    open(dst + "_summary.txt", "w").write("\n".join(lines) + "\n")
    json.dump(js, open(dst + "_summary.json", "w"), indent=1, sort_keys=True)
    if "hbm_bytes_per_launch" in js:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        sha = bench.kernel_source_sha()
        json.dump({"width": w, "height": h, "kernel": main_kernel, "hbm_bytes_per_launch": js["hbm_bytes_per_launch"],
                   "kernel_src_sha": sha, "source": os.path.basename(dst) + "_summary.json"},
                  open(os.path.join(os.path.dirname(dst) or ".", "traffic_latest.json"), "w"), indent=1)
        # one entry per frame size, so that bench.py can quote the counters for whichever size it runs
        by_size_path = os.path.join(os.path.dirname(dst) or ".", "traffic_by_size.json")
        by_size = json.load(open(by_size_path)) if os.path.exists(by_size_path) else {}
        by_size[f"{w}x{h}"] = {"kernel": main_kernel, "hbm_bytes_per_launch": js["hbm_bytes_per_launch"], "kernel_src_sha": sha,
                               "source": os.path.basename(dst) + "_summary.json"}
        json.dump(by_size, open(by_size_path, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
