#!/usr/bin/env python3
"""Reads the dispatch intervals of the render kernel from a `rocprofv3 --kernel-trace` database of
scripts/dev_two_in_flight.py and reports, per phase (one stream / two streams), the span per frame, the kernel's own
duration and how much of a dispatch overlaps the neighbouring one.

    python scripts/summarise_overlap.py gpurun_out/prof_two/trace/x_results.db profiles/r02_1080p_two_in_flight.txt
"""
import sqlite3
import sys


def main():
    db, dst = sys.argv[1], sys.argv[2]
    con = sqlite3.connect(db)
    rows = con.execute("select name, stream_id, start, end from kernels where name like '%rm_queue_kernel4%' order by start").fetchall()
    con.close()
    streams = sorted({r[1] for r in rows})
    lines = [f"kernel: {rows[0][0]}", f"dispatches: {len(rows)} on streams {streams} (start/end in ns from rocprofv3 --kernel-trace)"]
    # phase 1 = the run of dispatches on one stream before the streams start to alternate (warm-up dispatches excluded by position)
    def phase_stats(sel, label):
        n = len(sel)
        span = sel[-1][3] - sel[0][2]
        dur = sum(r[3] - r[2] for r in sel) / n
        ov = 0
        for a, b in zip(sel, sel[1:]):
            ov += max(0, min(a[3], b[3]) - b[2])
        conc = sum(r[3] - r[2] for r in sel) / span
        lines.append(f"{label}: {n} dispatches, span {span / 1e3:.1f} us = {span / n / 1e3:.2f} us per frame; average dispatch duration "
                     f"{dur / 1e3:.2f} us; overlap with the next dispatch {ov / (n - 1) / 1e3:.2f} us on average "
                     f"({100.0 * ov / (n - 1) / dur:.0f} % of a dispatch); average number of dispatches in flight {conc:.2f}")
        return span / n
    # find the first index from which the streams alternate
    alt = next(i for i in range(len(rows) - 3) if rows[i][1] != rows[i + 1][1] and rows[i + 1][1] != rows[i + 2][1] and i > 40)
    single = [r for r in rows[:alt] if r[1] == rows[alt - 1][1]]
    # the single phase is the last contiguous run on one stream before the alternation
    j = len(rows[:alt])
    k = j
    while k > 0 and rows[k - 1][1] == rows[j - 1][1]:
        k -= 1
    single = rows[k:j]
    single = single[20:] if len(single) > 60 else single          # warm-up frames of that stream lead the run
    pair = rows[alt:]
    a = phase_stats(single, "one context, one stream      ")
    b = phase_stats(pair, "two contexts, two streams    ")
    lines.append(f"per-frame span ratio two-in-flight / single: {b / a:.3f}")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
