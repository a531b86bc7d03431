#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (…_results.db) as text: per-kernel duration statistics
(the `--kernel-trace --stats` view) and, when the run collected PMC counters, the per-kernel
average of every counter.  Used to produce the files committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof/run_results.db > profiles/r01_xyz.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# source: %s" % path)
    rows = cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("## kernel trace stats (durations in us)")
    print("%-8s %10s %9s %9s %9s %6s | %4s %4s %4s %6s %7s | %9s %5s | %s" % (
        "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid_x", "wg_x", "kernel"))
    for r in rows:
        name, n, s, a, mn, mx, vg, ag, sg, lds, scr, gx, wx = r
        print("%-8d %10.1f %9.2f %9.2f %9.2f %6.2f | %4s %4s %4s %6s %7s | %9s %5s | %s" % (
            n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot, vg, ag, sg, lds, scr, gx, wx, name))
    try:
        # (per kernel and GRID: round 6's launches that serve a group of batches are the same kernel on another grid)
        prow = cur.execute(
            "select kernel_name || '  [grid ' || grid_size_x || ' x ' || workgroup_size_x || ']', counter_name, count(*), avg(value), sum(value) "
            "from counters_collection group by kernel_name, grid_size_x, workgroup_size_x, counter_name "
            "order by kernel_name, grid_size_x, counter_name").fetchall()
    except sqlite3.Error:
        try:
            prow = cur.execute(
                "select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            prow = []
    if prow:
        print("\n## PMC counters (per dispatch average; summed over all instances of the counter's block)")
        last = None
        for k, c, n, a, s in prow:
            if k != last:
                print("\n%s" % k)
                last = k
            print("    %-40s dispatches=%-5d avg=%.1f" % (c, n, a))
    # how much of the kernels' time overlaps another kernel (two batches in flight: streams interleave)
    try:
        iv = cur.execute("select start, end, name from kernels order by start").fetchall()
    except sqlite3.Error:
        iv = []
    if len(iv) > 1:
        busy = overlap = 0
        span0, span1 = iv[0][0], max(r[1] for r in iv)
        cur_end = iv[0][0]
        n_over = 0
        for (a, b, _), nxt in zip(iv, iv[1:] + [None]):
            busy += b - a
            if nxt is not None and nxt[0] < b:
                n_over += 1
                overlap += min(b, nxt[1]) - nxt[0]
        print("\n## concurrency: %d kernels over a span of %.1f us; summed kernel time %.1f us; %d launches start before the "
              "previous kernel ended (%.1f us of pairwise overlap); span / launches = %.2f us per launch"
              % (len(iv), (span1 - span0) / 1e3, busy / 1e3, n_over, overlap / 1e3, (span1 - span0) / 1e3 / len(iv)))
    # the same per kernel VARIANT (name x grid x block): how many of its launches run while another kernel is still running, and the
    # steady-state distance between its launches (round 6: the timed region's launches serve a group of batches each and alternate
    # between two streams -- this is where the trace shows what overlaps)
    try:
        kv = cur.execute("select name, grid_x, workgroup_x, start, end from kernels order by start").fetchall()
    except sqlite3.Error:
        kv = []
    if len(kv) > 1:
        import statistics
        groups = {}
        run_end = 0  # the latest end of any kernel that started earlier
        for i, (name, gx, wx, a, b) in enumerate(kv):
            g = groups.setdefault((name, gx, wx), {"n": 0, "dur": [], "over_n": 0, "over": [], "starts": []})
            g["n"] += 1
            g["dur"].append(b - a)
            g["starts"].append(a)
            if i and run_end > a:
                g["over_n"] += 1
                g["over"].append(min(run_end, b) - a)
            run_end = max(run_end, b)
        print("\n## overlap by kernel variant (launches >= 20 only): launches | avg us | launches that start while an earlier kernel still runs "
              "(mean us of such overlap) | median distance between consecutive launches of the variant, us")
        for (name, gx, wx), g in sorted(groups.items(), key=lambda kv_: -sum(kv_[1]["dur"])):
            if g["n"] < 20:
                continue
            d = [y - x for x, y in zip(g["starts"], g["starts"][1:])]
            print("%-6d %9.2f   %5d (%.2f)   %9.2f | grid %s x %s | %s" % (
                g["n"], statistics.mean(g["dur"]) / 1e3, g["over_n"], (statistics.mean(g["over"]) / 1e3) if g["over"] else 0.0,
                statistics.median(d) / 1e3 if d else 0.0, gx, wx, name[:110]))
    try:
        mc = cur.execute("select count(*), sum(size), avg(duration) from memory_copies").fetchone()
        if mc and mc[0]:
            print("\n## memory copies: n=%d bytes=%s avg_us=%.2f" % (mc[0], mc[1], (mc[2] or 0) / 1e3))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
