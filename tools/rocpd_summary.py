#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats [--pmc ...]) into small CSVs for profiles/.

    python tools/rocpd_summary.py <results.db> <out_prefix>
writes <out_prefix>_kernel_stats.csv (calls, total/avg duration per kernel) and, when counters were collected,
<out_prefix>_pmc.csv (per kernel: dispatches, mean counter value per dispatch).
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    return name if len(name) < 160 else name[:157] + "..."


def main(db_path, prefix):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    # per-dispatch durations for the median / minimum (a kernel whose first launch does extra work -- the assembly
    # integrates the cached blocks once -- has a mean that no single launch shows)
    per = {}
    try:
        for name, dur in cur.execute('select name, ("end" - start) from kernels'):
            per.setdefault(name, []).append(dur)
    except sqlite3.Error as e:
        print("no per-dispatch view:", e)
        print("objects:", [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")][:60])
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent", "median_ms", "min_ms"])
        for name, calls, tot, avg, pct in rows:
            d = sorted(per.get(name, []))
            med = f"{d[len(d) // 2] / 1e6:.3f}" if d else ""
            mn = f"{d[0] / 1e6:.3f}" if d else ""
            w.writerow([short(name), calls, f"{tot / 1e3:.3f}", f"{avg / 1e3:.3f}", f"{pct:.2f}", med, mn])
    # per (kernel, grid size): the same kernel runs on every multigrid level
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else ("grid_size" if "grid_size" in cols else None))
        if gcol is None:
            print("kernels view columns:", cols)
        else:
            g = {}
            for name, grid, dur in cur.execute(f'select name, {gcol}, ("end" - start) from kernels'):
                g.setdefault((name, grid), []).append(dur)
            with open(prefix + "_by_grid.csv", "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["kernel", "grid", "calls", "total_ms", "avg_ms", "median_ms", "min_ms", "max_ms"])
                for (name, grid), d in sorted(g.items(), key=lambda kv: -sum(kv[1])):
                    if sum(d) / 1e6 < 0.5:
                        continue
                    ds = sorted(d)
                    w.writerow([short(name)[:90], grid, len(d), f"{sum(d) / 1e6:.3f}", f"{sum(d) / len(d) / 1e6:.4f}",
                                f"{ds[len(ds) // 2] / 1e6:.4f}", f"{ds[0] / 1e6:.4f}", f"{ds[-1] / 1e6:.4f}"])
    except sqlite3.Error as e:
        print("no per-grid view:", e)
    try:
        q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
             "group by kernel_name, counter_name")
        prow = list(cur.execute(q))
    except sqlite3.Error as e:
        prow = []
        print("no counters:", e)
    if prow:
        # per-dispatch values too: the median of a kernel whose first launch does extra work is the steady-state launch
        per_d = {}
        try:
            for name, ctr, disp, val in cur.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection "
                                                    "group by kernel_name, counter_name, dispatch_id"):
                per_d.setdefault((name, ctr), []).append(val)
        except sqlite3.Error as e:
            print("no per-dispatch counters:", e)
        with open(prefix + "_pmc.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "sum", "median_per_dispatch", "min_per_dispatch"])
            for name, ctr, n, mean, tot in prow:
                d = sorted(per_d.get((name, ctr), []))
                nd = max(len(d), 1)
                # counters_collection holds one row per (dispatch, counter instance): mean over rows is not per dispatch
                w.writerow([short(name), ctr, len(d) or n, f"{tot / nd:.6g}", f"{tot:.6g}", f"{d[len(d) // 2]:.6g}" if d else "", f"{d[0]:.6g}" if d else ""])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
