#!/usr/bin/env python3
"""Summaries of a rocprofv3 results database (rocpd SQLite, the default output of ROCm 7.2):
  rocpd_summary.py stats <results.db> <out.csv>          kernel-trace: per-kernel calls / total / average (us)
  rocpd_summary.py pmc <results.db> <out.csv> [needle]   counter collection: per-kernel, per-counter average per launch
Kernel names are shortened to the function name (template arguments kept)."""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*>)?)", name)
    return m.group(1) if m else name[:80]


def main():
    mode, db, out = sys.argv[1:4]
    con = sqlite3.connect(db)
    if mode == "stats":
        rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            for n, c, t, a, p in rows:
                w.writerow([short(n), c, round(t, 3), round(a, 3), round(p, 3)])
    else:
        needle = sys.argv[4] if len(sys.argv) > 4 else ""
        cols = [d[1] for d in con.execute("pragma table_info(counters_collection)")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
        rows = con.execute(f"select {name_col}, {cnt_col}, value from counters_collection").fetchall()
        agg = {}
        for n, c, v in rows:
            if needle and needle not in n:
                continue
            k = (short(n), c)
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(v)
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Counter", "Launches", "AveragePerLaunch", "Total"])
            for (n, c), (k, s) in sorted(agg.items()):
                w.writerow([n, c, k, round(s / k, 3), round(s, 3)])
    print(open(out).read())


if __name__ == "__main__":
    main()
