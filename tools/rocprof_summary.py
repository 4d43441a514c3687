#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel launches, total / average duration.
usage: tools/rocprof_summary.py gpurun_out/prof/x_results.db [--by-grid] > profiles/<name>.txt
--by-grid: one row per (kernel, grid): the training step's kernels by layer shape."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
by_grid = "--by-grid" in sys.argv[2:]
gcols = [c for c in ("grid_x", "grid_y", "grid_z", "workgroup_x") if c in cols] if by_grid else []
if by_grid and not gcols:
    print("columns:", cols)
key = name_col + "".join(f" || ' ' || {c}" for c in gcols)
rows = db.execute(f"select {key}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                  f"from kernels group by {key} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*\)(?=( \d+)*$)", "", n)
    n = n.replace("void rldm::", "").replace("rldm::", "")
    return n[:86]


print(f"{'kernel':88s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share':>6s}")
for n, c, s, a, mn, mx in rows:
    print(f"{short(n):88s} {c:7d} {s / 1e6:10.3f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {s / tot:6.1%}")
print(f"{'TOTAL':88s} {sum(r[1] for r in rows):7d} {tot / 1e6:10.3f}")
