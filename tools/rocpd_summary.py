#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) as a per-kernel table: calls, total / average / min / max
duration.  Usage: rocpd_summary.py results.db [> profiles/rNN_kernel_stats.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [])
        a.append((e - s) / 1e3)  # ns -> us
    total = sum(sum(v) for v in agg.values())
    print(f"# source: {path}  ({len(rows)} kernel dispatches, {total / 1e3:.3f} ms total)")
    print(f"{'kernel':100s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name[:100]:100s} {len(v):7d} {sum(v):12.1f} {sum(v) / len(v):10.2f} {min(v):10.2f} {max(v):10.2f} "
              f"{100 * sum(v) / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
