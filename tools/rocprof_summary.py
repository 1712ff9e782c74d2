#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats result (rocpd sqlite .db) as a text table: per kernel name calls,
total and average duration (us), share.  Usage: tools/rocprof_summary.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel", "-" * 100]
    for name, calls, tot, avg, pct in rows:
        short = name if len(name) < 150 else name[:147] + "..."
        lines.append(f"{calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {short}")
    lines.append("-" * 100)
    lines.append(f"{sum(r[1] for r in rows):7d} {sum(r[2] for r in rows):12.1f} {'':>10} {'':>6}  TOTAL (units as reported by rocprofv3 top_kernels)")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    else:
        sys.stdout.write(out)


if __name__ == "__main__":
    main()
