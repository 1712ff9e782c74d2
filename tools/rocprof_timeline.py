#!/usr/bin/env python3
"""One training step of a rocprofv3 --kernel-trace --output-format csv run as a timeline: kernel, duration, gap to the previous kernel's end.
Usage: tools/rocprof_timeline.py <dir with *kernel_trace.csv> [out.txt]   [step index]   (a step = from one k_pixel_losses launch to the next)"""
import csv
import glob
import os
import sys


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("k_pixel_losses")]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 6            # which step (the timed loop follows 3 warm-up steps; later ones belong to bench.py's probes)
    a, b = marks[k], marks[k + 1]
    out = [f"{'t_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel"]
    t0 = rows[a][0]
    busy = gap_total = 0.0
    for i in range(a, b):
        s, e, n = rows[i]
        gap = (s - rows[i - 1][1]) / 1e3
        busy += (e - s) / 1e3
        gap_total += max(gap, 0.0)
        out.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {n[:110]}")
    out.append(f"step: {(rows[b][0] - t0) / 1e3:.1f} us wall, {busy:.1f} us in {b - a} kernels, {gap_total:.1f} us of gaps")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
