#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per (kernel, counter) the mean value per dispatch.
Usage: tools/pmc_parse.py <dir-with-*_counter_collection.csv> [...]   (kernel names are cut at the first '(')"""
import csv
import glob
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
for d in sys.argv[1:]:
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0][-60:]
            k = (name, row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    for (name, ctr), (tot, n) in sorted(acc.items()):
        print(f"{d}: {name:60s} {ctr:28s} mean/dispatch {tot / n:16.1f}  dispatches {n}")
