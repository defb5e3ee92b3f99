#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv rows per (kernel, counter)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[(name[-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
        acc[(name[-40:], "~duration_us")].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
for (k, c), v in sorted(acc.items()):
    if any(s in k for s in ("mp_", "encode", "decode", "train_", "segment_sum")):
        print(f"{k:42s} {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
