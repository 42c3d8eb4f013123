#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: sum of each counter over
dispatches and the per-dispatch mean."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            short = k.split("(")[0].replace("void ", "")
            if "rtc::" not in short:
                continue
            c = row["Counter_Name"]
            acc[short][c] += float(row["Counter_Value"])
            cnt[short][c] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = cnt[k][c]
        print("   %-36s dispatches %3d  mean %.6g" % (c, n, acc[k][c] / n))
