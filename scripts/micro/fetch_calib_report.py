#!/usr/bin/env python3
"""Puts the counters of the two dispatches of fetch_calib beside the byte counts the program knows."""
import csv
import glob
import os
import sys

root = sys.argv[1]
known = {}
for line in open(os.path.join(root, "known.txt")):
    k, v = line.split()
    known[k] = int(v)
counters = {}
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        kern = "gather" if "gather_kernel" in row["Kernel_Name"] else "stream" if "stream_kernel" in row["Kernel_Name"] else None
        if kern:
            counters.setdefault(kern, {})[row["Counter_Name"]] = float(row["Counter_Value"])
print("FETCH_SIZE calibration on this GPU (scripts/micro/fetch_calib.hip; FETCH_SIZE is in KiB)")
for k in sorted(known):
    print("  %-28s %15d" % (k, known[k]))
for kern in ("stream", "gather"):
    c = counters.get(kern, {})
    print(kern, "dispatch:")
    for name in sorted(c):
        print("  %-28s %15.0f" % (name, c[name]))
    if "FETCH_SIZE" not in c:
        continue
    fetched = c["FETCH_SIZE"] * 1024.0
    if kern == "stream":
        print("  FETCH_SIZE bytes / stream bytes                 %.4f   (the guide's rule: 0.5)" % (fetched / known["stream_bytes"]))
    else:
        print("  FETCH_SIZE bytes / algorithmic bytes (48 B)     %.4f" % (fetched / known["gather_algorithmic_bytes"]))
        print("  FETCH_SIZE bytes / 64-byte sectors touched      %.4f" % (fetched / known["gather_sector64_bytes"]))
        print("  FETCH_SIZE bytes / 128-byte lines touched       %.4f" % (fetched / known["gather_line128_bytes"]))
        if "TCC_EA0_RDREQ_sum" in c:
            print("  fabric read requests per item                   %.4f   (32-byte ones: %.4f)" % (
                c["TCC_EA0_RDREQ_sum"] / known["gather_items"], c.get("TCC_EA0_RDREQ_32B_sum", 0.0) / known["gather_items"]))
