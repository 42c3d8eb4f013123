#!/usr/bin/env python3
"""Per-dispatch PMC table of the trace kernels: one line per dispatch, counters side by side."""
import csv, glob, os, sys
from collections import defaultdict, OrderedDict
root = sys.argv[1]
rows = OrderedDict()
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    seq = defaultdict(int)
    with open(f) as fh:
        per = OrderedDict()
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "trace_kernel" not in k:
                continue
            per.setdefault((row["Dispatch_Id"], k), {})[row["Counter_Name"]] = float(row["Counter_Value"])
        for i, ((d, k), c) in enumerate(per.items()):
            rows.setdefault((i, k), {}).update(c)
names = sorted({c for v in rows.values() for c in v})
print("idx kernel " + " ".join(names))
for (i, k), c in rows.items():
    print(i, k.replace("rtc::trace_kernel", "tk").replace("rtc::", ""), " ".join("%.4g" % c.get(n, float("nan")) for n in names))
