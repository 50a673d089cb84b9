#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (last dispatch of each kernel)."""
import collections
import csv
import sys

for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    d = collections.OrderedDict()
    for r in rows:
        k = r["Kernel_Name"][:44]
        if "rocclr" in k:
            continue
        e = d.setdefault(k, {})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
        e["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(path)
    for k, v in d.items():
        print("  %-46s %8.0f us  " % (k, v.pop("dur_us")) + "  ".join("%s=%.4g" % kv for kv in sorted(v.items())))
