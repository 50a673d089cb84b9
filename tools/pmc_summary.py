#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel: the longest dispatch of each kernel (the full-batch launch —
phx_upload also launches k_features piece by piece; among equals the last).

With --json FILE also writes {kernel: {counter: value, ...}} plus derived `hbm_bytes` per launch:
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import json
import sys

args = sys.argv[1:]
jpath = None
if args and args[0] == "--json":
    jpath, args = args[1], args[2:]
allk = collections.OrderedDict()
for path in args:
    rows = list(csv.DictReader(open(path)))
    d = collections.OrderedDict()
    best = {}
    for r in rows:
        k = r["Kernel_Name"]
        if "rocclr" in k:
            continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        did = r.get("Dispatch_Id")
        if k not in best or dur >= best[k][0] or did == best[k][1]:
            if k not in best or did != best[k][1]:
                d[k] = {}
            best[k] = (max(dur, best[k][0]) if k in best and did == best[k][1] else dur, did)
            d[k][r["Counter_Name"]] = float(r["Counter_Value"])
            d[k]["dur_us"] = dur
    print(path)
    for k, v in d.items():
        dur = v.pop("dur_us")
        print("  %-46s %8.0f us  " % (k[:46], dur) + "  ".join("%s=%.4g" % kv for kv in sorted(v.items())))
        allk.setdefault(k, {}).update(v)
if jpath:
    for k, v in allk.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    json.dump(allk, open(jpath, "w"), indent=1)
