#!/usr/bin/env python3
"""Development: kernels and copies of the last batches of tools/pipe_trace.py from a rocprofv3 --kernel-trace --memory-copy-trace run, one line
per kernel / copy of >= 5 us: start (us), duration, queue / direction, name.   pipe_timeline.py kernel_trace.csv memory_copy_trace.csv [first_us last_us]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        q = r.get("Stream_Id") or r.get("Queue_Id") or "?"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + q, r["Kernel_Name"].split("(")[0].replace("void ", "")[-34:]))
with open(sys.argv[2]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", r.get("Direction", "?") + " " + r.get("Size", "")))
rows.sort()
# the two-context region: find the whole-batch k_orf_stats launches, take a window in the middle
big = [i for i, r in enumerate(rows) if "k_orf_stats" in r[3] and r[1] - r[0] > 60000]
i0 = big[len(big) // 2 - 3] if len(big) > 8 else 0
i1 = big[len(big) // 2 + 1] if len(big) > 8 else len(rows)
t0 = rows[i0][0]
for s, e, q, n in rows[i0:i1]:
    if e - s < 5000:
        continue
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {q:6s} {n}")
