#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 --kernel-trace CSV: for the last step of the run, every kernel's start
relative to the step's first kernel, its duration, and the idle time on the critical chain before it
(start minus the latest end among the kernels that started earlier).  Usage: timeline.py kernel_trace.csv"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-28:]))
rows.sort()
# a step begins at k_features
starts = [i for i, r in enumerate(rows) if "k_features" in r[2] and r[1] - r[0] > 200000]  # (whole-batch launches: phx_upload also launches the kernel piece by piece)
# which step: argv[2] = index of the step's k_features launch (default: the last complete step of the trace)
if len(sys.argv) > 2 and len(starts) > int(sys.argv[2]) + 1:
    k = int(sys.argv[2]); step = rows[starts[k]:starts[k + 1]]
else:
    sel = starts[-2] if len(starts) > 1 else 0
    step = rows[sel:starts[-1]] if len(starts) > 1 else rows
t0 = step[0][0]; last_end = t0; idle = 0
for s, e, n in step:
    gap = s - last_end
    if gap > 0: idle += gap
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {n}")
    last_end = max(last_end, e)
print(f"step span {(last_end - t0) / 1e3:.1f} us, idle (no kernel running) {idle / 1e3:.1f} us")
