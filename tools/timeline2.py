#!/usr/bin/env python3
"""Two batches in flight, from a rocprofv3 --kernel-trace CSV of bench.py (pipeline region = the last launches of the run): the
kernels of the last steps by queue, and how long one batch's solver (k_sssp_duo / k_sssp_wave) ran while the other batch's throughput kernels ran.
Usage: timeline2.py kernel_trace.csv [steps]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        q = r.get("Stream_Id") or r.get("Queue_Id") or "?"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), q, r["Kernel_Name"].split("(")[0].replace("void ", "")[-30:]))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
starts = [i for i, r in enumerate(rows) if "k_orf_stats" in r[3] and r[1] - r[0] > 60000]  # whole-batch launches (one per step)
sel = rows[starts[-nsteps]:]
t0 = sel[0][0]
qs = sorted({r[2] for r in sel})
print("queues:", qs)
for s, e, q, n in sel:
    if e - s < 8000 and "k_sssp" not in n and "k_features" not in n:
        continue  # (short kernels left out)
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{qs.index(q)}  {n}")
# overlap of a wavefront-solver launch with any k_features / k_edges / k_orf launch that is not of the same step (started after it)
big = [r for r in rows[starts[-min(len(starts), 12)]:] if r[1] - r[0] >= 8000]
sol = [r for r in big if ("k_sssp_duo" in r[3] or "k_sssp_wave" in r[3]) and r[1] - r[0] > 100000]
tot = ov = 0
for s, e, q, n in sol:
    tot += e - s
    cover = sorted((max(s, a), min(e, b)) for a, b, q2, n2 in big if "k_sssp" not in n2 and a < e and b > s)
    cur = s
    for a, b in cover:
        if b > cur:
            ov += b - max(a, cur); cur = b
print(f"wavefront solver launches: {len(sol)}, {tot / len(sol) / 1e3:.1f} us each; another kernel (>= 8 us) ran during {100.0 * ov / tot:.0f} % of that time")
