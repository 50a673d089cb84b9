"""Per-segment solver records of a lone contig (phx_seg_stats): windows, nodes swept, time, phases, step-backs.  python tools/seg_stats.py lambda|t4"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import phanotate_amd as pa
from conftest import load_golden

name = sys.argv[1] if len(sys.argv) > 1 else "t4"
seq = load_golden({"lambda": "NC_001416.1", "t4": "NC_000866.1"}[name])[2]
a = pa.Annotator()
a.annotate_flat([seq])
for _ in range(4):
    a.run()
st = a.seg_stats(0)
print("segment windows nodes[first,end) swept  us  phases packs step-backs status")
for s, r in enumerate(st):
    if r[0] == 0 and r[3] == 0:
        continue
    print("%3d %5d  [%5d,%5d) %5d  %6.1f %5d %4d %3d  %d" % (s, r[0] & 0xffff, r[2], r[3], r[3] - r[2], r[4] * 0.01, r[5], r[6], r[7], r[1]))
a.close()
