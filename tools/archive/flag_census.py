"""How many edges does the fp64 pipeline flag (the work list of k_refine), by kind, on benchmark contigs?   python tools/flag_census.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
a = pa.Annotator(flags=("no_certify",)); a.upload(seqs); a.run()
tot = {"edges": 0, "orf": 0, "overlap": 0, "gap": 0}
per = []
for i in range(n):
    nd, ed = a.nodes(i), a.edges(i)
    f = ed["inexact"] != 0
    s, d = ed["src"], ed["dst"]
    ts, td, fs, fd, ps, pd = nd["type"][s], nd["type"][d], nd["frame"][s], nd["frame"][d], nd["pos"][s], nd["pos"][d]
    orf = (ts < 2) & (td < 2) & (fs == fd) & (((fs > 0) & (ts == 0) & (td == 1)) | ((fs < 0) & (ts == 1) & (td == 0)))
    ov = ~orf & (ts < 2) & (td < 2) & (ps > pd)
    tot["edges"] += len(ed); tot["orf"] += int((f & orf).sum()); tot["overlap"] += int((f & ov).sum()); tot["gap"] += int((f & ~orf & ~ov).sum())
    per.append(int(f.sum()))
print(tot, "flagged per contig: min %d median %d max %d" % (min(per), int(np.median(per)), max(per)))
