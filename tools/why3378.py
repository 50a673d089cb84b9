import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, phanotate_amd as pa, certify_probe
seqs = [pa.synth_contig(3378, 50000)]
ann = pa.Annotator(); ann.upload(seqs); ann.run()
print("cert", ann.certified().tolist())
nd, ed, dist = ann.nodes(0), ann.edges(0), ann.dist(0)
path = [int(x) for x in ann.path(0)[0]]
why, ok = certify_probe.certify(nd, ed, dist, path)
print("prototype:", ok, why)
if why.startswith("edge"):
    k = int(why.split()[1]); e = ed[k]
    print("edge", k, "src", e["src"], nd[e["src"]]["pos"], "dst", e["dst"], nd[e["dst"]]["pos"], "w", e["w"], "inexact", e["inexact"], "src on path", int(e["src"]) in path, "dst on path", int(e["dst"]) in path)
    print("n inexact edges", int(ed["inexact"].sum()), "of", len(ed))
