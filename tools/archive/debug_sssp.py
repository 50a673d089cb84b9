#!/usr/bin/env python3
"""Run a few synthetic contigs through libphx and compare path / distance with the oracle (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
from oracle import oracle

def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    seqs = [pa.synth_contig(100 + i, L) for i in range(n)]
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    for i, s in enumerate(seqs):
        g = ann.globals(i)
        o = oracle.run(s)
        p, dist = ann.path(i)
        nd = ann.nodes(i)
        ok = len(p) == len(o["path"]) and np.array_equal(nd["refidx"][p], o["path"])
        print("contig %d: status %d V %d E %d limbs %d sweeps %d iters %d path %d/%d dist_ok %s path_ok %s" % (
            i, g.status, g.n_node, g.n_edge, g.n_limbs, g.sssp_sweeps, g.sssp_iters, len(p), len(o["path"]), dist == o["path_dist"], ok))
if __name__ == "__main__":
    main()
