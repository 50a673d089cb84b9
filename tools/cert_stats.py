"""Development: with a library built with EXTRA=-DCERT_DEBUG, k_certify leaves its queue fill (edges that needed the exact check) and the
number of inexact tree edges in the solver's counters: their distribution over the benchmark batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ann = pa.Annotator()
ann.annotate_flat(seqs)
rows = []
for i in range(n):
    g = ann.globals(i)
    rows.append((g.n_node, g.n_edge, g.sssp_iters, g.sssp_sweeps, g.certified))
a = np.array(rows, float)
fr = a[:, 2] / a[:, 1]
print("contigs", n, "certified", int(a[:, 4].sum()))
print("queued / edges: median %.3f p90 %.3f max %.3f;  queued / nodes: median %.2f p90 %.2f max %.2f (capacity 2)" % (np.median(fr), np.percentile(fr, 90), fr.max(), np.median(a[:, 2] / a[:, 0]), np.percentile(a[:, 2] / a[:, 0], 90), (a[:, 2] / a[:, 0]).max()))
print("inexact tree edges / nodes: median %.3f max %.3f" % (np.median(a[:, 3] / a[:, 0]), (a[:, 3] / a[:, 0]).max()))
