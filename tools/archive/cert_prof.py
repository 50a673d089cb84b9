"""Development: with a library built with EXTRA="-DCERT_PROFILE -DCERT_DEBUG", where a workgroup of k_certify spends its time
(wall-clock ticks of 10 ns at the section ends, left in the contig's RBS background counters; they are read after the run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ann = pa.Annotator()
ann.annotate_flat(seqs)
ann.run()
names = ["tree edges", "kappa", "sigma", "ranks", "pass over the edges", "queued edges", "corrections"]
rows = []
for i in range(n):
    g = ann.globals(i)
    t = [int(x) for x in g.rbs_background_count[:7]]
    rows.append([t[0]] + [t[k] - t[k - 1] for k in range(1, 7)] + [t[6], g.sssp_iters, g.n_edge, g.sssp_sweeps])
a = np.array(rows, float)
for k, nm in enumerate(names):
    print("%-20s median %6.1f us   p90 %6.1f   max %6.1f" % (nm, np.median(a[:, k]) / 100, np.percentile(a[:, k], 90) / 100, a[:, k].max() / 100))
print("%-20s median %6.1f us   p90 %6.1f   max %6.1f" % ("whole workgroup", np.median(a[:, 7]) / 100, np.percentile(a[:, 7], 90) / 100, a[:, 7].max() / 100))
print("queued edges / edges: median %.3f max %.3f; inexact nodes median %d max %d" % (np.median(a[:, 8] / a[:, 9]), (a[:, 8] / a[:, 9]).max(), np.median(a[:, 10]), a[:, 10].max()))
