"""Development (library built with EXTRA=-DWV_PROFILE): per-contig wall time of k_sssp_wave on the benchmark batch — the kernel
lasts as long as its slowest wavefront."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ann = pa.Annotator(flags=("no_certify",))
ann.annotate_flat(seqs)
ann.run(); ann._download_flat()
g = [ann.globals(i) for i in range(n)]
t = np.array([x.rbs_background_count[6] for x in g]) / 100.0  # wall_clock64: 100 MHz -> us
nw = np.array([x.rbs_background_count[7] for x in g])
it = np.array([x.sssp_iters for x in g])
nn = np.array([x.n_node for x in g]); ne = np.array([x.n_edge for x in g])
print("per-contig solver time us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f min %.1f" % (t.mean(), np.median(t), np.percentile(t, 90), np.percentile(t, 99), t.max(), t.min()))
for name, x in (("windows", nw), ("phases", it), ("nodes", nn), ("edges", ne)):
    print("  %-8s mean %.0f max %d  corr with time %.3f" % (name, x.mean(), x.max(), np.corrcoef(x, t)[0, 1]))
o = np.argsort(-t)[:5]
print("  slowest:", [(int(i), float(t[i]), int(nw[i]), int(it[i]), int(nn[i]), int(ne[i])) for i in o])
tp = {"1 setup": np.array([x.gc_max_count[1] for x in g]), "2 gather": np.array([x.gc_max_count[2] for x in g]), "3 stage next": np.array([x.gc_max_count[3] for x in g]),
      "4 phases": np.array([x.rbs_background_count[4] for x in g]), "5 results+stepback": np.array([x.gc_min_count[1] for x in g]), "10 epilogue": np.array([x.rbs_background_count[5] for x in g])}
rest = t * 100.0 - sum(tp.values())
tp["0 wait dma + ring entry (+prologue)"] = rest
for k_, v in sorted(tp.items()):
    print("  tick %-36s mean %.1f us (%.0f%%)  slowest contig %.1f us" % (k_, v.mean() / 100.0, 100.0 * v.mean() / (t.mean() * 100.0), v[o[0]] / 100.0))
tr = np.array([[x.rbs_training_count[j] for j in range(4)] for x in g])
print("  per contig: 64-bit phases %.0f, redone exactly %.0f, exact phases %.0f, rebases %.1f" % tuple(tr.mean(axis=0)))
bgv = np.array([[x.rbs_background_count[j] for j in range(4)] for x in g]) / 100.0
print("  fine ticks 6..9 (gather: cached in-edges / slow conversions / spill list with -DWV_PROFILE_GATHER; phases with _FINE): mean", bgv.mean(axis=0).round(1), "slowest", bgv[o[0]].round(1))
