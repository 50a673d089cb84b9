"""Development (library built with EXTRA=-DDUO_PROFILE): where the two wavefronts of k_sssp_duo spend a contig's time on the benchmark
batch.  Solver: waiting for a pack / taking it (header, ring entry, lane records, copy) / phases / results + step-back test.
Feeder: part 1 (loads + conversion in registers) / waiting for the solver's acknowledgement / part 2 (side list, spill) / writing the pack."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
if len(sys.argv) > 1 and sys.argv[1] in ("lambda", "t4"):  # a lone genome (tests/golden)
    import gzip
    f = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", {"lambda": "NC_001416.1", "t4": "NC_000866.1"}[sys.argv[1]] + ".fasta.gz")
    seqs = ["".join(l.strip() for l in gzip.open(f, "rt").read().split("\n")[1:])]
    n = 1
else:
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ann = pa.Annotator(flags=("no_certify",))
ann.annotate_flat(seqs)
ann.run(); ann.run(); ann._download_flat()
g = [ann.globals(i) for i in range(n)]
t = np.array([x.rbs_background_count[6] for x in g]) / 100.0  # wall_clock64: 100 MHz -> us
packs = np.array([x.rbs_background_count[7] for x in g])
it = np.array([x.sssp_iters for x in g])
print("solver wavefront per contig us: mean %.1f median %.1f p90 %.1f max %.1f min %.1f; packs %.1f, phases %.1f" % (t.mean(), np.median(t), np.percentile(t, 90), t.max(), t.min(), packs.mean(), it.mean()))
o = np.argsort(-t)[:3]
sp = np.array([[x.gc_max_count[j] for j in range(4)] for x in g]) / 100.0
for j, nm in enumerate(("wait for the pack", "take the pack", "phases", "results + step-back test")):
    print("  solver %-26s mean %6.1f us (%4.1f%%)  slowest contig %6.1f" % (nm, sp[:, j].mean(), 100 * sp[:, j].mean() / t.mean(), sp[o[0], j]))
ft = np.array([[x.rbs_training_count[j] for j in range(6)] for x in g]).astype(float)
for j, nm in enumerate(("part 1 (loads + convert)", "wait for the acknowledgement", "part 2 (side list, spill)", "write the pack")):
    print("  feeder %-28s mean %6.1f us  slowest contig %6.1f" % (nm, ft[:, j].mean() / 100.0, ft[o[0], j] / 100.0))
print("  feeder: packs with spill / side entries %.1f of %.1f" % (ft[:, 4].mean(), ft[:, 5].mean()))
ab = np.array([[x.rbs_background_count[j] for j in range(5)] for x in g]).astype(float)
if ab[:, 2].sum() > 0:  # -DDUO_PROFILE_AB
    print("  phases A: %.1f per contig, %.3f us each; B: %.1f, %.3f us each; exact phases %.1f" % (ab[:, 2].mean(), ab[:, 0].sum() / ab[:, 2].sum() / 100.0, ab[:, 3].mean(), ab[:, 1].sum() / ab[:, 3].sum() / 100.0, ab[:, 4].mean()))
for i in np.argsort(-t)[:8]:
    print("  slow contig %4d: %.0f us, packs %d, phases %d, nodes %d, edges %d; solver wait/take/phases/results %s; feeder part1/wait/part2/write %s, packs with side entries %d" % (i, t[i], packs[i], it[i], g[i].n_node, g[i].n_edge, " ".join("%.0f" % x for x in sp[i]), " ".join("%.0f" % (x / 100.0) for x in ft[i, :4]), ft[i, 4]))
for i in np.argsort(t)[n // 2 - 2:n // 2 + 2]:
    print("  median contig %4d: %.0f us, packs %d, phases %d, nodes %d, edges %d; solver wait/take/phases/results %s; feeder part1/wait/part2/write %s, packs with side entries %d" % (i, t[i], packs[i], it[i], g[i].n_node, g[i].n_edge, " ".join("%.0f" % x for x in sp[i]), " ".join("%.0f" % (x / 100.0) for x in ft[i, :4]), ft[i, 4]))
print("  solver wavefront percentiles (us): " + " ".join("p%d %.0f" % (q, np.percentile(t, q)) for q in (10, 25, 50, 75, 90, 95, 98, 99, 100)))
nn = np.array([x.n_node for x in g], float); ne = np.array([x.n_edge for x in g], float)
print("  predictors of the solver's time: corr with nodes %.3f, with edges %.3f, with phases %.3f" % (np.corrcoef(t, nn)[0, 1], np.corrcoef(t, ne)[0, 1], np.corrcoef(t, it)[0, 1]))
for frac in (0.25, 0.33, 0.5):
    k = int(n * frac)
    for nm, key in (("nodes", nn), ("edges", ne)):
        A = set(np.argsort(-key)[:k].tolist())
        rest = [i for i in range(n) if i not in A]
        print("  slowest-first by %s, first %.0f %%: max of the rest %.0f us (of all: %.0f), contigs of the slowest 10 %% inside: %d of %d" % (nm, 100 * frac, t[rest].max(), t.max(), len(A & set(np.argsort(-t)[: n // 10].tolist())), n // 10))
