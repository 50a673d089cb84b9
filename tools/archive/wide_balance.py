"""Development (library built with EXTRA=-DWV_PROFILE): k_sssp_wave's per-contig time for contigs with a very long ORF."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
rng = np.random.RandomState(42)
sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
def wide(seed, ncod):
    body = "".join(rng.choice(sense, ncod))
    return (pa.synth_contig(900 + seed, 20000).decode() + "atg" + body + "taa" + pa.synth_contig(1900 + seed, 20000).decode()).encode()
ncod = int(sys.argv[1]) if len(sys.argv) > 1 else 2200
seqs = [pa.synth_contig(i, 50000) for i in range(100)] + [wide(k, ncod) for k in range(20)]
ann = pa.Annotator(flags=("no_certify",))
ann.annotate_flat(seqs); ann.run(); ann._download_flat()
for i in list(range(3)) + list(range(100, 120)):
    x = ann.globals(i)
    print(i, "limbs", x.n_limbs, "kernel", x.sssp_kernel, "us %.1f" % (x.rbs_background_count[6] / 100.0), "windows", x.rbs_background_count[7], "phases", x.sssp_iters,
          "64-bit/redo/exact/rebase", [x.rbs_training_count[j] for j in range(4)], "gather us %.1f phases us %.1f" % (x.gc_max_count[2] / 100.0, x.rbs_background_count[4] / 100.0), "nodes", x.n_node, "edges", x.n_edge)
