#!/usr/bin/env python3
"""What a few wide-integer contigs cost a batch: the benchmark's 1000 contigs alone, then with 20 contigs whose path sums need
256 bits (they are solved by k_sssp_lds<4>, one workgroup each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, phanotate_amd as pa
rng = np.random.RandomState(42)
sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
def wide(seed, ncod):
    body = "".join(rng.choice(sense, ncod))
    return (pa.synth_contig(900 + seed, 20000).decode() + "atg" + body + "taa" + pa.synth_contig(1900 + seed, 20000).decode()).encode()
base = [pa.synth_contig(i, 50000) for i in range(1000)]
for ncod in (0, 2200, 3000):
    seqs = base + ([wide(k, ncod) for k in range(20)] if ncod else [])
    a = pa.Annotator(); a.annotate(seqs); a.run(); a.set_profiling(True); a.stage_ms(reset=True)
    for _ in range(5): a.run()
    st = a.stage_ms()
    limbs = {}
    for i in range(1000, len(seqs)):
        g = a.globals(i); limbs[(g.n_limbs, g.sssp_kernel)] = limbs.get((g.n_limbs, g.sssp_kernel), 0) + 1
    print("extra contigs with a %d-codon ORF: %s; sssp stage %.3f ms, step %.3f ms" % (ncod, limbs, st["sssp"][0] / 5, sum(v[0] for v in st.values()) / 5))
    a.close()
