"""Development: which solver kernel / configuration the contigs of the fuzz generator and the benchmark get (sssp_kernel)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import phanotate_amd as pa
from fuzz_gpu import make
ann = pa.Annotator(flags=("no_certify",))
for seed in range(1, 9):
    rng = np.random.RandomState(seed)
    seqs = [make(rng) for _ in range(60)]
    seqs = [s for s in seqs if not set(s.lower()) - set("acgtnryswkmbvdh")]
    ann.annotate_flat(seqs)
    c = collections.Counter((ann.globals(i).sssp_kernel, ann.globals(i).n_limbs) for i in range(len(seqs)) if ann.globals(i).n_node > 2)
    roomy = [(i, len(seqs[i])) for i in range(len(seqs)) if ann.globals(i).sssp_kernel == 3][:4]
    print("seed", seed, dict(c), "roomy:", roomy)
for L in (20000, 100000, 300000):
    seqs = [pa.synth_contig(100 + k, L) for k in range(40)]
    ann.annotate_flat(seqs)
    print("synthetic", L, dict(collections.Counter((ann.globals(i).sssp_kernel, ann.globals(i).n_limbs) for i in range(len(seqs)))))
