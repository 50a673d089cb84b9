#!/usr/bin/env python3
"""Contexts created with PHX_CREATE_POISON (every device buffer filled with a garbage pattern when it is allocated or grown): results must
equal those of a plain context — for lone contigs, small batches (the solver follows its planner) and the benchmark's kind of batch.
    python tools/poison_fuzz.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
from fuzz_gpu import make
rng = np.random.RandomState(77)
bad = 0
plain = pa.Annotator()
for n in (1, 1, 1, 3, 17, 100, 400):
    seqs = [make(rng) for _ in range(n)] if n < 400 else [pa.synth_contig(5000 + i, 50000) for i in range(n)]
    want = plain.annotate_flat(seqs)
    for rep in range(2):
        p = pa.Annotator(flags=("poison",))
        got = p.annotate_flat(seqs)
        p.run(); again = p.download_flat()
        for x in (got, again):
            bad += 0 if all(a.tobytes() == b.tobytes() for a, b in zip(x, want)) else 1
        p.close()
print("poison_fuzz: %d result sets of poisoned contexts differ from the plain context's" % bad)
sys.exit(1 if bad else 0)
