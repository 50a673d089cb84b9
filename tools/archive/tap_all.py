"""Development: edge tap check (recomputed fp64 weights against the solver's integers) and certificate for every contig of a batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
lo, n, step = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seeds = [lo + k * step for k in range(n)]
seqs = [pa.synth_contig(i, 50000) for i in seeds]
ann = pa.Annotator()
ann.upload(seqs); ann.run()
cert = ann.certified()
print("batch of", n, "uncertified:", [seeds[i] for i in np.nonzero(cert == 0)[0]])
bad = 0
for i in range(0, n, max(1, n // 200)):
    try:
        ann.edges(i)
    except Exception as e:
        bad += 1
        if bad <= 3: print("  contig", i, e)
print("tap failures:", bad)
