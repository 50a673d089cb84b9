"""Development: over a range of benchmark seeds, the contigs k_certify does not certify and the contigs whose edge tap fails its check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
lo, hi, step = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1
BATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 500
ann = pa.Annotator()
for b0 in range(lo, hi, BATCH * step):
    seeds = list(range(b0, min(hi, b0 + BATCH * step), step))
    seqs = [pa.synth_contig(i, 50000) for i in seeds]
    ann.upload(seqs); ann.run()
    cert = ann.certified()
    for i in np.nonzero(cert == 0)[0]:
        print("seed", seeds[i], "not certified")
        try:
            ann.edges(int(i))
        except Exception as e:
            print("  ", e)
print("scanned", lo, hi, step)
