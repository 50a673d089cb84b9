"""Development: how many step-backs k_sssp_wave takes on the benchmark batch (sssp_sweeps - 1 per contig)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ann = pa.Annotator(flags=("no_certify",))
ann.annotate_flat(seqs)
sw = np.array([ann.globals(i).sssp_sweeps for i in range(n)])
it = np.array([ann.globals(i).sssp_iters for i in range(n)])
print("step-backs per contig: mean %.3f, contigs with none %d, max %d; phases per contig: median %d max %d" % ((sw - 1).mean(), int((sw == 1).sum()), int(sw.max() - 1), int(np.median(it)), int(it.max())))
