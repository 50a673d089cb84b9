"""Development: host ASCII -> host gene lists for a stream of batches through pipeline.Pipeline, by depth (contexts taking turns) and by the form the
batches are handed over in (python lists of bytes, or the C-ABI's own arguments: addresses and lengths)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ptrs = np.array([C.cast(C.c_char_p(s), C.c_void_p).value for s in seqs], np.uint64)
lens = np.array([len(s) for s in seqs], np.int64)
for depth in (2, 3, 4):
    pipe = pa.Pipeline(depth=depth)
    for form, batch in (("lists", seqs), ("raw", (ptrs, lens, seqs))):
        for _ in pipe.run([batch] * (depth + 2)):
            pass
        t0 = time.perf_counter()
        for _ in pipe.run([batch] * K):
            pass
        print("depth %d, %s: %.3f ms per batch" % (depth, form, (time.perf_counter() - t0) / K * 1e3), flush=True)
    pipe.close()
