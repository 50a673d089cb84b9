"""Where the host-to-host time of config 5's 10 000 contigs on one GPU goes: upload / run / certificate / download, ms.
   python tools/h2h10k.py [contigs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
ptrs = np.array([C.cast(C.c_char_p(s), C.c_void_p).value for s in seqs], np.uint64)
lens = np.array([len(s) for s in seqs], np.int64)
a = pa.Annotator()
a.annotate_flat_raw(ptrs, lens, seqs)
for rep in range(3):
    t0 = time.perf_counter(); a.upload_raw(ptrs, lens, seqs)
    t1 = time.perf_counter(); a.run()
    t2 = time.perf_counter(); c = a.certified()
    t3 = time.perf_counter(); r = a.download_flat()
    t4 = time.perf_counter()
    print("upload %.2f  run %.2f  certificate %.2f  download %.2f  total %.2f ms; not certified on the device %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, int((c != 1).sum())), flush=True)
a.set_profiling(True); a.stage_ms(reset=True)
a.run(); a.certified()
print({k: round(v[0], 3) for k, v in a.stage_ms().items() if v[1]})
t0 = time.perf_counter()
for _ in range(3):
    a.annotate_flat_raw(ptrs, lens, seqs)
print("annotate_flat_raw x3: %.2f ms each" % ((time.perf_counter() - t0) / 3 * 1e3))
