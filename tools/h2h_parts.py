"""Development: where a host-to-host step goes: upload (staging + H2D), run, download, certificate; by staging thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
seqs = [pa.synth_contig(i, 50000) for i in range(1000)]
ann = pa.Annotator()
ann.annotate_flat(seqs)
for _ in range(3): ann.annotate_flat(seqs)
T = dict(upload=0.0, run=0.0, download=0.0, cert=0.0)
K = 20
for _ in range(K):
    t0 = time.perf_counter(); ann.upload(seqs)
    t1 = time.perf_counter(); ann.run()
    t2 = time.perf_counter(); r = ann._download_flat()
    t3 = time.perf_counter(); c = ann.certified()
    t4 = time.perf_counter()
    T["upload"] += t1 - t0; T["run"] += t2 - t1; T["download"] += t3 - t2; T["cert"] += t4 - t3
print({k: round(v / K * 1e3, 3) for k, v in T.items()}, "ms; python-side list/ctypes marshalling is inside upload")
# the raw-pointer upload (what the CLI uses): no per-contig bytes -> char* conversion
import ctypes as C
ptrs = np.array([C.cast(C.c_char_p(s), C.c_void_p).value for s in seqs], np.uint64)
lens = np.array([len(s) for s in seqs], np.int64)
t0 = time.perf_counter()
for _ in range(K): ann.upload_raw(ptrs, lens, seqs)
print("upload_raw %.3f ms" % ((time.perf_counter() - t0) / K * 1e3))
