"""Development: where the host thread of a two-context pipeline spends a batch (pipeline.Pipeline._run_one by hand, with time stamps):
download of the oldest run (wait + certificate + D2H), upload of the next batch (staging + H2D enqueue), run_async (enqueue)."""
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
anns = [pa.Annotator(), pa.Annotator()]
for a in anns:
    for _ in range(3):
        a.annotate_flat_raw(ptrs, lens, seqs)
T = dict(download=0.0, upload=0.0, launch=0.0)
busy = []
t_begin = time.perf_counter()
for k in range(K):
    if len(busy) == 2:
        t0 = time.perf_counter(); busy.pop(0)._download_flat(); T["download"] += time.perf_counter() - t0
    a = anns[k % 2]
    t0 = time.perf_counter(); a.upload_raw(ptrs, lens, seqs)
    t1 = time.perf_counter(); a.run_async()
    t2 = time.perf_counter()
    T["upload"] += t1 - t0; T["launch"] += t2 - t1
    busy.append(a)
while busy:
    t0 = time.perf_counter(); busy.pop(0)._download_flat(); T["download"] += time.perf_counter() - t0
dt = time.perf_counter() - t_begin
print("two contexts, %d contigs: %.3f ms per batch; host thread per batch: %s" % (n, dt / K * 1e3, {k: round(v / K * 1e3, 3) for k, v in T.items()}))
# one context, the same calls
a = anns[0]
T = dict(download=0.0, upload=0.0, launch=0.0, wait=0.0)
t_begin = time.perf_counter()
for k in range(K):
    t0 = time.perf_counter(); a.upload_raw(ptrs, lens, seqs)
    t1 = time.perf_counter(); a.run_async()
    t2 = time.perf_counter(); a.wait()
    t3 = time.perf_counter(); a._download_flat()
    t4 = time.perf_counter()
    T["upload"] += t1 - t0; T["launch"] += t2 - t1; T["wait"] += t3 - t2; T["download"] += t4 - t3
dt = time.perf_counter() - t_begin
print("one context: %.3f ms per batch; %s" % (dt / K * 1e3, {k: round(v / K * 1e3, 3) for k, v in T.items()}))
