#!/usr/bin/env python3
"""Two (or more) contexts, each with the WHOLE batch resident, running concurrently from their own host threads: does the
latency-bound shortest-path kernel of one pass hide behind the throughput kernels of the other?   python tools/pipe_probe.py [contexts] [steps]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phanotate_amd as pa
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
anns = [pa.Annotator() for _ in range(nctx)]
for a in anns:
    a.upload(seqs); a.run(); a.run(); a.run()
def single(k):
    t0 = time.perf_counter()
    for _ in range(k): anns[0].run()
    return time.perf_counter() - t0
t1 = single(steps)
def worker(a, k):
    for _ in range(k): a.run()
best = 1e9
for rep in range(3):
    th = [threading.Thread(target=worker, args=(a, steps // nctx)) for a in anns]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    best = min(best, time.perf_counter() - t0)
print("%d contigs: 1 context: %.3f ms/pass; %d contexts concurrently: %.3f ms/pass" % (n, t1 / steps * 1e3, nctx, best / (steps // nctx * nctx) * 1e3))
