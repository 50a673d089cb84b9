"""Time of refine + certificate on top of a run (the library in place).   python tools/cert_time.py [contigs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seqs = [pa.synth_contig(i, 50000) for i in range(n)]
a = pa.Annotator(flags=("no_exact",)); a.upload(seqs); a.run(); a.certified()
for _ in range(3): a.run(); a.certified()
t0 = time.perf_counter()
for _ in range(20): a.run()
t1 = time.perf_counter()
for _ in range(20): a.run(); c = a.certified()
t2 = time.perf_counter()
a.set_profiling(True); a.stage_ms(reset=True)
for _ in range(5): a.run(); a.certified()
st = a.stage_ms()
print("%d contigs: run %.3f ms, run + certificate %.3f ms, on top %.3f ms; certify stage (events) %.3f ms; not certified %d" % (n, (t1 - t0) / 20 * 1e3, (t2 - t1) / 20 * 1e3, (t2 - t1 - (t1 - t0)) / 20 * 1e3, st["certify"][0] / 5, int((c != 1).sum())))
