#!/usr/bin/env python3
"""Two contexts with the benchmark batch resident: ways of keeping both busy from the host.   python tools/pipe_probe2.py [steps]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PROBE_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
import phanotate_amd as pa
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seqs = [pa.synth_contig(i, 50000) for i in range(1000)]
if os.environ.get("PROBE_THIRD"):
    third = pa.Annotator(); third.annotate_flat(seqs); third.run(); third.set_profiling(True); third.run(); third.set_profiling(False)
anns = [pa.Annotator() for _ in range(2)]
for a in anns:
    a.upload(seqs); a.run(); a.run(); a.run(); a.run()
t0 = time.perf_counter()
for _ in range(steps): anns[0].run()
t1 = (time.perf_counter() - t0) / steps
print("one context: %.3f ms/step" % (t1 * 1e3))
for skew in (0.0, 0.3, 0.5, 0.7):
    t0 = time.perf_counter()
    for k in range(steps):
        anns[k % 2].run_async()
        if k == 0 and skew: time.sleep(skew * t1)
    for a in anns: a.wait()
    print("one host thread, run_async alternating, skew %.1f: %.3f ms/step" % (skew, (time.perf_counter() - t0) / steps * 1e3))
def worker(a, k):
    for _ in range(k): a.run()
th = [threading.Thread(target=worker, args=(a, steps // 2)) for a in anns]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
print("two host threads, run(): %.3f ms/step" % ((time.perf_counter() - t0) / steps * 1e3))
def worker2(a, k):
    for _ in range(k): a.run_async(); a.wait()
th = [threading.Thread(target=worker2, args=(a, steps // 2)) for a in anns]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
print("two host threads, run_async + wait: %.3f ms/step" % ((time.perf_counter() - t0) / steps * 1e3))
