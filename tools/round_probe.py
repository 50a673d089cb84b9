"""Development: the rounds of tests/test_handshake_gpu.py::test_two_contexts_in_flight_follow_their_planners, one line per slow round."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
def batch(n, seed0):
    rng = np.random.RandomState(seed0)
    return [pa.synth_contig(seed0 + i, int(rng.choice([3000, 20000, 50000, 90000]))) for i in range(n)]
seqs = [batch(n, 9000), batch(n, 29000)]
pipe = pa.Pipeline(device=0, depth=2)
anns = pipe.anns
for a, s in zip(anns, seqs):
    a.annotate_flat(s)
ts = []
for r in range(60):
    t0 = time.perf_counter()
    for a in anns:
        a.run_async()
    t1 = time.perf_counter()
    w = []
    for a in anns:
        a.wait(); w.append(time.perf_counter())
        a.download_flat(exact=False); w.append(time.perf_counter())
    ts.append((time.perf_counter() - t0) * 1e3)
    if ts[-1] > 6.0 or r < 3:
        print("round %d: %.2f ms; launch %.2f, then %s" % (r, ts[-1], (t1 - t0) * 1e3, " ".join("%.2f" % ((x - t1) * 1e3) for x in w)))
print("rounds: median %.2f max %.2f; resolved %s" % (np.median(ts), max(ts), [a.plan_timeouts() for a in anns]))
