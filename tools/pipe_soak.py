"""Development: soak of the two-context pipeline with the certificate behind every asynchronous run (phx_run_async): batches of changing size and content through
pipeline.Pipeline, every result equal to what one context delivers for the same batch (Annotator.annotate_flat), certificates all 1.   python tools/pipe_soak.py [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(12)
pool = [pa.synth_contig(50000 + i, int(rng.choice([2000, 12000, 50000, 70000]))) for i in range(1500)]
batches = []
for k in range(nb):
    n = int(rng.choice([1, 3, 40, 300, 700, 1000]))
    idx = rng.choice(len(pool), n, replace=False)
    batches.append([pool[i] for i in idx])
ref = pa.Annotator()
bad = 0
t0 = time.perf_counter()
with pa.Pipeline(depth=2) as pipe:
    for k, got in enumerate(pipe.run(batches)):
        want = ref.annotate_flat(batches[k])
        if any(a.tobytes() != b.tobytes() for a, b in zip(got, want)):
            bad += 1
            print("batch %d (%d contigs) differs" % (k, len(batches[k])), flush=True)
print("pipe_soak: %d batches (%d contigs) through two contexts, %d differ from one context; %.1f s" % (nb, sum(len(b) for b in batches), bad, time.perf_counter() - t0))
sys.exit(1 if bad else 0)
