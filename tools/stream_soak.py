#!/usr/bin/env python3
"""Soak of the planner / solver hand-shake (batches of up to 800 contigs): many runs of the captured graph per batch size;
every run's records must equal the first run's, and no contig may have been handed to the workgroup kernel because its solver saw no
progress from the planner (phx_globals.sssp_handed_back == 5).   python tools/stream_soak.py [runs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(5)
bad = timeouts = 0
for n in (1, 2, 8, 64, 300, 800):
    seqs = [pa.synth_contig(9000 + i, int(rng.choice([3000, 20000, 50000, 90000]))) for i in range(n)]
    ann = pa.Annotator()
    first = ann.annotate_flat(seqs)
    t0 = time.perf_counter()
    for r in range(runs):
        ann.run()
        got = ann.download_flat(exact=False)
        if any(a.tobytes() != b.tobytes() for a, b in zip(got, first)): bad += 1
        if n <= 8 or r % 25 == 0:
            for i in (range(n) if n <= 64 else rng.choice(n, 32, replace=False)):
                timeouts += 1 if ann.globals(int(i)).sssp_handed_back == 5 else 0
    print("n=%d: %d runs, %.3f ms per run + download, runs that differ from the first so far %d, planner time-outs seen %d" % (n, runs, (time.perf_counter() - t0) / runs * 1e3, bad, timeouts))
    ann.close()
sys.exit(1 if bad or timeouts else 0)
