#!/usr/bin/env python3
"""Run with a library built with -DWV_PLAN_TEST_NOPUB -DWV_PLAN_SPINS=3000 (the planner never publishes its progress): every solver that
follows a planner must run into its time-out and hand the contig to the workgroup kernel — same genes as the oracle, no hang."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
from oracle import oracle
seqs = [pa.synth_contig(300 + i, 20000 + 3000 * i) for i in range(6)]
ann = pa.Annotator()
bad = 0; why = {}
for batch in ([seqs[0]], seqs[:3], seqs):
    t0 = time.perf_counter()
    res = ann.annotate(batch)
    dt = time.perf_counter() - t0
    for i, (st, g) in enumerate(res):
        o = oracle.run(batch[i])
        ok = st == o["status"] == 0 and np.array_equal(g["left"], o["gene_left"]) and np.array_equal(g["right"], o["gene_right"])
        bad += 0 if ok else 1
        gl = ann.globals(i); why[(gl.sssp_kernel, gl.sssp_handed_back)] = why.get((gl.sssp_kernel, gl.sssp_handed_back), 0) + 1
    print("batch of %d: %.1f ms" % (len(batch), dt * 1e3))
print("nopub_check: %d contigs differ from the oracle; (solver kernel, handed back) counts %s  [expected: all (1, 5)]" % (bad, why))
sys.exit(1 if bad else 0)
