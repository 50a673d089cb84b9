#!/usr/bin/env python3
"""Development: the shortest-path stage when k_sssp_lds<2> is the solver (flag solver_no_wave) — the benchmark batch and a batch of
20 000 short contigs.  Used to compare register budgets of that kernel (-DSW_WPS=5 / 6)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa

def stage(seqs, flags, steps=10):
    ann = pa.Annotator(flags=flags)
    ann.upload(seqs); ann.run(); ann.run()
    ann.set_profiling(True)
    for _ in range(steps): ann.run()
    ms = ann.stage_ms()
    st = ann._download_flat()[0]
    ann.close()
    return ms, int((st < 0).sum())

big = [pa.synth_contig(i, 50000) for i in range(1000)]
rng = np.random.RandomState(7)
small = ["".join(rng.choice(list("acgt"), int(rng.randint(300, 3000)))) for _ in range(20000)]
for name, seqs in (("1000 x 50 kb", big), ("20000 x 0.3-3 kb", small)):
    for flags in (("solver_no_wave",), ()):
        ms, bad = stage(seqs, flags)
        per = {k: v[0] / max(v[1], 1) for k, v in ms.items()}
        print("%-18s flags %-20s sssp %.4f ms  (all stages %.4f, failing contigs %d)" % (name, flags, per["sssp"], sum(v for k, v in per.items() if k != "wave_plan"), bad))
