"""Development: what the host re-solve of an uncertified contig costs (seed 3378 of the benchmark generator is one)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, phanotate_amd as pa
seqs = [pa.synth_contig(i, 50000) for i in (3378, 1, 2)]
ann = pa.Annotator()
t0 = time.perf_counter(); r = ann.annotate_flat(seqs); t1 = time.perf_counter()
print("certified", ann.certified().tolist(), "resolved", ann.resolved, "annotate_flat %.3f s" % (t1 - t0))
t0 = time.perf_counter(); r2 = ann.annotate_flat(seqs); print("again %.3f s" % (time.perf_counter() - t0))
# the full Decimal replay gives the same genes
raw = ann._download_flat()
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
import decimal_replay as dump
full = ann.resolve_uncertified(ann._seq_of, raw)  # flagged-only
import phanotate_amd.api as api
print("genes equal device:", all(a.tobytes() == b.tobytes() for a, b in zip(raw, r[:3])))
import cProfile, pstats
cProfile.run("ann.resolve_uncertified(ann._seq_of, raw)", "/tmp/prof.out")
pstats.Stats("/tmp/prof.out").sort_stats("cumulative").print_stats(14)
