"""Segments (phx_sssp_seg.inc) on random contigs: every result must equal the one-sweep solver's byte for byte; how often a run is given up
(the frames had not run together within the margin) is what this counts.  python tools/seg_fuzz.py [n_lone] [n_batches] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import phanotate_amd as pa

n_lone = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.RandomState(seed)
a = pa.Annotator()
w = pa.Annotator(flags=("no_seg",))
bad = 0
fallbacks = 0


def one(seqs):
    global bad
    before = a.seg_runs()
    fb0 = a.seg_fallbacks()
    x = a.annotate_flat(seqs)
    y = w.annotate_flat(seqs)
    after = a.seg_runs()
    global fallbacks
    fallbacks += a.seg_fallbacks() - fb0
    if any(p.tobytes() != q.tobytes() for p, q in zip(x, y)):
        bad += 1
        print("MISMATCH", len(seqs), [len(s) for s in seqs][:4], flush=True)
    return after < 0 or after == before  # given up (or segments not used at all)


for margin in ("6000",):
    giveups = 0
    for i in range(n_lone):
        L = int(rng.choice([8000, 20000, 50000, 100000, 200000]) * rng.uniform(0.7, 1.3))
        if one([pa.synth_contig(100000 * seed + i, L)]):
            giveups += 1
            if os.environ.get("PHX_DEBUG_SEG"):
                print("  given up: synth_contig(%d, %d)" % (100000 * seed + i, L), flush=True)
    print("lone contigs: %d runs, %d contigs solved by one sweep behind their segments, %d runs without segments (contigs too short for two, or a run repeated), %d mismatches" % (n_lone, fallbacks, giveups, bad), flush=True)
    fallbacks = 0
    giveups = 0
    tot = 0
    for i in range(n_batch):
        n = int(rng.randint(2, 33))
        seqs = [pa.synth_contig(100000 * seed + 50000 + 40 * i + k, int(rng.uniform(5000, 80000))) for k in range(n)]
        tot += n
        if one(seqs):
            giveups += 1
    print("batches of 2-32: %d runs (%d contigs), %d contigs solved by one sweep behind their segments, %d runs without segments (contigs too short for two, or a run repeated), %d mismatches" % (n_batch, tot, fallbacks, giveups, bad), flush=True)
sys.exit(1 if bad else 0)
