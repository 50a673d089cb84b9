#!/usr/bin/env python3
"""Resident small batches with and without the fused front end (k_front, phx_front.inc): ms per run of the captured graph.
   python tools/small_fused.py        (on the GPU box)"""
import gzip, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phanotate_amd as pa

def fa(case):
    with gzip.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", case + ".fasta.gz"), "rt") as f:
        return "".join(f.read().split("\n")[1:]).encode()

cases = [("lambda", [fa("NC_001416.1")]), ("t4", [fa("NC_000866.1")]), ("phiX174", [fa("phiX174")])]
for n in (2, 8, 32, 64):
    cases.append(("%d x 50 kb" % n, [pa.synth_contig(i, 50000) for i in range(n)]))
cases.append(("64 x 3 kb", [pa.synth_contig(i, 3000) for i in range(64)]))
for name, seqs in cases:
    row = []
    for flags in ((), ("no_fuse",)):
        a = pa.Annotator(flags=flags)
        a.annotate_flat(seqs)
        for _ in range(60):
            a.run()
        t0 = time.perf_counter()
        for _ in range(200):
            a.run()
        row.append((time.perf_counter() - t0) / 200 * 1e3)
        fr = a.front_runs()
        a.close()
    print("%-12s fused %.4f ms   staged %.4f ms   (front runs of the staged context: %d)" % (name, row[0], row[1], fr))
