"""How many contigs of the benchmark series (seeds a..b, 50 kb) does the device certify, what is left for the host, and what does
the certificate cost?   python tools/cert_count.py [first_seed] [n] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import phanotate_amd as pa
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
raw = pa.Annotator(flags=("no_exact",))
ann = pa.Annotator()
tot = {"contigs": 0, "device": 0, "host": 0, "neither": 0}
for b0 in range(first, first + n, batch):
    seqs = [pa.synth_contig(s, 50000) for s in range(b0, min(b0 + batch, first + n))]
    raw.upload(seqs); raw.run()
    t0 = time.time(); c0 = raw.certified(); t1 = time.time()
    raw.run(); t2 = time.time(); c0 = raw.certified(); t3 = time.time()
    r0 = raw.download_flat()
    ann.upload(seqs); ann.run()
    t4 = time.time(); r1 = ann.download_flat(); t5 = time.time()
    c1 = ann.certified()
    un = [b0 + int(i) for i in np.nonzero(c0 == 0)[0]]
    changed = [b0 + i for i in ann.resolved if r0[2][r0[1][i]:r0[1][i + 1]].tobytes() != r1[2][r1[1][i]:r1[1][i + 1]].tobytes()]
    tot["contigs"] += len(seqs); tot["device"] += int((c1 == 1).sum()); tot["host"] += int((c1 == 2).sum()); tot["neither"] += int((c1 == 0).sum())
    flagged = edges = 0
    for i in range(0, len(seqs), 97):
        ed = raw.edges(i); flagged += int(ed["inexact"].sum()); edges += len(ed)
    print("seeds %d..%d: not certified on the device %s; solved again on the host %s, of which with other genes %s; refine + certificate %.2f ms (first call %.2f), download with the guarantee %.1f ms; still flagged %d of %d sampled edges"
          % (b0, b0 + len(seqs) - 1, un, [b0 + i for i in ann.resolved], changed, (t3 - t2) * 1e3, (t1 - t0) * 1e3, (t5 - t4) * 1e3, flagged, edges), flush=True)
print(tot)
