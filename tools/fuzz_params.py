#!/usr/bin/env python3
"""tools/fuzz_gpu.py's contigs under OTHER flags than the defaults: start codon sets and weights (-s), stop codon sets (-e), minimum
ORF lengths (-l) drawn per batch of 60 contigs; libphx (with the certificate and the host re-solve on) against the oracle run with the
same flags.  A contig on which the two disagree while the library says it solved it again on the host is decided by python's decimal
(dump.python_resolve), as in fuzz_gpu.py.  Run on the GPU box:
    python tools/fuzz_params.py [n_batches] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ProcessPoolExecutor
import numpy as np
from fuzz_gpu import make

def draw_flags(rng):
    codons = ["atg", "gtg", "ttg", "ctg", "att", "ata"]
    k = int(rng.choice([1, 2, 3, 3, 4, 6]))
    cs = ["atg"] + list(rng.choice(codons[1:], k - 1, replace=False))
    ws = []
    for i in range(k):
        nd = int(rng.choice([1, 2, 3, 6]))
        w = round(float(rng.uniform(0.01, 1.0)), nd)
        ws.append(("%." + str(nd) + "f") % max(w, 10.0 ** -nd))
    if rng.rand() < 0.3: ws[0] = "1"
    stops = str(rng.choice(["tag,tga,taa", "tag,taa", "taa,tga", "taa", "tga,tag,taa"]))
    return dict(start_codons=",".join(c + ":" + w for c, w in zip(cs, ws)), stop_codons=stops, minlen=int(rng.choice([6, 30, 60, 90, 90, 150, 300])))

def orc(arg):
    seq, kw = arg
    from oracle import oracle
    o = oracle.run(seq, oracle.make_params(**kw))
    if o["status"] < 0:
        return int(o["status"]), ([], [], [])
    return 0, (np.asarray(o["gene_left"]).tolist(), np.asarray(o["gene_right"]).tolist(), np.asarray(o["gene_strand"]).tolist())

def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    import phanotate_amd as pa
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
    import decimal_replay as dump
    bad = n = n_host = exact_wins = 0
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        for b in range(nb):
            kw = draw_flags(rng)
            seqs = [make(rng) for _ in range(60)]
            want = list(ex.map(orc, [(s, kw) for s in seqs], chunksize=4))
            ann = pa.Annotator(pa.make_params(**kw))
            res = ann.annotate(seqs)
            cert = ann.certified()
            n_host += int((cert == 2).sum())
            for i, (status, genes) in enumerate(res):
                st, exp = want[i]
                n += 1
                ok = (status == st) if st < 0 else (status >= 0 and [int(x) for x in genes["left"]] == exp[0] and [int(x) for x in genes["right"]] == exp[1] and [int(x) for x in genes["strand"]] == exp[2])
                if not ok and cert[i] == 2:
                    py = dump.python_resolve(ann, i, seqs[i], kw["start_codons"])
                    if [(int(x["left"]), int(x["right"]), int(x["strand"])) for x in genes] == [t[:3] for t in py]:
                        exact_wins += 1
                        continue
                if not ok:
                    bad += 1
                    if bad <= 5: print("MISMATCH batch %d contig %d (len %d) flags %s: status %d vs %d, %d vs %d genes, cert %d" % (b, i, len(seqs[i]), kw, status, st, len(genes), len(exp[0]), cert[i]))
            ann.close()
    print("fuzz_params seed %d: %d contigs in %d batches of their own flags, %d mismatches; solved again on the host %d (of which decimal.Decimal sides with the library against the fp64 oracle: %d)" % (seed, n, nb, bad, n_host, exact_wins))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
