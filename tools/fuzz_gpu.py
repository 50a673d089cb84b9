#!/usr/bin/env python3
"""Fuzz libphx against the oracle on sequences unlike the benchmark's: GC from 20 % to 80 %, start-codon-rich and
stop-poor stretches, tandem repeats, homopolymers, N runs, IUPAC codes.  Run on the GPU box:
    python tools/fuzz_gpu.py [n_contigs] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ProcessPoolExecutor
import numpy as np

def make(rng):
    L = int(rng.choice([300, 2000, 6000, 12000, 25000, 40000]))
    gc = rng.uniform(0.2, 0.8)
    p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    s = rng.choice(list("acgt"), L, p=p)
    kind = rng.randint(9)
    def put(at, text):
        text = text[: max(0, L - at)]
        s[at:at + len(text)] = list(text)
    if kind == 1:  # start-codon-rich stretch without stops
        n = int(rng.randint(200, 1500))
        put(int(rng.randint(0, max(1, L - 3 * n))), "".join(rng.choice(["atg", "gtg", "ttg", "gcc", "gac", "ctc"], n)))
    elif kind == 2:  # tandem repeats
        unit = "".join(rng.choice(list("acgt"), int(rng.randint(1, 12))))
        put(int(rng.randint(0, L)), unit * int(rng.randint(20, 400)))
    elif kind == 3:  # N run and IUPAC codes
        put(int(rng.randint(0, L)), "n" * int(rng.randint(1, 700)))
        for _ in range(20): s[int(rng.randint(L))] = rng.choice(list("ryswkmbvdh"))
    elif kind == 4:  # stop-codon-rich
        n = int(rng.randint(100, 2000))
        put(int(rng.randint(0, L)), "".join(rng.choice(["taa", "tag", "tga", "tta", "cta", "tca"], n)))
    elif kind == 5:  # many short ORFs back to back: dense nodes
        n = int(rng.randint(20, 300))
        put(int(rng.randint(0, L)), "".join("atg" + "".join(rng.choice(["gcc", "gtg", "aaa", "ctg"], 31)) + "taa" for _ in range(n)))
    elif kind == 6:  # GC-rich long open frames
        n = int(rng.randint(500, 3000))
        put(int(rng.randint(0, max(1, L - 3 * n))), "atg" + "".join(rng.choice(["gcc", "ggc", "gtg", "cgc", "ccg", "gcg"], n)) + "tga")
    elif kind == 7:  # starts of both strands mixed into stop-free frames: many close AND open nodes within 500 bp
        for _ in range(int(rng.randint(1, 4))):
            n = int(rng.randint(150, 900))
            q = rng.uniform(0.03, 0.2)
            cod = ["atg", "gtg", "ttg", "cat", "cac", "caa", "gcc", "gac", "ctc", "aaa", "ggc", "acg"]
            pr = [q / 3] * 6 + [(1 - 2 * q) / 6] * 6
            put(3 * int(rng.randint(0, max(1, (L - 3 * n) // 3))), "".join(rng.choice(cod, n, p=pr)))
    seq = "".join(s)
    if rng.rand() < 0.2: seq = seq.upper()
    return seq

def orc(seq):
    from oracle import oracle
    o = oracle.run(seq)
    if o["status"] == -7:  # the oracle's 256-bit integers overflow: not compared here (tests/ has python-int solves for such cases)
        return 0, None
    if o["status"] < 0:  # no genes: only the status is compared (-9: the relaxation never settles, a cycle of negative length)
        return int(o["status"]), ([], [], [], None)
    return int(o["status"]), (np.asarray(o["gene_left"]).tolist(), np.asarray(o["gene_right"]).tolist(), np.asarray(o["gene_strand"]).tolist(), int(o["path_dist"]) if len(o["path"]) else None)

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    seqs = [make(rng) for _ in range(n)]
    import phanotate_amd as pa
    ann = pa.Annotator()
    bad = 0; ties = 0; fixed = 0; kern = {}; back = 0; skipped = 0; why = {}; n_host = 0; exact_wins = 0
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        want = list(ex.map(orc, seqs, chunksize=4))
    for b0 in range(0, n, 100):
        part = seqs[b0:b0 + 100]
        res = ann.annotate(part)
        cert = ann.certified()  # 1 proven on the device, 2 solved again on the host in the reference's own arithmetic (inside the library)
        n_host += int((cert == 2).sum())
        for i, (status, genes) in enumerate(res):
            g = ann.globals(i)
            if g.n_node > 2 and status >= 0: kern[(g.n_limbs, g.sssp_kernel)] = kern.get((g.n_limbs, g.sssp_kernel), 0) + 1
            back += 1 if g.sssp_handed_back else 0
            why[g.sssp_handed_back] = why.get(g.sssp_handed_back, 0) + (1 if g.sssp_handed_back else 0)
            st, exp = want[b0 + i]
            if exp is None: skipped += 1; continue
            ok = (status == st) if st < 0 else (status >= 0 and [int(x) for x in genes["left"]] == exp[0] and [int(x) for x in genes["right"]] == exp[1] and [int(x) for x in genes["strand"]] == exp[2])
            if st >= 0 and status == 0 and g.n_node > 2: ties += 1 if g.tie else 0; fixed += 1 if g.tie == 2 else 0
            if not ok and cert[i] == 2:
                # the fp64-level oracle and the reference's integers disagree: decimal.Decimal itself decides (dump.python_resolve)
                import os as _os, sys as _sys
                _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
                import decimal_replay as dump
                py = dump.python_resolve(ann, i, part[i])
                if [(int(x["left"]), int(x["right"]), int(x["strand"])) for x in genes] == [t[:3] for t in py]:
                    exact_wins += 1
                    continue
            if not ok:
                bad += 1
                if bad <= 5: print("MISMATCH contig %d (len %d): status %d vs %d, %d vs %d genes" % (b0 + i, len(part[i]), status, st, len(genes), len(exp[0])))
    print("fuzz seed %d: %d contigs, %d mismatches (a differently resolved tie is a mismatch), %d contigs with equal-length alternatives (%d paths replaced by k_inorder), %d beyond the oracle's integers; solved again on the host %d (of which the Decimal integers gave another path than the fp64 oracle: %d); (limbs, kernel) counts %s; handed back %d (by reason %s)" % (seed, n, bad, ties, fixed, skipped, n_host, exact_wins, kern, back, {k: v for k, v in why.items() if k}))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
