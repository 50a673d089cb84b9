#!/usr/bin/env python3
"""20000 short contigs (300-3000 bp) in one batch: libphx vs the oracle on a sample of them (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ProcessPoolExecutor

def orc(seq):
    from oracle import oracle
    o = oracle.run(seq)
    return int(o["status"]), np.asarray(o["gene_left"]).tolist(), np.asarray(o["gene_right"]).tolist()

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.RandomState(7)
    seqs = []
    for i in range(n):
        L = int(rng.randint(300, 3000))
        gc = rng.uniform(0.3, 0.7)
        seqs.append("".join(rng.choice(list("acgt"), L, p=[(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])))
    import phanotate_amd as pa
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    pick = rng.choice(n, 400, replace=False)
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        want = list(ex.map(orc, [seqs[i] for i in pick], chunksize=8))
    bad = 0
    for i, (st, gl, gr) in zip(pick, want):
        status, genes = res[i]
        ok = (status == st) if st < 0 else (status >= 0 and [int(x) for x in genes["left"]] == gl and [int(x) for x in genes["right"]] == gr)
        bad += 0 if ok else 1
    kern = {}
    for i in pick[:50]:
        g = ann.globals(int(i)); kern[g.sssp_kernel] = kern.get(g.sssp_kernel, 0) + 1
    print("many_small: %d contigs in one batch, %d of 400 sampled differ from the oracle (ties included); kernels of 50: %s" % (n, bad, kern))
    return 1 if bad > 8 else 0

if __name__ == "__main__":
    sys.exit(main())
