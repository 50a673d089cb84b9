#!/usr/bin/env python3
"""One-off validation: every contig of the benchmark batch against the oracle (gene coordinates, strands, path length).
Run on the GPU box:  python tools/validate_batch.py [contigs] [length]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ProcessPoolExecutor
import numpy as np

def orc(args):
    seed, L = args
    import phanotate_amd as pa
    from oracle import oracle
    o = oracle.run(pa.synth_contig(seed, L))
    return seed, int(o["status"]), np.asarray(o["gene_left"]).tolist(), np.asarray(o["gene_right"]).tolist(), np.asarray(o["gene_strand"]).tolist()

def main():
    C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    L_ = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    import phanotate_amd as pa
    seqs = [pa.synth_contig(i, L_) for i in range(C_)]
    ann = pa.Annotator()
    t0 = time.time()
    res = ann.annotate(seqs)
    kern = [ann.globals(i).sssp_kernel for i in range(C_)]
    back = sum(ann.globals(i).sssp_handed_back for i in range(C_))
    print("gpu done in %.2f s; solver kernels used: %s; handed back %d" % (time.time() - t0, dict(zip(*np.unique(kern, return_counts=True))), back))
    bad = 0
    with ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        for seed, st, gl, gr, gs in ex.map(orc, [(i, L_) for i in range(C_)], chunksize=8):
            status, genes = res[seed]
            ok = status == st and [int(x) for x in genes["left"]] == gl and [int(x) for x in genes["right"]] == gr and [int(x) for x in genes["strand"]] == gs
            if not ok:
                bad += 1
                if bad <= 5: print("MISMATCH contig", seed, status, st, len(genes), len(gl))
    print("contigs %d, mismatches %d, oracle+compare %.1f s" % (C_, bad, time.time() - t0))
    return 1 if bad else 0
if __name__ == "__main__":
    sys.exit(main())
