#!/usr/bin/env python3
"""CPU census (dev probe, uses the oracle as the checker): on how many contigs does the best path admit an
equal-length alternative?  A contig is "ambiguous" when the backward closure of the target over tight edges
(d[u] + w == d[v]) contains a node with two or more tight in-edges; only then can the relaxation order of the
solver (fastpathz / make_golden.bellman_ford: in-place, iteredges order) matter.
    python tools/tie_census.py bench 0 200      # benchmark contigs, seeds 0..199
    python tools/tie_census.py fuzz 300 101     # tools/fuzz_gpu.py contigs"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concurrent.futures import ProcessPoolExecutor

import numpy as np


def census(seq):
    from oracle import oracle
    o = oracle.run(seq)
    if o["status"] != 0 or len(o.get("path", [])) == 0:
        return None
    src = o["edge_src"].tolist()
    dst = o["edge_dst"].tolist()
    w = [oracle.limbs_to_int(x) for x in o["edge_wint_limbs"]]
    V = len(o["node_pos"])
    s, t = V - 2, V - 1
    dist = [None] * V
    dist[s] = 0
    par = [-1] * V
    for _ in range(V):
        ch = False
        for e in range(len(src)):
            du = dist[src[e]]
            if du is None:
                continue
            nd = du + w[e]
            if dist[dst[e]] is None or nd < dist[dst[e]]:
                dist[dst[e]] = nd
                par[dst[e]] = e
                ch = True
        if not ch:
            break
    tight_in = [[] for _ in range(V)]
    for e in range(len(src)):
        if dist[src[e]] is not None and dist[dst[e]] is not None and dist[src[e]] + w[e] == dist[dst[e]]:
            tight_in[dst[e]].append(e)
    seen = {t}
    stack = [t]
    merges = 0
    while stack:
        v = stack.pop()
        if len(tight_in[v]) > 1:
            merges += 1
        for e in tight_in[v]:
            if src[e] not in seen:
                seen.add(src[e])
                stack.append(src[e])
    return merges, len(seen), len(o["path"])


def main():
    kind = sys.argv[1]
    if kind == "bench":
        import phanotate_amd as pa
        a, b = int(sys.argv[2]), int(sys.argv[3])
        seqs = [pa.synth_contig(s, 50000) for s in range(a, b)]
    else:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import fuzz_gpu
        n, seed = int(sys.argv[2]), int(sys.argv[3])
        rng = np.random.RandomState(seed)
        seqs = [fuzz_gpu.make(rng) for _ in range(n)]
    with ProcessPoolExecutor(max_workers=os.cpu_count()) as ex:
        res = list(ex.map(census, seqs, chunksize=2))
    res = [r for r in res if r is not None]
    amb = [r for r in res if r[0] > 0]
    print("%s: %d contigs with a path, %d ambiguous (merge nodes among the target's tight ancestors: %s); closure size / path length of the ambiguous ones: %s"
          % (kind, len(res), len(amb), [r[0] for r in amb][:20], [(r[1], r[2]) for r in amb][:20]))


if __name__ == "__main__":
    main()
