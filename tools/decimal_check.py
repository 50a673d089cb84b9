#!/usr/bin/env python3
"""libphx's integer edge weights are trunc(fp64(w) * 1000), the reference hands fastpathz trunc(Decimal(w) * 1000) with
28 digits.  For |w| beyond ~9e12 the two integers differ in their low digits, so a near-tie could resolve differently.  This
tool measures it: for every contig it replays the reference's Decimal weights on the integers the GPU delivers
(tests/decimal_replay.py, the code behind the byte-exact --dump), solves the graph with the golden generator's in-order
Bellman-Ford over Graph.iteredges order on those integers (python ints), and compares the node path with the one libphx returned.
Run on the GPU box:   python tools/decimal_check.py [n_contigs] [seed] [max_len]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import phanotate_amd as pa
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
from decimal_replay import decimal_weights, edge_order
from fuzz_gpu import make


def solve(nd, ed, weight):
    order = edge_order(nd, ed)
    E = [(int(ed[k]["src"]), int(ed[k]["dst"]), int(weight[k] * 1000)) for k in order]
    V = len(nd)
    dist, par = [None] * V, [-1] * V
    dist[V - 2] = 0
    for _ in range(V + 1):
        ch = False
        for u, v, w in E:
            du = dist[u]
            if du is None:
                continue
            x = du + w
            if dist[v] is None or x < dist[v]:
                dist[v] = x; par[v] = u; ch = True
        if not ch:
            break
    if dist[V - 1] is None:
        return []
    path = [V - 1]
    while path[-1] != V - 2:
        path.append(par[path[-1]])
    return path[::-1]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    max_len = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
    rng = np.random.RandomState(seed)
    seqs = []
    while len(seqs) < n:
        s = make(rng)
        if len(s) <= max_len and not set(s.lower()) - set("acgt"):  # (the Decimal replay follows the reference, which has no result for other letters)
            seqs.append(s)
    seqs += [pa.synth_contig(1000 + k, 50000).decode() for k in range(max(1, n // 50))]  # and a few of the benchmark's kind
    ann = pa.Annotator()
    t0 = time.time()
    same = diff = skipped = wide = 0
    for b0 in range(0, len(seqs), 50):
        part = seqs[b0:b0 + 50]
        res = ann.annotate(part)
        for i, (status, genes) in enumerate(res):
            if status < 0 or ann.globals(i).n_node <= 2:
                skipped += 1
                continue
            nd, ed, w = decimal_weights(ann, i, part[i])
            fp = ed["w"]
            wide += sum(1 for k in range(len(ed)) if int(w[k] * 1000) != int(np.trunc(float(fp[k]) * 1000.0)))
            p_dec = solve(nd, ed, w)
            p_gpu = [int(x) for x in ann.path(i)[0]]
            if p_dec == p_gpu:
                same += 1
            else:
                diff += 1
                if diff <= 5:
                    print("DIFFERENT PATH contig %d (len %d): %d vs %d nodes" % (b0 + i, len(part[i]), len(p_gpu), len(p_dec)))
    print("decimal check seed %d: %d contigs solved with the reference's Decimal-derived integers: %d identical node paths, %d different; %d edges whose integer differs from trunc(fp64 * 1000); %d contigs without a graph or with an error status; %.0f s"
          % (seed, same + diff, same, diff, wide, skipped, time.time() - t0))


if __name__ == "__main__":
    main()
