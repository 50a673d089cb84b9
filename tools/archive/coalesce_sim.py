"""CPU experiment behind phx_sssp_seg.inc (round 5): a Bellman-Ford sweep started at position p0 — the nodes of its first 500 bp at distance
0, everything to the left and the source node absent — against the true distances: from which position on is the difference ONE constant?
Uses the oracle (test infrastructure) for the graph.  python tools/archive/coalesce_sim.py tests/golden/NC_001416.1.fasta.gz 1000"""
import sys, gzip, math, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import oracle as orc
from tests.conftest import read_fasta_gz

def graph(seq):
    o = orc.run(seq, stages=2)
    pos = o["node_pos"].astype(np.int64)
    V = len(pos)
    src, dst, w = o["edge_src"], o["edge_dst"], o["edge_weight"]
    W = [int(math.trunc(float(x) * 1000.0)) for x in w]
    return o, V, pos, src.astype(np.int64), dst.astype(np.int64), W

def bf(V, order_edges, seeds, allowed):
    # order_edges: list of (u, v, w) sorted by position of v; allowed: bool array of nodes in play
    INF = None
    d = [None] * V
    for s in seeds: d[s] = 0
    rounds = 0
    while True:
        ch = False
        rounds += 1
        for (u, v, w) in order_edges:
            du = d[u]
            if du is None: continue
            nd = du + w
            if d[v] is None or nd < d[v]:
                d[v] = nd; ch = True
        if not ch or rounds > 200: break
    return d, rounds

def main(path, step, name):
    recs = read_fasta_gz(path)
    seq = recs[1] if isinstance(recs[1], (str, bytes)) else recs[1][0]
    o, V, pos, src, dst, W = graph(seq)
    L = o["L"]
    # identify source / target: the oracle's last two nodes? find node with no in-edges and many out-edges
    indeg = np.bincount(dst, minlength=V); outdeg = np.bincount(src, minlength=V)
    S = [i for i in range(V) if indeg[i] == 0 and outdeg[i] > 0]
    T = [i for i in range(V) if outdeg[i] == 0 and indeg[i] > 0]
    print(name, "V", V, "E", len(W), "L", L, "sources", S[:5], "targets", T[:5], "types", o["node_type"][S[0]] if S else None)
    tt = o['node_type']
    s = [i for i in range(V) if tt[i]==2][0]
    T = [i for i in range(V) if tt[i]==3] or T
    print(' src', s, 'tgt', T, 'type counts', np.bincount(tt))
    # sort edges by (pos of dst, pos of src)
    key = np.lexsort((pos[src], pos[dst]))
    edges = [(int(src[i]), int(dst[i]), W[i]) for i in key]
    d, r = bf(V, edges, [s], None)
    print(" true BF rounds", r)
    res = []
    for p0 in range(step, L - 3000, step):
        sub = [(u, v, w) for (u, v, w) in edges if pos[u] >= p0 and pos[v] >= p0 and u != s]
        seeds = [i for i in range(V) if p0 <= pos[i] < p0 + 500 and i != s and i not in T]
        d2, r2 = bf(V, sub, seeds, None)
        # offset profile for nodes with pos >= p0
        idx = [i for i in np.argsort(pos) if pos[i] >= p0 and i != s and i not in T]
        diffs = []
        for i in idx:
            if d[i] is None and d2[i] is None: diffs.append((pos[i], "U"))
            elif d[i] is None or d2[i] is None: diffs.append((pos[i], "X"))
            else: diffs.append((pos[i], d2[i] - d[i]))
        # find last position where diff != final diff
        final = [x for (p, x) in diffs if x != "U"][-1]
        lastbad = p0
        for (p, x) in diffs:
            if x != final and x != "U": lastbad = p
        res.append((p0, lastbad - p0))
    m = [x[1] for x in res]
    print(" margins needed: n", len(m), "max", max(m), "p90", sorted(m)[int(len(m)*0.9)], "median", sorted(m)[len(m)//2])
    print(" worst:", sorted(res, key=lambda t: -t[1])[:8])

if __name__ == "__main__" and len(sys.argv) <= 3:
    main(sys.argv[1], int(sys.argv[2]), sys.argv[1].split("/")[-1])

def debug(path, p0):
    recs = read_fasta_gz(path); seq = recs[1]
    o, V, pos, src, dst, W = graph(seq)
    tt = o['node_type']; s = [i for i in range(V) if tt[i]==2][0]; T=[i for i in range(V) if tt[i]==3]
    key = np.lexsort((pos[src], pos[dst]))
    edges = [(int(src[i]), int(dst[i]), W[i]) for i in key]
    d, r = bf(V, edges, [s], None)
    sub = [(u, v, w) for (u, v, w) in edges if pos[u] >= p0 and pos[v] >= p0 and u != s]
    seeds = [i for i in range(V) if p0 <= pos[i] < p0 + 500 and i != s and i not in T]
    d2, r2 = bf(V, sub, seeds, None)
    idx = [i for i in np.argsort(pos, kind='stable') if pos[i] >= p0 and i != s and i not in T]
    from collections import Counter
    c = Counter()
    rows=[]
    for i in idx:
        if d[i] is None and d2[i] is None: x="U"
        elif d[i] is None or d2[i] is None: x="X"
        else: x=d2[i]-d[i]
        c[x]+=1; rows.append((int(pos[i]), x, int(tt[i]), int(o['node_frame'][i])))
    print(c.most_common(6))
    final = c.most_common(1)[0][0]
    bad=[r for r in rows if r[1]!=final]
    print(len(bad), bad[:10], bad[-10:])
if len(sys.argv) > 3: debug(sys.argv[1], int(sys.argv[3]))
