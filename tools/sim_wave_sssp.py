#!/usr/bin/env python3
"""CPU prototype of the single-wavefront SSSP schedule (window = ADV advance nodes + 500 bp look-ahead,
A/B phases, rollback when a close node near the window start changes).  Checks distances/paths against the
oracle and reports rounds per window and rollback rates.  Development tool, not part of the product."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle

lib = C.CDLL(os.path.join(ROOT, 'phanotate_amd', 'libphx.so'))
def synth(seed, L):
    b = C.create_string_buffer(L); lib.phx_synth_contig(C.c_uint64(seed), C.c_int64(L), b); return b.raw[:L].decode()

def run(seed, L=50000, ADV=64, CAPC=64, CAPO=64, ECAP=1280, verbose=False):
    r = oracle.run(synth(seed, L), stages=3)
    t = r['node_type']; f = r['node_frame']; pos0 = r['node_pos']
    src0 = r['edge_src']; dst0 = r['edge_dst']
    wint = [oracle.limbs_to_int(x) for x in r['edge_wint_limbs']]
    V = len(t)
    # device order: real nodes sorted by position (stable), then source, target
    real = [i for i in range(V) if t[i] < 2]
    real.sort(key=lambda i: (pos0[i],))  # oracle's order among equal pos kept (stable)
    s_id = [i for i in range(V) if t[i] == 2][0]; t_id = [i for i in range(V) if t[i] == 3][0]
    order = real + [s_id, t_id]
    newid = {o: n for n, o in enumerate(order)}
    pos = [int(pos0[o]) for o in order]; pos[V-2] = pos[V-1] = 1 << 29
    typ = [int(t[o]) for o in order]; frm = [int(f[o]) for o in order]
    close = [(typ[i] == 1 and frm[i] > 0) or (typ[i] == 0 and frm[i] < 0) for i in range(V)]
    inedges = [[] for _ in range(V)]
    for e in range(len(src0)):
        inedges[newid[int(dst0[e])]].append((newid[int(src0[e])], wint[e]))
    for v in range(V): inedges[v].sort(key=lambda x: x[0])
    deg = [len(x) for x in inedges]
    SRC, TGT, ncds = V-2, V-1, V-2
    INF = 1 << 200
    d = [INF]*V; d[SRC] = 0
    par = [-1]*V
    def plan(v0):
        # extent limited by class caps and edge cap
        nc = no = ne = 0; j = 0
        while v0 + j < V and j < 128:
            c = v0 + j
            nc2 = nc + (1 if close[c] else 0); no2 = no + (0 if close[c] else 1); ne2 = ne + deg[c]
            if nc2 > CAPC or no2 > CAPO or ne2 > ECAP: break
            nc, no, ne = nc2, no2, ne2; j += 1
        nmax = j
        if v0 + nmax >= V: amax = nmax
        else:
            lim = pos[v0 + nmax]
            amax = 0
            while amax < nmax and pos[v0 + amax] + 500 <= lim: amax += 1
        a = min(ADV, amax)
        if a == 0: raise RuntimeError('cannot advance')
        va = v0 + a
        v1 = va
        while v1 < v0 + nmax and pos[v1] < pos[va-1] + 500: v1 += 1
        return va, v1
    v0 = 0; nwin = 0; rounds = 0; phases = 0; rollbacks = 0; maxdegA = []; maxdegB = []; sizes = []
    while v0 < V:
        va, v1 = plan(v0)
        nodes = range(v0, v1)
        A = [v for v in nodes if close[v]]; B = [v for v in nodes if not close[v] and v != SRC]
        sizes.append((v1 - v0, len(A), len(B), sum(deg[v] for v in nodes)))
        maxdegA.append(max([deg[v] for v in A], default=0)); maxdegB.append(max([deg[v] for v in B], default=0))
        old = {v: d[v] for v in A}
        first = True; ph = 0
        while True:
            grp = B if ph else A
            chg = False
            new = {}
            for v in grp:
                best = d[v]
                for (u, w) in inedges[v]:
                    if u >= v1 and u != SRC: continue  # beyond the window: not loaded yet
                    c = d[u] + w
                    if c < best: best = c
                if best < d[v]: new[v] = best
            for v, x in new.items(): d[v] = x; chg = True
            phases += 1
            if not chg and (ph == 1 or not first): break
            first = False; ph ^= 1
        nwin += 1
        # parents for advance nodes
        for v in range(v0, va):
            par[v] = -1
            if v == SRC or d[v] >= INF: continue
            for (u, w) in inedges[v]:
                if (u < v1 or u == SRC) and d[u] + w == d[v]: par[v] = u; break
        # trigger: a close node within 500 bp of the window start changed
        trig = [v for v in A if v0 > 0 and d[v] < old[v] and pos[v] < pos[v0-1] + 500]
        if trig:
            rollbacks += 1
            x = trig[0]
            rr = v0
            while rr > 0 and pos[rr-1] > pos[x] - 500: rr -= 1
            v0 = rr
            if rollbacks > 1000: raise RuntimeError('rollback storm')
            continue
        v0 = va
    # check against the oracle
    od = int(r['path_dist'])
    okd = (od is None) or d[TGT] == od
    path = [TGT]
    while path[-1] != SRC and len(path) <= V: path.append(par[path[-1]])
    path.reverse()
    opath = [newid[int(x)] for x in r['path']]
    # full fixed-point verification
    viol = 0
    for v in range(V):
        for (u, w) in inedges[v]:
            if d[u] < INF and d[u] + w < d[v]: viol += 1
    s = np.array(sizes)
    return dict(seed=seed, V=V, E=len(src0), windows=nwin, phases=phases, rollbacks=rollbacks, viol=viol, dist_ok=okd, path_ok=path == opath,
                win_nodes=s[:, 0].mean(), win_close=s[:, 1].mean(), win_open=s[:, 2].mean(), win_edges=s[:, 3].mean(), win_edges_max=s[:, 3].max(),
                maxdegA=float(np.mean(maxdegA)), maxdegB=float(np.mean(maxdegB)), maxB=max(maxdegB), maxA=max(maxdegA))

if __name__ == '__main__':
    adv = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    tot = {}
    for seed in range(n):
        o = run(seed, ADV=adv)
        print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in o.items()})
