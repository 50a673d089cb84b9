#!/usr/bin/env python3
"""Prototype of the Decimal-vs-fp64 certificate, run on the GPU box:   python tools/certify_probe.py [n] [seed]

libphx solves on W = trunc(fp64(w) * 1000), the reference on W* = trunc(Decimal(w) * 1000).  With a per-edge bound
|W* - W| <= eps_e, the device's path P is the reference's path whenever a dual certificate holds (LP duality on the scenario
"P's edges as heavy, every other edge as light as the bounds allow"):
  tree = the device's shortest-path tree (P = its path to the target), sigma[v] = signed sum of eps along the tree path to v
  (+ on P, - off P), kappa[v] = last node on that path whose tree edge has eps > 0.  For every non-tree edge e = (u -> v) of a
  reached u:   r(e) + sigma[u] - sigma[v] - eps_e > 0          (r = d[u] + W_e - d[v], exact)
  or  r(e) == 0 and eps_e == 0 and kappa[u] == kappa[v]        (a tie that is exact in the reference's integers too).
The probe checks (1) that eps_e really bounds |W* - W| on every edge (W* replayed by phanotate_amd/dump.py), (2) how many
contigs the certificate covers, (3) that for the covered ones the Decimal-derived in-order solve gives the same path."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import phanotate_amd as pa
from decimal_check import solve
from fuzz_gpu import make
from phanotate_amd.dump import decimal_weights


def eps_of(w, inexact, scale=1.0):
    """Bound on |trunc(Decimal w * 1000) - trunc(fp64 w * 1000)|: 0 for an edge the device did not flag (its p = w * 1000 is farther
    from the next integer than its error bound), else from the integer the solver uses — cert_eps in phx_certify.inc, the same
    operations in the same order."""
    if not inexact:
        return 0
    t = math.trunc(w * 1000.0)
    a = float(abs(t)) + 1.0 if abs(t) < 2 ** 62 else abs(float(t))
    ex = abs(math.frexp(a)[1])
    err = a * float(ex + 8) * (scale * 2.0 ** -46)
    return int(math.floor(err)) + 1


def flag_of(w, scale=1.0):
    """The device's inexact flag, restated (cert_eps_is_zero_fast, phx_kernels.hip)."""
    if w == -20.0:
        return 0  # the tRNA edge: a constant in the reference too (functions.py:509)
    p = w * 1000.0
    a = abs(p)
    if a < 2.0 ** 52:
        f = a - abs(float(math.trunc(p)))
        if min(f, 1.0 - f) > a * (60.0 * (scale * 2.0 ** -46)):
            return 0
    return 1


def certify(nd, ed, dist, path, scale=1.0, repair=True):
    """The certificate of phx_certify.inc in python ints.  repair=False: the form of k_certify_wide (no corrections)."""
    V = len(nd)
    src, dst = ed["src"].tolist(), ed["dst"].tolist()
    W = [int(math.trunc(float(x) * 1000.0)) for x in ed["w"]]
    flags = ed["inexact"].tolist()
    eps = [eps_of(float(x), f, scale) for x, f in zip(ed["w"], flags)]
    on_path_edge = {}
    for a, b in zip(path[:-1], path[1:]):
        on_path_edge[b] = a
    # tree edge of every reached node: the path's edge on P, else the lowest-index tight in-edge
    tree = [-1] * V
    for k in range(len(src)):
        u, v = src[k], dst[k]
        if dist[u] is None or dist[v] is None or dist[u] + W[k] != dist[v]:
            continue
        if v in on_path_edge:
            if on_path_edge[v] == u and (tree[v] < 0 or src[tree[v]] != u):
                tree[v] = k
        elif tree[v] < 0:
            tree[v] = k
    s = V - 2
    onP = set(path)
    sigma, kappa = [None] * V, [None] * V
    sigma[s], kappa[s] = 0, -1
    pending = [v for v in range(V) if dist[v] is not None and v != s]
    while pending:  # parents before children
        nxt = []
        for v in pending:
            k = tree[v]
            if k < 0:
                return "no tree edge", 0
            u = src[k]
            if sigma[u] is None:
                nxt.append(v)
                continue
            sign = 1 if (v in onP and on_path_edge.get(v) == u) else -1
            sigma[v] = sigma[u] + sign * eps[k]
            kappa[v] = v if eps[k] > 0 else kappa[u]
        if len(nxt) == len(pending):
            return "tree cycle", 0
        pending = nxt
    delta = [0] * V
    for rnd in range(26):
        changed = False
        for k in range(len(src)):
            u, v = src[k], dst[k]
            if dist[u] is None:
                continue
            is_tree = tree[v] == k
            if is_tree and (v in onP or rnd == 0):
                continue
            if rnd > 0 and delta[u] == 0:
                continue
            r = dist[u] + W[k] - dist[v]
            if r < 0:
                return "negative reduced cost", 0
            r0 = 0 if is_tree else r + sigma[u] - sigma[v] - eps[k]
            need = delta[u] - r0
            if v in onP:
                if not (need < 0 or (r == 0 and eps[k] == 0 and kappa[u] == kappa[v] and delta[u] == 0)):
                    return "edge %d into the path: r=%d r0=%d eps=%d delta[u]=%d" % (k, r, r0, eps[k], delta[u]), 0
            elif need > delta[v]:
                if not repair or need > 60000:
                    return "edge %d off the path: r=%d r0=%d eps=%d" % (k, r, r0, eps[k]), 0
                delta[v] = need
                changed = True
        if not changed:
            return "", 1
    return "corrections do not settle", 0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    seqs = []
    while len(seqs) < n:
        s = make(rng)
        if len(s) <= 12000 and not set(s.lower()) - set("acgt"):
            seqs.append(s)
    seqs += [pa.synth_contig(2000 + k, 50000).decode() for k in range(max(2, n // 10))]
    ann = pa.Annotator()
    t0 = time.time()
    n_ok = n_cert = n_same = n_diff = bad_eps = n_edges = n_inexact = ties_cert = 0
    reasons = []
    for b0 in range(0, len(seqs), 50):
        part = seqs[b0 : b0 + 50]
        res = ann.annotate(part)
        for i, (status, genes) in enumerate(res):
            gl = ann.globals(i)
            if status < 0 or gl.n_node <= 2:
                continue
            n_ok += 1
            nd, ed, wdec = decimal_weights(ann, i, part[i])
            Wref = [int(w * 1000) for w in wdec]
            Wdev = [int(math.trunc(float(x) * 1000.0)) for x in ed["w"]]
            for k in range(len(ed)):
                assert int(ed["inexact"][k]) == flag_of(float(ed["w"][k])), "the device's inexact flag differs from its statement"
                e = eps_of(float(ed["w"][k]), int(ed["inexact"][k]))
                n_edges += 1
                n_inexact += e > 0
                if abs(Wref[k] - Wdev[k]) > e:
                    bad_eps += 1
                    if bad_eps <= 5:
                        print("EPS TOO SMALL contig %d edge %d: w=%r dev %d ref %d eps %d" % (b0 + i, k, float(ed["w"][k]), Wdev[k], Wref[k], e))
            dist = ann.dist(i)
            path = [int(x) for x in ann.path(i)[0]]
            why, ok = certify(nd, ed, dist, path)
            p_dec = solve(nd, ed, wdec)
            same = p_dec == path
            n_cert += ok
            ties_cert += ok and gl.tie != 0
            if ok:
                n_same += same
                n_diff += not same
                if not same:
                    print("CERTIFIED BUT DIFFERENT contig %d" % (b0 + i))
            else:
                reasons.append((b0 + i, len(part[i]), int(gl.tie), why, same))
    print("certify probe seed %d: %d contigs, %d certified (%d of them with ties), of those %d same / %d different Decimal path; eps violated on %d of %d edges (%d inexact); %.0f s"
          % (seed, n_ok, n_cert, ties_cert, n_same, n_diff, bad_eps, n_edges, n_inexact, time.time() - t0))
    for r in reasons[:40]:
        print("  uncertified contig %d (L %d, tie flag %d): %s; Decimal path %s" % (r[0], r[1], r[2], r[3], "same" if r[4] else "DIFFERENT"))


if __name__ == "__main__":
    main()
