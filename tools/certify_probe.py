#!/usr/bin/env python3
"""Prototype of the Decimal-vs-fp64 certificate, run on the GPU box:   python tools/certify_probe.py [n] [seed]

libphx solves on W = trunc(fp64(w) * 1000), the reference on W* = trunc(Decimal(w) * 1000).  With per-edge bounds
lo_e <= W* - W <= hi_e (k_refine, phx_refine.inc: the flagged edges once more in double-double + the error bound of the reference's own
28-digit chain), the device's path P is the reference's path whenever a dual certificate holds (LP duality on the scenario
"P's edges as heavy, every other edge as light as the bounds allow"):
  tree = the device's shortest-path tree (P = its path to the target), sigma[v] = sum of (hi on P, lo off P) along the tree path to v,
  kappa[v] = last node on that path whose tree edge is flagged.  For every non-tree edge e = (u -> v) of a reached u:
  r(e) + sigma[u] - sigma[v] + lo_e > 0                        (r = d[u] + W_e - d[v], exact)
  or  r(e) == 0 and e not flagged and kappa[u] == kappa[v]     (a tie that is exact in the reference's integers too).
The probe checks (1) that the bounds really hold W* on every edge (W* replayed by tests/decimal_replay.py with Python's decimal), (2) how
many contigs the certificate covers, (3) that for the covered ones the Decimal-derived in-order solve gives the same path."""
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
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
from decimal_replay import decimal_weights


def bounds_of(ed):
    """Per edge (lo, hi) with lo <= W* - W <= hi as the device states them after k_refine (phx_refine.inc; the edge tap's inexact, d1, d2,
    err): (0, 0) for an edge whose flag is cleared, else D -+ eps with D = d1 + d2 and eps = 0 if err == 0 else floor(err) + 1;
    None where nothing is known (err = inf)."""
    out = []
    for f, d1, d2, err in zip(ed["inexact"].tolist(), ed["d1"].tolist(), ed["d2"].tolist(), ed["err"].tolist()):
        if not f:
            out.append((0, 0))
        elif not err < 1e290:
            out.append(None)
        else:
            D = int(d1) + int(d2)
            eps = 0 if err == 0 else int(math.floor(err)) + 1
            out.append((D - eps, D + eps))
    return out


def device_int(w):
    """The integer the solver adds for an edge whose tapped fp64 weight is w (ew_encode, phx_kernels.hip)."""
    return int(math.trunc(w * 1000.0))


def certify(nd, ed, dist, path, repair=True, ties=None):
    """The certificate of phx_certify.inc in python ints, on the bounds the device states (bounds_of).  repair=False: the form of
    k_certify_wide (no corrections).  ties: the contig has equal-length alternatives (phx_globals.tie != 0): a flagged edge into a node
    off the path must then keep slack."""
    V = len(nd)
    src, dst = ed["src"].tolist(), ed["dst"].tolist()
    W = [device_int(float(x)) for x in ed["w"]]
    flags = ed["inexact"].tolist()
    bnd = bounds_of(ed)
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
            if flags[k] and bnd[k] is None:
                return "tree edge %d without bounds" % k, 0
            on = v in onP and on_path_edge.get(v) == u
            sigma[v] = sigma[u] + (bnd[k][1] if on else bnd[k][0])  # as heavy as allowed on the path, as light off it
            kappa[v] = v if flags[k] else kappa[u]
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
            if flags[k] and bnd[k] is None and not is_tree:
                return "edge %d without bounds" % k, 0
            r0 = 0 if is_tree else r + sigma[u] - sigma[v] + bnd[k][0]
            need = delta[u] - r0
            if v in onP:
                if not (need < 0 or (r == 0 and not flags[k] and kappa[u] == kappa[v] and delta[u] == 0)):
                    return "edge %d into the path: r=%d r0=%d bounds=%s delta[u]=%d" % (k, r, r0, bnd[k], delta[u]), 0
            elif need == 0:
                if flags[k] and ties and not is_tree:
                    return "edge %d off the path: no slack on a flagged edge of a contig with ties" % k, 0
            elif need > delta[v]:
                if not repair or need > 60000:
                    return "edge %d off the path: r=%d r0=%d bounds=%s" % (k, r, r0, bnd[k]), 0
                delta[v] = need
                changed = True
        if not changed:
            return "", 1
    return "corrections do not settle", 0


def bounds_hold(ed, wdec):
    """The soundness of what the device states: for every edge the reference's integer W* = int(Decimal weight * 1000) lies inside the
    bounds (== W where the flag is cleared).  wdec: dump.decimal_weights' Decimal weights in tap order.  Returns the violating edges."""
    bad = []
    for k, (w, b) in enumerate(zip(ed["w"].tolist(), bounds_of(ed))):
        if b is None:
            continue
        d = int(wdec[k] * 1000) - device_int(w)
        if not (b[0] <= d <= b[1]):
            bad.append((k, w, d, b))
    return bad


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
    ann = pa.Annotator(flags=("no_exact",))
    t0 = time.time()
    n_ok = n_cert = n_same = n_diff = bad_b = n_edges = n_flag = n_eps = ties_cert = 0
    reasons = []
    for b0 in range(0, len(seqs), 50):
        part = seqs[b0 : b0 + 50]
        ann.upload(part); ann.run()
        cert = ann.certified()
        for i in range(len(part)):
            gl = ann.globals(i)
            if gl.status < 0 or gl.n_node <= 2:
                continue
            n_ok += 1
            nd, ed, wdec = decimal_weights(ann, i, part[i])
            n_edges += len(ed); n_flag += int(ed["inexact"].sum()); n_eps += int(((ed["inexact"] != 0) & (ed["err"] != 0)).sum())
            viol = bounds_hold(ed, wdec)
            bad_b += len(viol)
            for v in viol[:3]:
                print("BOUNDS VIOLATED contig %d edge %d: w=%r W*-W=%d bounds %s" % ((b0 + i,) + v))
            dist = ann.dist(i)
            path = [int(x) for x in ann.path(i)[0]]
            why, ok = certify(nd, ed, dist, path, ties=gl.tie != 0)
            assert ok == int(cert[i]), (b0 + i, why, int(cert[i]))
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
    print("certify probe seed %d: %d contigs, %d certified (%d of them with ties), of those %d same / %d different Decimal path; bounds violated on %d of %d edges (%d still flagged after k_refine, %d of them with eps > 0); %.0f s"
          % (seed, n_ok, n_cert, ties_cert, n_same, n_diff, bad_b, n_edges, n_flag, n_eps, time.time() - t0))
    for r in reasons[:40]:
        print("  uncertified contig %d (L %d, tie flag %d): %s; Decimal path %s" % (r[0], r[1], r[2], r[3], "same" if r[4] else "DIFFERENT"))


if __name__ == "__main__":
    main()
