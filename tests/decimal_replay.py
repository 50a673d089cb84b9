"""TEST INFRASTRUCTURE (moved out of the product package in round 5; the product prints --dump with its own C replay, phx_dump_text /
csrc/phx_exact.inc + phx_dec.c): the reference's Decimal arithmetic replayed with Python's own `decimal` module — the second,
independent checker of the C replay, of k_refine's bounds and of the host re-solve.  Imported by tests/ and tools/ only.

-d/--dump (phanotate.py:58,61): one line per edge of the first contig's graph,
    repr(source) TAB repr(target) TAB str(weight*1000)                      (edges.py:17-23, nodes.py:14-21)
in Graph.iteredges order, with the reference's 28-digit Decimal weights, so that the text can be diffed against an
upstream install line by line.

The GPU computes the weights in fp64; Decimal text cannot be printed from those.  What the GPU path does deliver exactly is
every INTEGER the reference's arithmetic starts from: the ORF table in iter_orfs order (start, stop, frame, RBS bin, start
codon), the GC-frame class of every codon (phx_tap_positions), the RBS / GC-frame training counters and the g+c count
(phx_tap_globals), other_end, and the node and edge lists.  This module replays the reference's Decimal operations on them,
operation by operation and in the same order (Decimal rounds after every operation, so the order matters):
functions.py:174-178 (pstop), orfs.py:162-173 (Orf.p_stop), functions.py:254-257, 281-301 + orfs.py:122-127 (ORF weight),
functions.py:26-46 (score_overlap / score_gap), functions.py:373-385 (o1, o2), and the branch of functions.py:334-452 that
created each edge.  It is host-side formatting of one contig, not part of the per-batch path.
"""
from decimal import Decimal

import numpy as np

TNAME = {0: "start", 1: "stop", 2: "source", 3: "target"}


def start_weights(start_codons):
    """file_handling.py:58-62: codon -> Decimal(weight) / max."""
    w = {}
    for item in start_codons.split(","):
        codon, weight = item.split(":")
        w[codon.lower()] = Decimal(weight)
    m = max(w.values())
    return {k: v / m for k, v in w.items()}


def _overlap(length, diff, pstop):  # functions.py:26-34
    score = 1 / (Decimal(Decimal(1 - pstop)) ** Decimal(length))
    return score + 1 / Decimal("0.05") if diff else score


def _gap(length, diff, pgap):  # functions.py:36-46
    g = Decimal(1 - pgap)
    if length > 300:
        return Decimal(g) ** Decimal(100) + length
    score = 1 / (Decimal(g) ** Decimal(length / 3))
    return score + 1 / Decimal("0.05") if diff else score


class _Lazy:
    """A list whose entries are computed when they are first read."""

    def __init__(self, n, f):
        self.v, self.f = [None] * n, f

    def __getitem__(self, k):
        if self.v[k] is None:
            self.v[k] = self.f(k)
        return self.v[k]


def orf_weights(seq, orf, gcc, gl, weights, start_names):
    """Decimal pstop and weight of every ORF of the tap `orf` (reference order), each computed when it is first asked for."""
    dna = seq.lower()
    comp = {"a": "t", "t": "a", "g": "c", "c": "g"}
    # functions.py:281-284
    pos_max = [Decimal(1) + int(gl.gc_max_count[k]) for k in range(4)]
    pos_min = [Decimal(1) + int(gl.gc_min_count[k]) for k in range(4)]
    y = max(pos_max)
    pos_max = [x / y for x in pos_max]
    y = max(pos_min)
    pos_min = [x / y for x in pos_min]
    fwd_cls = (gcc & 15).tolist()
    rev_cls = (gcc >> 4).tolist()
    def pstop_of(k):
        r = orf[k]
        start, stop, frame = int(r["start"]), int(r["stop"]), int(r["frame"])
        # Orf.p_stop on the coding strand (orfs.py:162-173); letters outside acgt only count in the length
        s = dna[start - 1 : stop + 2] if frame > 0 else dna[stop - 1 : start + 2]
        na, nt, ng = s.count("a"), s.count("t"), s.count("g")
        if frame < 0:
            na, nt, ng = nt, na, s.count("c")
        length = Decimal(len(s))
        Pa, Pt, Pg = na / length, nt / length, ng / length
        return Pt * Pa * Pa + Pt * Pg * Pa + Pt * Pa * Pg

    pstops = _Lazy(len(orf), pstop_of)

    def weight_of(k):
        r = orf[k]
        start, stop, frame = int(r["start"]), int(r["stop"]), int(r["frame"])
        pstop = pstops[k]
        # functions.py:286-298: hold *= ((1-pstop)**pos_max[imax])**pos_min[imin] per sense codon; the factor only depends on
        # the codon's class, the running product is rounded after every multiplication as in the reference
        fac = {}
        hold = 1
        one_minus = 1 - pstop
        if frame > 0:
            classes = fwd_cls[start - 1 : stop - 1 : 3]
        else:
            classes = rev_cls[start - 1 : stop - 1 : -3]
        for c in classes:
            f = fac.get(c)
            if f is None:
                f = fac[c] = (one_minus ** pos_max[c // 3 + 1]) ** pos_min[c % 3 + 1]
            hold = hold * f
        # Orf.score, orfs.py:122-127
        w = 1 / hold
        si = int(r["startidx"])
        if si >= 0:
            w = w * weights[start_names[si]]
        b = int(r["rbs"])
        w = w * Decimal(str(gl.training_rbs[b] / gl.background_rbs[b]))
        return -w

    return pstops, _Lazy(len(orf), weight_of)


from phanotate_amd.functions import edge_order  # noqa: E402  (Graph.iteredges order: part of the product's functions mirror)


def dump_lines(ann, i, seq, start_codons="atg:0.85,gtg:0.10,ttg:0.05"):
    """The --dump text of contig i of the batch `ann` last ran, as a list of lines (no newline)."""
    nd, ed, weight = decimal_weights(ann, i, seq, start_codons)

    def rep(v):
        n = nd[v]
        t = TNAME[int(n["type"])]
        gene = t if n["type"] >= 2 else ("tRNA" if abs(int(n["frame"])) == 4 else "CDS")
        return "Node(%r,%r,%r,%r)" % (gene, t, int(n["frame"]), int(n["pos"]))

    return ["%s\t%s\t%s" % (rep(int(ed[k]["src"])), rep(int(ed[k]["dst"])), str(weight[k] * 1000)) for k in edge_order(nd, ed)]


def decimal_weights(ann, i, seq, start_codons="atg:0.85,gtg:0.10,ttg:0.05", flagged_only=False):
    """(node tap, edge tap, the reference's Decimal weight of every tapped edge, in tap order) of contig i.  flagged_only: Decimal
    arithmetic only for the edges the device marked "inexact"; for every other edge trunc(Decimal(w) * 1000) equals the device's
    integer (that is what the flag says, phx_kernels.hip: cert_eps_is_zero_fast), and the entry is that integer / 1000."""
    if isinstance(seq, (bytes, bytearray)):
        seq = seq.decode()
    gl = ann.globals(i)
    L = int(gl.L)
    nd, ed, orf = ann.nodes(i), ann.edges(i), ann.orfs(i)
    gcc = ann.positions(i)["gcc"]
    weights = start_weights(start_codons)
    start_names = list(weights.keys())
    # functions.py:174-178: both strands are counted, so a == t and g == c
    fa, fg = Decimal(L - int(gl.gc_count)), Decimal(int(gl.gc_count))
    Pa, Pt, Pg = fa / (L * 2), fa / (L * 2), fg / (L * 2)
    pgap = Pt * Pa * Pa + Pt * Pg * Pa + Pt * Pa * Pg
    opstop, oweight = orf_weights(seq, orf, gcc, gl, weights, start_names)
    by_stop = {}
    for k, r in enumerate(orf):
        by_stop.setdefault(int(r["stop"]), {})[int(r["start"])] = k
    other_end = {int(n["pos"]): int(n["other"]) for n in nd if n["type"] < 2 and abs(int(n["frame"])) != 4}

    def o_term(p):  # functions.py:373-384
        if p in by_stop and other_end[p] in by_stop[p]:
            return opstop[by_stop[p][other_end[p]]]
        if p in by_stop:
            q = other_end[p]
            if q in by_stop and p in by_stop[q]:
                return opstop[by_stop[q][p]]
            return pgap  # (the reference raises here; libphx reports such a contig through its status)
        return pgap

    import math

    gap_memo = {}

    def _gap_m(length, diff):
        key = (length, diff)
        if key not in gap_memo:
            gap_memo[key] = _gap(length, diff, pgap)
        return gap_memo[key]

    weight = []
    typ, frm, pos = nd["type"].tolist(), nd["frame"].tolist(), nd["pos"].tolist()
    for s, d, wf, ix in zip(ed["src"].tolist(), ed["dst"].tolist(), ed["w"].tolist(), ed["inexact"].tolist()):
        if flagged_only and not ix:
            weight.append(Decimal(math.trunc(wf * 1000.0)) / 1000)
            continue
        ts, td = typ[s], typ[d]
        fs, fd = frm[s], frm[d]
        ps, pd = pos[s], pos[d]
        if ts < 2 and td < 2 and fs == fd and abs(fs) == 4 and ((fs > 0 and ts == 0 and td == 1) or (fs < 0 and ts == 1 and td == 0)):
            w = -Decimal(20)  # the tRNA edge, functions.py:509
        elif ts < 2 and td < 2 and fs == fd and ((fs > 0 and ts == 0 and td == 1) or (fs < 0 and ts == 1 and td == 0)):
            # ORF edge start -> stop / stop -> start (functions.py:311-318)
            start, stop = (ps, pd) if fs > 0 else (pd, ps)
            w = oweight[by_stop[stop][start]]
        elif ts == 2:  # functions.py:445-448
            w = _gap_m(pd, False)
        elif td == 3:  # functions.py:449-452
            w = _gap_m(L - ps, False)
        else:
            diff = fs * fd < 0 and abs(fs) != 4 and abs(fd) != 4  # a pair with a tRNA node is scored 'same' on both strand combinations (functions.py:388-399)
            if ps < pd:  # left -> right: a gap edge of the connect loop, or a bridge over a non-coding run (same formula)
                w = _gap_m(pd - ps - 3, diff)
            else:  # right -> left: overlap edge, pstop = ave([o1, o2]) (functions.py:385)
                pst = Decimal((o_term(pd) + o_term(ps)) / 2)
                w = _overlap(ps - pd + 3, diff, pst)
        weight.append(w)
    return nd, ed, weight


def python_resolve(ann, i, seq, start_codons="atg:0.85,gtg:0.10,ttg:0.05"):
    """The genes of contig i as the reference's own integers give them, in Python: Decimal weights of the edges the device flagged
    (decimal_weights), the solver's in-place Bellman-Ford over Graph.iteredges order with a strict '<' (phanotate.py:56-64;
    tests/golden/make_golden.py:100-133) in python ints.  The library does the same below the C-ABI (csrc/phx_exact.inc); this is the
    cross-check the tests hold it against.  Returns a list of (left, right, strand, frame, score)."""
    nd, ed, wdec = decimal_weights(ann, i, seq, start_codons, flagged_only=True)
    V = len(nd)
    esrc, edst = ed["src"].tolist(), ed["dst"].tolist()
    E = [(esrc[k], edst[k], int(wdec[k] * 1000), k) for k in edge_order(nd, ed)]
    dist, par = [None] * V, [-1] * V
    dist[V - 2] = 0
    for _ in range(V + 1):
        ch = False
        for u, v, w, k in E:
            du = dist[u]
            if du is not None and (dist[v] is None or du + w < dist[v]):
                dist[v] = du + w; par[v] = k; ch = True
        if not ch:
            break
    if V < 2 or dist[V - 1] is None:
        return []
    pe, v = [], V - 1
    while v != V - 2:
        pe.append(par[v]); v = esrc[par[v]]
    pe.reverse()  # edges source -> target; shortest_path[1:] pairwise = every second edge (phanotate.py:65-76)
    out = []
    for k in pe[1::2]:
        a, b_ = int(ed[k]["src"]), int(ed[k]["dst"])
        out.append((int(nd[a]["pos"]), int(nd[b_]["pos"]) + 2, -1 if nd[a]["frame"] < 0 else 1, int(nd[a]["frame"]), float(ed[k]["w"])))
    return out
