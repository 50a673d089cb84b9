"""Mirror of the reference's operator interface for the hot path, backed by libphx.

    orfs  = functions.get_orfs(locus)    # phanotate.py:45, functions.py:143-303
    graph = functions.get_graph(orfs)    # phanotate.py:49, functions.py:307-454

Same names, argument meaning and error behaviour as phanotate_modules.functions: `locus` only needs
.seq(), .start_codons (dict codon -> weight/max), .stop_codons (list) and .min_orf_len (phanotate.py:42-44,
orfs.py:8-15).  The objects returned are read-only views of what the GPU computed for that contig (ORF table
in the reference's iter_orfs order, nodes/edges in iternodes/iteredges order); weights are fp64 where the
reference holds Decimal.  Inputs on which the reference raises, raise here too (KeyError for a letter
outside the nucleotide alphabet, functions.py:20-24).
"""
from collections import OrderedDict

import numpy as np

from .api import Annotator, make_params

_cache = {}


def _annotator(locus):
    starts = ",".join("%s:%r" % (k, float(v)) for k, v in locus.start_codons.items())
    key = (starts, ",".join(locus.stop_codons), int(locus.min_orf_len))
    if key not in _cache:
        _cache[key] = Annotator(make_params(*key))
    return _cache[key]


class Orf:
    """orfs.py:71-95 (the fields the path uses)."""

    def __init__(self, rec, params):
        self.start, self.stop, self.frame, self.length = int(rec["start"]), int(rec["stop"]), int(rec["frame"]), int(rec["length"])
        self.rbs_score = int(rec["rbs"])
        self.pstop = float(rec["pstop"])
        self.weight_rbs = float(rec["weight_rbs"])
        self.weight = float(rec["weight"])
        self.S = float(rec["S"])
        self.hist = np.array(rec["hist"])

    def __repr__(self):
        return "Orf(%r,%r,%r,%r,%r)" % (self.start, self.stop, self.frame, self.weight_rbs, self.weight)


class Orfs(OrderedDict):
    """orfs.py:6-69: self[stop][start] -> Orf, plus other_end / pstop / contig_length."""

    def iter_orfs(self):
        for stop in self:
            for start in self[stop]:
                yield self[stop][start]

    def get_orf(self, start, stop):
        if stop in self:
            if start in self[stop]:
                return self[stop][start]
            raise ValueError("orf with start codon not found")
        raise ValueError(" orf with stop codon not found")


def get_orfs(locus):
    ann = _annotator(locus)
    seq = locus.seq()
    # tRNA masking belongs to get_graph in the reference (functions.py:357); the GPU builds ORFs and graph in one pass, so
    # the finders run here (phanotate_amd.trna does what functions.py:457-495 does, including the warning)
    from .trna import find_trnas

    hits = find_trnas(seq)
    (status, genes), = ann.annotate([seq], trnas=None if hits is None else [hits])
    if status == -2:
        raise KeyError("letter outside the nucleotide alphabet")  # rev_comp, functions.py:20-24
    if status == -3:
        raise UnboundLocalError("contig too short for the GC frame plot")  # gc_frame_plot.py:64-69
    if status < 0:
        raise ValueError("libphx status %d" % status)
    orfs = Orfs()
    gl = ann.globals(0)
    orfs.pstop = gl.pstop
    orfs.contig_length = len(seq)
    orfs.min_orf_len = locus.min_orf_len
    orfs.start_codons, orfs.stop_codons = locus.start_codons, locus.stop_codons
    orfs.seq = seq.lower()
    for rec in ann.orfs(0):
        orfs.setdefault(int(rec["stop"]), OrderedDict())[int(rec["start"])] = Orf(rec, ann.params)
    nd = ann.nodes(0)
    orfs.other_end = {int(n["pos"]): int(n["other"]) for n in nd if n["type"] < 2 and abs(int(n["frame"])) != 4}
    orfs.other_end.update({"t" + str(int(n["pos"])): int(n["other"]) for n in nd if n["type"] < 2 and abs(int(n["frame"])) == 4})  # functions.py:502-508
    orfs._ann, orfs._genes, orfs._status = ann, genes, status
    return orfs


class Node:
    """nodes.py:2-21."""

    def __init__(self, gene, type, frame, position):
        self.gene, self.type, self.frame, self.position = gene, type, frame, position

    def __repr__(self):
        return "Node(%r,%r,%r,%r)" % (self.gene, self.type, self.frame, self.position)

    def __hash__(self):
        return hash(repr(self))

    def __eq__(self, other):
        return hash(self) == hash(other)


class Edge:
    """edges.py:3-23."""

    def __init__(self, source, target, weight):
        self.source, self.target, self.weight = source, target, weight

    def __str__(self):
        return "%s\t%s\t%s" % (repr(self.source), repr(self.target), repr(self.weight * 1000))


class Graph(OrderedDict):
    """graphs.py: dict-of-dict adjacency in insertion order; iternodes / iteredges / weight."""

    def iternodes(self):
        return self.keys()

    def iteredges(self):
        for s in self:
            for t in self[s]:
                yield self[s][t]

    def weight(self, edge):
        if edge.source in self and edge.target in self[edge.source]:
            return self[edge.source][edge.target].weight
        return 0


def edge_order(nd, ed):
    """Indices of the tapped edges `ed` in Graph.iteredges order (graphs.py:121-126): by insertion rank of the source node, then
    in the order get_graph adds a node's out-edges: its ORF edge(s) (functions.py:311-318), bridges (334-354), the tRNA edge
    (509), the connect loop (360-438: right node outer, left node inner), source / target edges (440-452)."""
    ref, typ, frm, pos = nd["refidx"].tolist(), nd["type"].tolist(), nd["frame"].tolist(), nd["pos"].tolist()  # (lists: structured scalars are slow)
    keys = []
    for k, (s, d) in enumerate(zip(ed["src"].tolist(), ed["dst"].tolist())):
        ts, td, fs, fd = typ[s], typ[d], frm[s], frm[d]
        if ts < 2 and td < 2 and fs == fd and ((fs > 0 and ts == 0 and td == 1) or (fs < 0 and ts == 1 and td == 0)):
            cls = (1.5 if abs(fs) == 4 else 0, ref[d], 0)
        elif ts == 2 or td == 3:
            cls = (3, ref[d], 0)
        else:
            l, r = (s, d) if pos[s] < pos[d] else (d, s)
            cls = (1 if abs(pos[s] - pos[d]) >= 500 else 2, ref[r], ref[l])
        keys.append((ref[s], cls, k))
    keys.sort()
    return [k for _, _, k in keys]


def get_graph(my_orfs):
    """The graph libphx built for the contig of `my_orfs`, in the reference's insertion order; the weights are the device's
    fp64 values (phx_tap_edges), unchanged."""
    ann = my_orfs._ann
    nd = ann.nodes(0)
    ed = ann.edges(0)
    tname = {0: "start", 1: "stop", 2: "source", 3: "target"}
    order = np.argsort(nd["refidx"], kind="stable")
    by_id = {}
    G = Graph()
    for v in order:
        n = nd[v]
        t = tname[int(n["type"])]
        node = Node(t if n["type"] >= 2 else ("tRNA" if abs(int(n["frame"])) == 4 else "CDS"), t, int(n["frame"]), int(n["pos"]))
        by_id[int(v)] = node
        G[node] = OrderedDict()
    for k in edge_order(nd, ed):
        s, d = int(ed[k]["src"]), int(ed[k]["dst"])
        G[by_id[s]][by_id[d]] = Edge(by_id[s], by_id[d], float(ed[k]["w"]))
    return G
