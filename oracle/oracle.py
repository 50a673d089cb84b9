"""ctypes front-end of the CPU oracle (oracle/phx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker / reported baseline.  The product package
(phanotate_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAXC = 16


class Params(C.Structure):
    _fields_ = [
        ("minlen", C.c_int32),
        ("n_start", C.c_int32),
        ("start", (C.c_char * 4) * MAXC),
        ("start_w", C.c_double * MAXC),
        ("n_stop", C.c_int32),
        ("stop", (C.c_char * 4) * MAXC),
    ]


class Orf(C.Structure):
    _fields_ = [
        ("start", C.c_int32), ("stop", C.c_int32), ("frame", C.c_int32), ("length", C.c_int32), ("rbs", C.c_int32),
        ("first3_is_start", C.c_int32), ("first3_is_atg", C.c_int32), ("hist", C.c_int32 * 9),
        ("pstop", C.c_double), ("weight_rbs", C.c_double), ("S", C.c_double), ("weight", C.c_double),
    ]


class Node(C.Structure):
    _fields_ = [("type", C.c_int8), ("frame", C.c_int8), ("pos", C.c_int32)]


class Edge(C.Structure):
    _fields_ = [("src", C.c_int32), ("dst", C.c_int32), ("w", C.c_double), ("wint", C.c_uint64 * 4)]


class Gene(C.Structure):
    _fields_ = [("left", C.c_int32), ("right", C.c_int32), ("strand", C.c_int32), ("frame", C.c_int32), ("score", C.c_double)]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("L", C.c_int64), ("pstop", C.c_double),
        ("background_rbs", C.c_double * 28), ("training_rbs", C.c_double * 28),
        ("pos_max", C.c_double * 4), ("pos_min", C.c_double * 4),
        ("gcpf", C.POINTER(C.c_uint8)), ("n_gcpf", C.c_int32),
        ("n_orf", C.c_int32), ("orf", C.POINTER(Orf)),
        ("other_end", C.POINTER(C.c_int32)),
        ("n_node", C.c_int32), ("node", C.POINTER(Node)),
        ("n_edge", C.c_int32), ("edge", C.POINTER(Edge)),
        ("n_path", C.c_int32), ("path", C.POINTER(C.c_int32)),
        ("dist", C.c_uint64 * 4), ("bf_rounds", C.c_int32),
        ("n_gene", C.c_int32), ("gene", C.POINTER(Gene)),
        ("binF", C.POINTER(C.c_uint8)), ("binR", C.POINTER(C.c_uint8)),
        ("other_end_t", C.POINTER(C.c_int32)),
        ("wide", C.c_int32), ("pad_wide", C.c_int32), ("dist_wide", C.c_uint64 * 20),
    ]


ORF_DT = np.dtype([("start", "i4"), ("stop", "i4"), ("frame", "i4"), ("length", "i4"), ("rbs", "i4"),
                   ("first3_is_start", "i4"), ("first3_is_atg", "i4"), ("hist", "i4", (9,)),
                   ("pstop", "f8"), ("weight_rbs", "f8"), ("S", "f8"), ("weight", "f8")], align=True)
EDGE_DT = np.dtype([("src", "i4"), ("dst", "i4"), ("w", "f8"), ("wint", "u8", (4,))], align=True)
NODE_DT = np.dtype([("type", "i1"), ("frame", "i1"), ("pos", "i4")], align=True)
GENE_DT = np.dtype([("left", "i4"), ("right", "i4"), ("strand", "i4"), ("frame", "i4"), ("score", "f8")], align=True)

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _lib = C.CDLL(so)
        _lib.orc_run.argtypes = [C.c_char_p, C.c_int64, C.POINTER(Params), C.c_int, C.POINTER(Result)]
        _lib.orc_run.restype = C.c_int
        _lib.orc_run_trna.argtypes = [C.c_char_p, C.c_int64, C.POINTER(Params), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Result)]
        _lib.orc_run_trna.restype = C.c_int
        _lib.orc_free.argtypes = [C.POINTER(Result)]
        _lib.orc_sizeof_params.restype = C.c_size_t
        _lib.orc_score_rbs.argtypes = [C.c_char_p, C.c_int]
        _lib.orc_score_rbs.restype = C.c_int
        _lib.orc_sizeof_result.restype = C.c_size_t
        assert _lib.orc_sizeof_params() == C.sizeof(Params)
        assert _lib.orc_sizeof_result() == C.sizeof(Result)
        assert ORF_DT.itemsize == C.sizeof(Orf) and EDGE_DT.itemsize == C.sizeof(Edge)
        assert NODE_DT.itemsize == C.sizeof(Node) and GENE_DT.itemsize == C.sizeof(Gene)
    return _lib


def make_params(start_codons="atg:0.85,gtg:0.10,ttg:0.05", stop_codons="tag,tga,taa", minlen=90):
    """Mirror of file_handling.py:58-66 in fp64 (weights divided by their max)."""
    p = Params()
    p.minlen = minlen
    pairs = [x.split(":") for x in start_codons.split(",")]
    m = max(float(w) for _, w in pairs)
    p.n_start = len(pairs)
    for i, (c, w) in enumerate(pairs):
        p.start[i].value = c.lower().encode()
        p.start_w[i] = float(w) / m
    stops = stop_codons.split(",")
    p.n_stop = len(stops)
    for i, c in enumerate(stops):
        p.stop[i].value = c.lower().encode()
    return p


def _view(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dt)
    raw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n * dt.itemsize,))
    return raw.view(dt).copy()


def limbs_to_int(limbs):
    v = 0
    for i, x in enumerate(limbs):
        v |= int(x) << (64 * i)
    n = 64 * len(limbs)
    return v - (1 << n) if v >> (n - 1) else v


def run(seq, params=None, stages=3, trnas=None):
    """Run the oracle on one contig; returns a dict of numpy arrays / scalars (copies).
    trnas: None = no tRNA finder (functions.py:493-495), else the hit list [(start, stop)] add_trnas would hold (may be empty)."""
    if isinstance(seq, str):
        seq = seq.encode()
    params = params or make_params()
    r = Result()
    if trnas is None:
        lib().orc_run(seq, len(seq), C.byref(params), stages, C.byref(r))
    else:
        ts = np.ascontiguousarray([t[0] for t in trnas], np.int32)
        te = np.ascontiguousarray([t[1] for t in trnas], np.int32)
        lib().orc_run_trna(seq, len(seq), C.byref(params), stages, len(trnas), ts.ctypes.data_as(C.c_void_p), te.ctypes.data_as(C.c_void_p), C.byref(r))
    out = {"status": r.status, "L": r.L}
    if r.status == 0:
        L = r.L
        out["pstop"] = r.pstop
        out["background_rbs"] = np.array(r.background_rbs[:])
        out["training_rbs"] = np.array(r.training_rbs[:])
        out["pos_max"] = np.array(r.pos_max[:])
        out["pos_min"] = np.array(r.pos_min[:])
        out["gc_pos_freq"] = np.ctypeslib.as_array(r.gcpf, (r.n_gcpf * 3,)).reshape(-1, 3).copy()
        out["binF"] = np.ctypeslib.as_array(r.binF, (L,)).copy()
        out["binR"] = np.ctypeslib.as_array(r.binR, (L,)).copy()
        out["orf"] = _view(r.orf, r.n_orf, ORF_DT)
        out["other_end"] = np.ctypeslib.as_array(r.other_end, (L + 4,)).copy()
        if stages >= 2:
            nd = _view(r.node, r.n_node, NODE_DT)
            out["node_type"], out["node_frame"], out["node_pos"] = nd["type"], nd["frame"], nd["pos"]
            out["other_end_t"] = np.ctypeslib.as_array(r.other_end_t, (L + 4,)).copy()
            ed = _view(r.edge, r.n_edge, EDGE_DT)
            out["edge_src"], out["edge_dst"], out["edge_weight"] = ed["src"], ed["dst"], ed["w"]
            out["edge_wint_limbs"] = ed["wint"]
        if stages >= 3:
            out["path"] = np.array(r.path[: r.n_path], dtype=np.int32)
            out["wide"] = int(r.wide)  # the sums overflowed 256 bits: solved on 1280-bit integers (edge_wint_limbs are then meaningless)
            out["path_dist"] = limbs_to_int(r.dist_wide[:]) if r.wide else limbs_to_int(r.dist[:])
            out["bf_rounds"] = r.bf_rounds
            ge = _view(r.gene, r.n_gene, GENE_DT)
            out["gene_left"], out["gene_right"] = ge["left"], ge["right"]
            out["gene_strand"], out["gene_score"] = ge["strand"].astype(np.int8), ge["score"]
            out["gene_frame"] = ge["frame"].astype(np.int8)
    lib().orc_free(C.byref(r))
    return out


def score_rbs(seq):
    """functions.score_rbs on one window (<= 21 characters)."""
    if isinstance(seq, str):
        seq = seq.encode()
    return lib().orc_score_rbs(seq, len(seq))
