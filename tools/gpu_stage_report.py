#!/usr/bin/env python3
"""Developer aid: run every golden case through libphx on the GPU and print, stage by stage, how it
compares with the CPU oracle.  (Uses oracle/ as the checker, like the tests do.)"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_cases, golden_params, load_golden  # noqa: E402

import phanotate_amd as pa  # noqa: E402
from oracle import oracle  # noqa: E402


def cmp(name, a, b, tol=None):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        print("    %-14s SHAPE gpu %s oracle %s" % (name, a.shape, b.shape)); return False
    if a.size == 0:
        print("    %-14s ok (0)" % name); return True
    if tol is None:
        bad = np.flatnonzero((a != b).reshape(len(a), -1).any(axis=1)) if a.ndim > 1 else np.flatnonzero(a != b)
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        rel = np.where(a == b, 0, rel)
        bad = np.flatnonzero(rel > tol)
    if len(bad):
        i = bad[0]
        print("    %-14s MISMATCH %d/%d first@%d gpu=%s oracle=%s" % (name, len(bad), len(a), i, a[i], b[i])); return False
    print("    %-14s ok (%d)" % (name, len(a))); return True


def report(case_filter=None):
    cases = [c for c in golden_cases() if not case_filter or any(f in c for f in case_filter)]
    groups = {}
    for c in cases:
        g, name, seq = load_golden(c)
        key = (str(g["params_start"]), str(g["params_stop"]), int(g["params_minlen"]))
        groups.setdefault(key, []).append((c, g, seq))
    allok = True
    for key, items in groups.items():
        ann = pa.Annotator(pa.make_params(*key))
        t0 = time.time()
        res = ann.annotate([s for _, _, s in items])
        print("params %s: %d contigs in %.3fs" % (key, len(items), time.time() - t0))
        op = oracle.make_params(*key)
        for i, (c, g, seq) in enumerate(items):
            o = oracle.run(seq, op)
            st, genes = res[i]
            gl = ann.globals(i)
            print("  %s: L=%d status gpu=%d oracle=%d  n_orf=%d n_node=%d n_edge=%d n_bridge=%d limbs=%d sweeps=%d iters=%d" %
                  (c, len(seq), st, o["status"], gl.n_orf, gl.n_node, gl.n_edge, gl.n_bridge, gl.n_limbs, gl.sssp_sweeps, gl.sssp_iters))
            try:
                ok = True
                if o["status"] < 0:
                    ok &= st == o["status"] or (st < 0)
                    if st >= 0: print("    STATUS mismatch")
                    allok &= ok
                    continue
                if st < 0:
                    print("    STATUS gpu error"); allok = False; continue
                L = len(seq)
                pos = ann.positions(i)
                ok &= cmp("binF", pos["binF"][20:], o["binF"][20:])
                ok &= cmp("binR", pos["binR"], o["binR"])
                gcf = o["gc_pos_freq"]
                n = len(gcf) - 1
                a0, a1, a2 = gcf[1:, 0].astype(int), gcf[1:, 1].astype(int), gcf[1:, 2].astype(int)
                def mx(a, b, c): return np.where(a > b, np.where(a > c, 1, 3), np.where(b > c, 2, 3))
                def mn(a, b, c): return np.where(a > b, np.where(b > c, 3, 2), np.where(a > c, 3, 1))
                f = (mx(a0, a1, a2) - 1) * 3 + (mn(a0, a1, a2) - 1)
                r = (mx(a2, a1, a0) - 1) * 3 + (mn(a2, a1, a0) - 1)
                ok &= cmp("gcc", pos["gcc"][:n], (f | (r << 4)).astype(np.uint8))
                ok &= cmp("pstop", [gl.pstop], [o["pstop"]], 1e-15)
                ok &= cmp("background", list(gl.background_rbs), o["background_rbs"], 1e-15)
                ok &= cmp("training", list(gl.training_rbs), o["training_rbs"], 1e-15)
                ok &= cmp("pos_max", list(gl.pos_max), o["pos_max"], 1e-15)
                ok &= cmp("pos_min", list(gl.pos_min), o["pos_min"], 1e-15)
                orf = ann.orfs(i); oo = o["orf"]
                for k in ("start", "stop", "frame", "length", "rbs", "hist"):
                    ok &= cmp("orf." + k, orf[k], oo[k])
                ok &= cmp("orf.startidx", orf["startidx"], oo["first3_is_start"])
                ok &= cmp("orf.pstop", orf["pstop"], oo["pstop"], 1e-15)
                ok &= cmp("orf.S", orf["S"], oo["S"], 1e-15)
                ok &= cmp("orf.weight", orf["weight"], oo["weight"], 1e-9)
                nd = ann.nodes(i)
                perm = np.argsort(nd["refidx"], kind="stable")  # device order -> reference order
                ok &= cmp("node.refidx", np.sort(nd["refidx"]), np.arange(len(nd)))
                ok &= cmp("node.pos", nd["pos"][perm], o["node_pos"])
                ok &= cmp("node.type", nd["type"][perm], o["node_type"])
                ok &= cmp("node.frame", nd["frame"][perm], o["node_frame"])
                cds = nd["type"] < 2
                ok &= cmp("node.other", nd["other"][cds], o["other_end"][nd["pos"][cds]])
                ed = ann.edges(i)
                ref = nd["refidx"]
                gk = np.stack([ref[ed["src"]], ref[ed["dst"]]], 1) if len(ed) else np.zeros((0, 2), int)
                ok_ = np.stack([o["edge_src"], o["edge_dst"]], 1) if len(o["edge_src"]) else np.zeros((0, 2), int)
                gi = np.lexsort((gk[:, 1], gk[:, 0])); oi = np.lexsort((ok_[:, 1], ok_[:, 0]))
                ok &= cmp("edge.pairs", gk[gi], ok_[oi])
                if len(gk) == len(ok_):
                    ok &= cmp("edge.w", ed["w"][gi], o["edge_weight"][oi], 1e-9)
                p, dist = ann.path(i)
                ok &= cmp("path", ref[p] if len(p) else p, o["path"])
                if len(o["path"]):
                    rel = abs(dist - o["path_dist"]) / max(1, abs(o["path_dist"]))
                    print("    dist rel err %.2e (gpu %d)" % (rel, dist))
                ok &= cmp("gene.left", genes["left"], o["gene_left"])
                ok &= cmp("gene.right", genes["right"], o["gene_right"])
                ok &= cmp("gene.strand", genes["strand"], o["gene_strand"].astype(np.int32))
                ok &= cmp("gene.score", genes["score"], o["gene_score"], 1e-9)
                ok &= cmp("gene.left(gold)", genes["left"], g["gene_left"])
                ok &= cmp("gene.score(gold)", genes["score"], g["gene_score"], 1e-6)
                allok &= bool(ok)
            except Exception:
                traceback.print_exc(); allok = False
        ann.close()
    print("ALL OK" if allok else "SOME MISMATCH")
    return allok


if __name__ == "__main__":
    sys.exit(0 if report(sys.argv[1:]) else 1)
