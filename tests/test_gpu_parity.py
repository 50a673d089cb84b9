"""Parity of the HIP path (through the C-ABI, phanotate_amd.Annotator) with the CPU oracle and the
golden vectors.  Bar: bit-exact for every integer output (position bytes, ORF table, nodes, edge
endpoints, path, gene coordinates, strands); fp64 edge weights / scores within 1e-9 of the oracle
(same formulas, libm vs device math) and 1e-6 of the reference's Decimal values (north_star)."""
import os

import numpy as np
import pytest

from conftest import golden_cases, golden_params, load_golden

pytestmark = pytest.mark.gpu

WTOL = 1e-9


@pytest.fixture(scope="module")
def pa():
    import phanotate_amd

    return phanotate_amd


def _mx(a, b, c):
    return np.where(a > b, np.where(a > c, 1, 3), np.where(b > c, 2, 3))


def _mn(a, b, c):
    return np.where(a > b, np.where(b > c, 3, 2), np.where(a > c, 3, 1))


def check_contig(ann, i, seq, o, genes, status):
    """All stage taps of contig i against the oracle result o."""
    if o["status"] < 0:
        assert status == o["status"]
        assert len(genes) == 0
        return
    assert status >= 0
    gl = ann.globals(i)
    pos = ann.positions(i)
    assert np.array_equal(pos["binF"][20:], o["binF"][20:])
    assert np.array_equal(pos["binR"], o["binR"])
    gcf = o["gc_pos_freq"][1:].astype(int)
    f = (_mx(gcf[:, 0], gcf[:, 1], gcf[:, 2]) - 1) * 3 + (_mn(gcf[:, 0], gcf[:, 1], gcf[:, 2]) - 1)
    r = (_mx(gcf[:, 2], gcf[:, 1], gcf[:, 0]) - 1) * 3 + (_mn(gcf[:, 2], gcf[:, 1], gcf[:, 0]) - 1)
    assert np.array_equal(pos["gcc"][: len(gcf)], (f | (r << 4)).astype(np.uint8))
    assert gl.pstop == o["pstop"]
    assert np.array_equal(np.array(gl.background_rbs[:]), o["background_rbs"])
    assert np.array_equal(np.array(gl.training_rbs[:]), o["training_rbs"])
    assert np.array_equal(np.array(gl.pos_max[:]), o["pos_max"])
    assert np.array_equal(np.array(gl.pos_min[:]), o["pos_min"])
    orf, oo = ann.orfs(i), o["orf"]
    assert len(orf) == len(oo)
    for k in ("start", "stop", "frame", "length", "rbs", "hist"):
        assert np.array_equal(orf[k], oo[k]), k
    assert np.array_equal(orf["startidx"], oo["first3_is_start"])
    assert np.array_equal(orf["pstop"], oo["pstop"])  # same IEEE operations in the same order
    assert np.array_equal(orf["S"], oo["S"])
    if len(oo):
        np.testing.assert_allclose(orf["weight"], oo["weight"], rtol=WTOL)
    nd = ann.nodes(i)
    assert np.array_equal(np.sort(nd["refidx"]), np.arange(len(nd)))
    perm = np.argsort(nd["refidx"], kind="stable")
    assert np.array_equal(nd["pos"][perm], o["node_pos"])
    assert np.array_equal(nd["type"][perm], o["node_type"])
    assert np.array_equal(nd["frame"][perm], o["node_frame"])
    cds = nd["type"] < 2
    assert np.array_equal(nd["other"][cds], o["other_end"][nd["pos"][cds]])
    ed = ann.edges(i)
    assert len(ed) == len(o["edge_src"])
    if len(ed):
        ref = nd["refidx"]
        gk = np.stack([ref[ed["src"]], ref[ed["dst"]]], 1)
        ok = np.stack([o["edge_src"], o["edge_dst"]], 1)
        gi, oi = np.lexsort((gk[:, 1], gk[:, 0])), np.lexsort((ok[:, 1], ok[:, 0]))
        assert np.array_equal(gk[gi], ok[oi])
        np.testing.assert_allclose(ed["w"][gi], o["edge_weight"][oi], rtol=WTOL)
    p, dist = ann.path(i)
    assert np.array_equal(nd["refidx"][p] if len(p) else p, o["path"])
    if len(o["path"]):
        assert abs(dist - o["path_dist"]) <= abs(o["path_dist"]) * WTOL
    assert np.array_equal(genes["left"], o["gene_left"])
    assert np.array_equal(genes["right"], o["gene_right"])
    assert np.array_equal(genes["strand"], o["gene_strand"].astype(np.int32))
    if len(genes):
        np.testing.assert_allclose(genes["score"], o["gene_score"], rtol=WTOL)


@pytest.mark.parametrize("case", golden_cases())
def test_golden_case(case, pa, oracle):
    g, name, seq = load_golden(case)
    kw = golden_params(g)
    ann = pa.Annotator(pa.make_params(**kw))
    (status, genes), = ann.annotate([seq])
    o = oracle.run(seq, oracle.make_params(**kw))
    if str(g["error"]):
        assert status < 0 and o["status"] < 0 and len(genes) == 0
    else:
        check_contig(ann, 0, seq, o, genes, status)
        # the reference's own numbers (Decimal + exact-integer solver), tests/golden/*.npz
        assert np.array_equal(genes["left"], g["gene_left"])
        assert np.array_equal(genes["right"], g["gene_right"])
        assert np.array_equal(genes["strand"], g["gene_strand"].astype(np.int32))
        if len(genes):
            np.testing.assert_allclose(genes["score"], g["gene_score"], rtol=1e-6)
    ann.close()


def test_readme_pins_on_gpu(pa):
    """README.md:45-54 of the reference: the only outputs it pins."""
    g, name, seq = load_golden("phiX174")
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    got = {(int(x["left"]), int(x["right"])): "%E" % x["score"] for x in genes}
    assert got[(100, 627)] == "-4.827981E+02"
    assert got[(687, 1622)] == "-4.857517E+06"
    assert got[(1686, 3227)] == "-3.785434E+10"
    assert got[(3224, 3484)] == "-3.779878E+02"
    ann.close()


def test_mixed_batch_equals_single_contig_runs(pa, oracle):
    """All default-parameter golden inputs in ONE batch (ragged lengths 5..169 kb, error contigs in the
    middle): per-contig results and every stage tap must equal the oracle; errors must not fail the batch."""
    items = []
    for c in golden_cases():
        g, name, seq = load_golden(c)
        if golden_params(g) == dict(start_codons=str(load_golden("phiX174")[0]["params_start"]), stop_codons="tag,tga,taa", minlen=90):
            items.append((c, seq))
    assert len(items) >= 20
    seqs = [s for _, s in items] + ["", "acg"]  # plus an empty and a 3-base contig
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    assert len(res) == len(seqs)
    for i, (c, seq) in enumerate(items):
        check_contig(ann, i, seq, oracle.run(seq), res[i][1], res[i][0])
    assert res[-2][0] == -3 and res[-1][0] == -3
    ann.close()


def test_empty_batch(pa):
    ann = pa.Annotator()
    assert ann.annotate([]) == []
    ann.close()


def test_synthetic_contigs_against_oracle(pa, oracle):
    """32 synthetic contigs (seeds that are not in tests/golden), ragged lengths."""
    rng = np.random.RandomState(11)
    seqs = [pa.synth_contig(1000 + s, int(rng.randint(300, 60000))) for s in range(32)]
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    for i, s in enumerate(seqs):
        o = oracle.run(s)
        st, genes = res[i]
        assert st == o["status"] == 0
        assert np.array_equal(genes["left"], o["gene_left"]) and np.array_equal(genes["right"], o["gene_right"])
        assert np.array_equal(genes["strand"], o["gene_strand"].astype(np.int32))
        np.testing.assert_allclose(genes["score"], o["gene_score"], rtol=WTOL)
    for i in (0, 7, 19):
        check_contig(ann, i, seqs[i], oracle.run(seqs[i]), res[i][1], res[i][0])
    ann.close()


def _check_gene_structure(seq, genes, minlen=90):
    """Size-independent domain properties of a gene list (what phanotate.py:71-76 can output)."""
    L = len(seq)
    starts, stops = (b"atg", b"gtg", b"ttg"), (b"tag", b"tga", b"taa")
    comp = bytes.maketrans(b"acgt", b"tgca")
    left, right = genes["left"], genes["right"]
    # path order: a connector edge reaches back less than 500 bp (functions.py:372)
    assert np.all(left[1:] > right[:-1] - 502)
    assert np.all(right > left) and np.all((right - left + 1) % 3 == 0) and np.all(right - left + 1 >= minlen)
    assert np.all(np.isfinite(genes["score"])) and np.all(genes["score"] < 0)
    for g in genes:
        s = seq[g["left"] - 1 : g["right"]]
        if g["strand"] < 0:
            s = s.translate(comp)[::-1]
        cod = [s[k : k + 3] for k in range(0, len(s), 3)]
        at_edge_5 = (g["left"] <= 3) if g["strand"] > 0 else (g["right"] >= L - 2)
        at_edge_3 = (g["right"] >= L - 2) if g["strand"] > 0 else (g["left"] <= 3)
        assert cod[0] in starts or at_edge_5
        assert cod[-1] in stops or at_edge_3
        assert not any(c in stops for c in cod[:-1]), "in-frame stop inside a called gene"


def test_full_size_batch_properties(pa, oracle):
    """BASELINE config 4 at full size (1000 x 50 kb): determinism, batch-order independence,
    structural validity of every gene, and a spot check of 12 contigs against the oracle."""
    n = 1000
    seqs = [pa.synth_contig(s, 50000) for s in range(n)]
    ann = pa.Annotator()
    r1 = ann.annotate(seqs)
    assert all(st == 0 for st, _ in r1)
    ann.run()
    r2 = ann.download()  # idempotent: a second pass over the resident batch
    rng = np.random.RandomState(5)
    perm = rng.permutation(n)
    r3 = ann.annotate([seqs[k] for k in perm])
    for i in range(n):
        assert r1[i][1].tobytes() == r2[i][1].tobytes()
    for j, k in enumerate(perm):
        assert r3[j][1].tobytes() == r1[k][1].tobytes(), "result of a contig depends on its neighbours in the batch"
    for i in range(0, n, 7):
        _check_gene_structure(seqs[i], r1[i][1])
    for i in rng.choice(n, 12, replace=False):
        o = oracle.run(seqs[i])
        assert np.array_equal(r1[i][1]["left"], o["gene_left"]) and np.array_equal(r1[i][1]["right"], o["gene_right"])
        np.testing.assert_allclose(r1[i][1]["score"], o["gene_score"], rtol=WTOL)
    assert sum(len(g) for _, g in r1) > 40 * n
    ann.close()


def test_attach_device_resident_input(pa):
    """phx_attach: the concatenated ASCII already lives in HBM (a torch tensor), offsets on the host."""
    import torch

    seqs = [pa.synth_contig(50 + s, 7000 + 911 * s) for s in range(5)]
    ann = pa.Annotator(stream=torch.cuda.current_stream().cuda_stream)
    want = ann.annotate(seqs)
    buf = torch.frombuffer(bytearray(b"".join(seqs)), dtype=torch.uint8).cuda()
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
    ann.attach(buf.data_ptr(), offs)
    ann.run()
    got = ann.download()
    for (s1, g1), (s2, g2) in zip(want, got):
        assert s1 == s2 and g1.tobytes() == g2.tobytes()
    ann.close()


def _bellman_ford(V, src, dst, w, s, t):
    dist = [None] * V
    par = [-1] * V
    dist[s] = 0
    for _ in range(V):
        ch = False
        for u, v, x in zip(src, dst, w):
            if dist[u] is not None and (dist[v] is None or dist[u] + x < dist[v]):
                dist[v] = dist[u] + x
                par[v] = u
                ch = True
        if not ch:
            break
    return dist


@pytest.mark.parametrize("bits,nl", [(40, 2), (100, 2), (200, 4), (450, 8), (1000, 17)])
def test_solver_alone_exact_integers(pa, bits, nl):
    """The fastpathz boundary (phanotate.py:56-64): arbitrary-precision integer weights, cyclic graphs."""
    import random

    rnd = random.Random(bits)
    ann = pa.Annotator()
    for trial in range(6):
        V = rnd.randint(5, 700)
        E = rnd.randint(V, 6 * V)
        src = [rnd.randrange(V) for _ in range(E)]
        dst = [rnd.randrange(V) for _ in range(E)]
        # potentials make every cycle non-negative while most edges are negative-looking
        pot = [rnd.randrange(-(1 << bits), 1 << bits) for _ in range(V)]
        w = [rnd.randrange(0, 1 << (bits - 4)) + pot[u] - pot[v] for u, v in zip(src, dst)]
        s, t = 0, V - 1
        if trial == 5:  # unreachable target
            keep = [k for k in range(E) if dst[k] != t]
            src, dst, w = [src[k] for k in keep], [dst[k] for k in keep], [w[k] for k in keep]
        ref = _bellman_ford(V, src, dst, w, s, t)
        path, dist = ann.solve(V, src, dst, w, s, t, n_limbs=nl)
        if ref[t] is None:
            assert path == [] and dist is None
            continue
        assert dist == ref[t]
        assert path[0] == s and path[-1] == t
        tot = 0
        for a, b in zip(path[:-1], path[1:]):  # the returned path must realise the distance
            cands = [w[k] for k in range(len(w)) if src[k] == a and dst[k] == b]
            assert cands
            tot += min(cands)
        assert tot == dist
    ann.close()


def test_cli_tabular_and_dump(pa, tmp_path):
    """phanotate.py end to end: multi-contig FASTA in, the reference's tabular text out; and -d/--dump in the
    reference's Graph.iteredges order (weights compared numerically: Decimal text cannot be reproduced in fp64)."""
    import gzip
    import re
    import subprocess
    import sys

    from conftest import GOLDEN, ROOT

    cases = ["phiX174", "edge_L200", "synth6k_100", "edge_bridge"]
    fa = tmp_path / "in.fasta"
    want = ""
    with open(fa, "w") as f:
        for c in cases:
            g, name, seq = load_golden(c)
            f.write(">%s some description\n%s\n" % (name, seq))
            want += str(g["tabular"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "phanotate.py"), str(fa)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == want
    for c in ("phiX174", "edge_bridge"):
        g, name, seq = load_golden(c)
        one = tmp_path / (c + ".fa")
        one.write_text(">%s\n%s\n" % (name, seq))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "phanotate.py"), "-d", str(one)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().split("\n")
        assert len(lines) == len(g["edge_src"])
        tcode = {"start": 0, "stop": 1, "source": 2, "target": 3}
        pat = re.compile(r"Node\('(\w+)','(\w+)',(-?\d+),(\d+)\)")
        for k, line in enumerate(lines):
            s, d, w = line.split("\t")
            ms, md = pat.fullmatch(s), pat.fullmatch(d)
            si, di = int(g["edge_src"][k]), int(g["edge_dst"][k])
            assert (tcode[ms.group(2)], int(ms.group(3)), int(ms.group(4))) == (int(g["node_type"][si]), int(g["node_frame"][si]), int(g["node_pos"][si])), (c, k)
            assert (tcode[md.group(2)], int(md.group(3)), int(md.group(4))) == (int(g["node_type"][di]), int(g["node_frame"][di]), int(g["node_pos"][di])), (c, k)
            assert abs(float(w) / (float(g["edge_weight"][k]) * 1000) - 1) < 1e-9


class _Locus:
    def __init__(self, seq, start_codons=None, stop_codons=None, minlen=90):
        self._seq = seq
        self.start_codons = start_codons or {"atg": 1.0, "gtg": 0.10 / 0.85, "ttg": 0.05 / 0.85}
        self.stop_codons = stop_codons or ["tag", "tga", "taa"]
        self.min_orf_len = minlen

    def seq(self):
        return self._seq


def test_reference_style_driver_loop(pa):
    """The body of phanotate.py:40-76 written against the mirror modules (functions.get_orfs / get_graph and
    the fastpathz shim): same call sequence, and the same genes as the golden vectors."""
    from phanotate_amd import fastpathz as fz
    from phanotate_amd import functions

    for case in ("phiX174", "synth6k_101", "edge_L200"):
        g, name, seq = load_golden(case)
        locus = _Locus(seq)
        orfs = functions.get_orfs(locus)
        graph = functions.get_graph(orfs)
        assert [o.start for o in orfs.iter_orfs()] == list(g["orf_start"])
        assert [n.position for n in graph.iternodes()] == list(g["node_pos"])
        fz.empty_graph()
        for e in graph.iteredges():
            fz.add_edge(str(e))
        source = "Node('source','source',0,0)"
        target = "Node('target','target',0," + str(len(seq) + 1) + ")"
        shortest_path = fz.get_path(source=source, target=target)[1:] if len(graph) > 2 else []
        got = []
        it = iter(shortest_path)
        for s, t in zip(it, it):
            left, right = eval(s, {"Node": functions.Node}), eval(t, {"Node": functions.Node})
            got.append((left.position, right.position + 2, graph.weight(functions.Edge(left, right, 0))))
        assert [x[0] for x in got] == list(g["gene_left"]) and [x[1] for x in got] == list(g["gene_right"])
        np.testing.assert_allclose([x[2] for x in got], g["gene_score"], rtol=1e-6)
    with pytest.raises(KeyError):
        functions.get_orfs(_Locus("acgtxacgt" * 50))
