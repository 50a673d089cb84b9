"""Parity of the HIP path (through the C-ABI, phanotate_amd.Annotator) with the CPU oracle and the
golden vectors.  Bar: bit-exact for every integer output (position bytes, ORF table, nodes, edge
endpoints, path, gene coordinates, strands); fp64 edge weights / scores within 1e-9 of the oracle
(same formulas, libm vs device math) and 1e-6 of the reference's Decimal values (north_star)."""
import os

import numpy as np
import pytest

from conftest import exact_dist_from_device_edges, golden_cases, golden_params, golden_trnas, inorder_bellman_ford, load_golden

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


def codon_classes(seq, starts=("atg", "gtg", "ttg"), stops=("tag", "tga", "taa")):
    """functions.py:198-215 per position: 0 none, 1 codon in start_codons, 2 rev_comp(codon) in start_codons, 3 codon in
    stop_codons, 4 rev_comp(codon) in stop_codons (first match of the elif chain); codons with a letter outside acgt match nothing."""
    s = (seq.decode() if isinstance(seq, (bytes, bytearray)) else seq).lower()
    comp = {"a": "t", "c": "g", "g": "c", "t": "a"}
    out = np.zeros(len(s), np.uint8)
    for p in range(len(s) - 2):
        c = s[p : p + 3]
        if any(x not in comp for x in c):
            continue
        rc = comp[c[2]] + comp[c[1]] + comp[c[0]]
        out[p] = 1 if c in starts else 2 if rc in starts else 3 if c in stops else 4 if rc in stops else 0
    return out


def check_exact_distances(ann, i):
    """Every node's distance, as the device holds it, equals an exact python-int Bellman-Ford over the device's own edges."""
    want = exact_dist_from_device_edges(ann, i)
    got = ann.dist(i)
    assert len(got) == len(want)
    bad = [v for v in range(len(want)) if got[v] != want[v]]
    assert not bad, "contig %d: %d of %d distances differ, first at node %d: %r vs %r" % (i, len(bad), len(want), bad[0], got[bad[0]], want[bad[0]])


def check_contig(ann, i, seq, o, genes, status, params=None, fp64_decides=True):
    """All stage taps of contig i against the oracle result o.  fp64_decides=False: a contig whose path the fp64-level oracle cannot know
    (the neartie fixtures): everything but path and genes."""
    if o["status"] == -7 and status == 0:
        # an ORF weight beyond fp64: the oracle (fp64 weights) gives up, libphx solves the contig on the host in the reference's own
        # arithmetic (phx_exact.inc) — checked against the reference's fixture in test_contig_beyond_the_device_integers_is_solved_on_the_host
        assert ann.certified()[i] == 2 and len(genes) > 0
        return
    if o["status"] < 0:
        assert status == o["status"]
        assert len(genes) == 0
        return
    assert status >= 0
    gl = ann.globals(i)
    pos = ann.positions(i)
    kw = params or {}
    starts = tuple(x.split(":")[0] for x in kw["start_codons"].split(",")) if "start_codons" in kw else ("atg", "gtg", "ttg")
    stops = tuple(kw["stop_codons"].split(",")) if "stop_codons" in kw else ("tag", "tga", "taa")
    assert np.array_equal(pos["cls"] & 7, codon_classes(seq, starts, stops))
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
    trn = (nd["type"] < 2) & (np.abs(nd["frame"]) == 4)
    cds = (nd["type"] < 2) & ~trn
    assert np.array_equal(nd["other"][cds], o["other_end"][nd["pos"][cds]])
    assert np.array_equal(nd["other"][trn], o["other_end_t"][nd["pos"][trn]])  # other_end['t' + str(pos)], functions.py:502-508
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
    if not fp64_decides:
        check_exact_distances(ann, i)
        return
    assert np.array_equal(nd["refidx"][p] if len(p) else p, o["path"])
    if len(o["path"]):
        assert abs(dist - o["path_dist"]) <= abs(o["path_dist"]) * WTOL  # the oracle's weights come from the host libm
        if gl.n_node <= 8000:
            check_exact_distances(ann, i)  # ... against the device's own weights the sums are exact
    assert np.array_equal(genes["left"], o["gene_left"])
    assert np.array_equal(genes["right"], o["gene_right"])
    assert np.array_equal(genes["strand"], o["gene_strand"].astype(np.int32))
    assert np.array_equal(genes["frame"], o["gene_frame"].astype(np.int32))
    if len(genes):
        np.testing.assert_allclose(genes["score"], o["gene_score"], rtol=WTOL)


@pytest.mark.parametrize("case", golden_cases())
def test_golden_case(case, pa, oracle):
    g, name, seq = load_golden(case)
    kw = golden_params(g)
    tr = golden_trnas(g)
    ann = pa.Annotator(pa.make_params(**kw))
    (status, genes), = ann.annotate([seq], trnas=None if tr is None else [tr])
    o = oracle.run(seq, oracle.make_params(**kw), trnas=tr)
    if str(g["error"]):
        assert status < 0 and o["status"] < 0 and len(genes) == 0
        if str(g["error"]) == "ValueError":
            assert status == -6 and o["status"] == -6  # "parallel edges are forbidden", graphs.py:74 (PHX_S_PARALLEL)
    elif case == "edge_huge":
        return  # path sums beyond the device's integers: test_contig_beyond_the_device_integers_is_solved_on_the_host
    else:
        if case == "edge_wide":  # 431-bit path sums: the device's 512-bit class, the oracle's 1280-bit solver
            assert o["status"] == 0 and o["wide"] == 1 and status == 0 and ann.globals(0).n_limbs == 8
            nd = ann.nodes(0)
            assert np.array_equal(nd["refidx"][ann.path(0)[0]], g["path"])
            want_d = int(str(g["path_dist"]))  # the reference's own integers: the device's differ in the low digits of the 1e126 edge
            assert abs(ann.path(0)[1] - want_d) * 10 ** 12 < abs(want_d)
        check_contig(ann, 0, seq, o, genes, status, kw, fp64_decides=not case.startswith("neartie"))
        # the reference's own numbers (Decimal + exact-integer solver), tests/golden/*.npz
        assert np.array_equal(genes["left"], g["gene_left"])
        assert np.array_equal(genes["right"], g["gene_right"])
        assert np.array_equal(genes["strand"], g["gene_strand"].astype(np.int32))
        if "gene_frame" in g:
            assert np.array_equal(genes["frame"], g["gene_frame"].astype(np.int32))  # +-4: a tRNA feature
        if len(genes):
            np.testing.assert_allclose(genes["score"], g["gene_score"], rtol=1e-6)
        if tr is not None:
            from phanotate_amd.cli import format_tabular

            assert format_tabular([name], np.array([status], np.int32), np.array([0, len(genes)], np.int64), genes).decode() == str(g["tabular"])  # CDS features only
        # the same contig once more: the first run of a context sizes its buffers between the staged kernels, the second is a lone
        # contig's steady state — the front end as ONE launch (k_front, phx_front.inc).  Every tap again.
        ann.run()
        st2, _, fl2 = ann.download_flat()
        assert ann.front_runs() == (1 if len(seq) <= 40 << 10 else 0) and int(st2[0]) == status and fl2.tobytes() == genes.tobytes()
        check_contig(ann, 0, seq, o, fl2, int(st2[0]), kw, fp64_decides=not case.startswith("neartie"))
    ann.close()


def test_contig_beyond_the_device_integers_is_solved_on_the_host(pa, oracle):
    """tests/golden/edge_huge (generated through the reference): one ORF of 21 000 sense codons, weight -1.13e331 — more than a double
    holds, path sums of 1110 bits, beyond the device's widest integer class (1088).  The reference's Decimal and its solver's integers
    have no limit (CHANGELOG.md:11-13); libphx used to refuse such a contig (PHX_S_OVERFLOW).  Now the device builds its graph, and
    phx_download* solve it on the host in the reference's own arithmetic (phx_exact.inc: every flagged edge replayed in Decimal, the
    in-place Bellman-Ford on integers as wide as the weights need): the reference's genes, certified == 2.  With exactness off the
    status is PHX_S_OVERFLOW as before; the other contigs of the batch do not notice."""
    g, name, seq = load_golden("edge_huge")
    _, _, lam = load_golden("NC_001416.1")
    gl_, _, _ = load_golden("NC_001416.1")
    ann = pa.Annotator()
    res = ann.annotate([lam, seq, pa.synth_contig(77, 9000)])
    (s0, g0), (status, genes), (s2, g2) = res
    assert s0 == 0 and s2 == 0 and np.array_equal(g0["left"], gl_["gene_left"])
    assert status == 0
    assert np.array_equal(genes["left"], g["gene_left"]) and np.array_equal(genes["right"], g["gene_right"])
    assert np.array_equal(genes["strand"], g["gene_strand"].astype(np.int32))
    fin = np.isfinite(g["gene_score"])
    assert (~fin).sum() == 1 and np.array_equal(np.isfinite(genes["score"]), fin)  # '%E' % float(Decimal('-1.13E+331')) is -INF in the reference too
    np.testing.assert_allclose(genes["score"][fin], g["gene_score"][fin], rtol=1e-6)
    cert = ann.certified()
    assert list(cert) == [1, 2, 1]
    gl = ann.globals(1)
    assert gl.n_limbs == 0 and gl.sssp_kernel == 4
    # the structure the host solved on is the device's: ORF table and graph against the fixture
    orf = ann.orfs(1)
    for k, gk in (("start", "orf_start"), ("stop", "orf_stop"), ("frame", "orf_frame"), ("length", "orf_length"), ("rbs", "orf_rbs")):
        assert np.array_equal(orf[k], g[gk]), k
    nd = ann.nodes(1)
    perm = np.argsort(nd["refidx"], kind="stable")
    assert np.array_equal(nd["pos"][perm], g["node_pos"])
    assert len(ann.edges(1)) == len(g["edge_src"])
    # tabular text and --dump, byte for byte
    from phanotate_amd.cli import format_tabular
    import hashlib

    assert format_tabular([name], np.array([status], np.int32), np.array([0, len(genes)], np.int64), genes).decode() == str(g["tabular"])
    assert hashlib.md5(ann.dump_text(1)).hexdigest() == str(g["dump_md5"])
    # exactness off: no device kernel has this contig's genes
    st, offs, fl = ann.download_flat(exact=False)
    assert list(st) == [0, -7, 0] and offs[2] - offs[1] == 0
    st, offs, fl = ann.download_flat()
    assert list(st) == [0, 0, 0] and offs[2] - offs[1] == len(g["gene_left"])
    ann.close()
    bare = pa.Annotator(flags=("no_exact",))
    (sb, gb), = bare.annotate([seq])
    assert sb == -7 and len(gb) == 0
    bare.close()


def test_small_batches_fused_front_end_equals_the_staged_kernels(pa):
    """Batches of up to 4 contigs run ORF count ... edge fill as one kernel with grid barriers (k_front) once the context's buffers are
    sized; PHX_CREATE_NO_FUSE keeps the staged kernels.  Same records byte for byte — genes, ORF tables, nodes, edges — for 1, 2, 3 and 4
    contigs of mixed length, with a bad-letter contig, a too-short one and tRNA hits in the batch; 5 and 33 contigs stay staged."""
    rng = np.random.RandomState(8)
    for n in (1, 2, 3, 4, 5, 33):
        seqs = [pa.synth_contig(3000 + 100 * n + i, int(rng.choice([600, 3000, 6000, 9000] if n <= 4 else [600, 3000, 20000, 40000]))) for i in range(n)]  # (<= 40 kb in all: beyond, the staged kernels are used)
        if n >= 4:
            seqs[3] = b"acgtnnacgx" * 50
            seqs[1] = b"acgta"
        hits = [[(100, 180), (400, 320)] if (i % 3 == 0 and len(s) > 1000) else [] for i, s in enumerate(seqs)]
        got = {}
        for mode in ("fused", "staged"):
            a = pa.Annotator(flags=() if mode == "fused" else ("no_fuse",))
            a.upload(seqs)
            a.set_trnas(hits)
            a.run()
            a.run()
            a.run()  # (the third run on a layout replays the captured graph)
            assert a.front_runs() == (2 if (mode == "fused" and n <= 4) else 0), (n, mode, a.front_runs())  # (every run after the sizing one, the graph's replay included)
            flat = a.download_flat()
            taps = []
            for i in sorted(set([0, n // 2, n - 1] + ([3, 1] if n >= 4 else []))):
                gl = a.globals(i)
                taps.append((gl.n_orf, gl.n_node, gl.n_edge, a.orfs(i).tobytes(), a.nodes(i).tobytes(), a.edges(i).tobytes()))
            got[mode] = (flat, taps)
            a.close()
        for x, y in zip(got["fused"][0], got["staged"][0]):
            assert x.tobytes() == y.tobytes(), n
        assert got["fused"][1] == got["staged"][1], n


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
        if golden_trnas(g) is not None:
            continue
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
    # the guarantee holds on the zero-copy path too: contigs the certificate leaves open (most of them with the inflated bounds of
    # cert_tight) are solved again on the host from bases the library fetches back from the caller's buffer (this path used
    # to skip the re-solve silently)
    tight = pa.Annotator(flags=("cert_tight",))
    tight.attach(buf.data_ptr(), offs)
    tight.run()
    st, o2, g2 = tight.download_flat()
    assert len(tight.resolved) >= 1 and (tight.certified() != 0).all()
    for i, (s1, g1) in enumerate(want):
        g = g2[o2[i] : o2[i + 1]]
        assert st[i] == s1 and all(np.array_equal(g[f], g1[f]) for f in ("left", "right", "strand", "frame"))
    tight.close()


def test_attach_more_contigs_than_a_grid_has_rows(pa):
    """70 000 tiny contigs through phx_attach: k_pack_planes packs the caller's letters with the contig index in gridDim.x (it
    was gridDim.y, which ends at 65 535).  Same records as the upload path, which packs on the host."""
    import torch

    kinds = [pa.synth_contig(7000 + s, 100 + (s % 7) * 30) for s in range(350)]
    seqs = [kinds[i % 350] for i in range(70000)]
    ann = pa.Annotator()
    want = ann.annotate_flat(seqs)
    buf = torch.frombuffer(bytearray(b"".join(seqs)), dtype=torch.uint8).cuda()
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
    ann.attach(buf.data_ptr(), offs)
    ann.run()
    got = ann.download_flat()
    assert all(a.tobytes() == b.tobytes() for a, b in zip(got, want))
    assert int(want[1][-1]) > 0  # genes were called
    ann.close()


def test_large_batch_side_streams_equal_small_batches(pa):
    """Batches of 600 contigs or more write the ORF edges' rows by k_edges_orf (a thread per ORF, on a side stream beside the neighbour scans of the open
    nodes) and run k_score beside k_node_attr; smaller batches keep everything in k_edges<true> on one stream.  720 contigs with tRNA hits on every fifth,
    a bad-letter and a too-short contig among them: genes, and the tapped ORF / node / edge tables of a sample, equal to the same contigs in batches of 90."""
    rng = np.random.RandomState(66)
    seqs = [pa.synth_contig(66000 + i, int(rng.choice([900, 3000, 8000, 15000]))) for i in range(720)]
    seqs[17] = b"acgtnnacgx" * 60
    seqs[401] = b"acgta"
    hits = [[(100, 180), (400, 320)] if (i % 5 == 0 and len(s) > 1000) else [] for i, s in enumerate(seqs)]
    sample = [0, 5, 17, 100, 401, 640, 719]

    def run(lo, hi):
        a = pa.Annotator()
        a.upload(seqs[lo:hi])
        a.set_trnas(hits[lo:hi])
        a.run()
        a.run()  # (steady state)
        st, offs, genes = a.download_flat()
        taps = {}
        for i in sample:
            if lo <= i < hi:
                gl = a.globals(i - lo)
                taps[i] = (gl.n_orf, gl.n_node, gl.n_edge, a.orfs(i - lo).tobytes(), a.nodes(i - lo).tobytes(), a.edges(i - lo).tobytes())
        per = [(int(st[k]), genes[offs[k]:offs[k + 1]].tobytes()) for k in range(hi - lo)]
        a.close()
        return per, taps

    big, big_taps = run(0, 720)
    small, small_taps = [], {}
    for lo in range(0, 720, 90):
        p, t = run(lo, lo + 90)
        small += p
        small_taps.update(t)
    assert big == small
    assert big_taps == small_taps
    assert sum(1 for s_, g in big if s_ == 0 and g) > 600


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
    """phanotate.py end to end: multi-contig FASTA in, the reference's tabular text out; and -d/--dump: the reference's
    edge text (Graph.iteredges order, Decimal weights), checked against the fixtures' md5 of that text."""
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
        import hashlib

        assert hashlib.md5(r.stdout.encode()).hexdigest() == str(g["dump_md5"]), c


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


def _py_bellman_ford_genes(o):
    """Exact reference solve (python ints) over the oracle's stage-2 graph: for contigs whose path sums do not
    fit the oracle's 256-bit integers."""
    import math

    src, dst, w = o["edge_src"], o["edge_dst"], o["edge_weight"]
    wi = [int(math.trunc(float(x) * 1000.0)) for x in w]  # python float*1000 == C double*1000; int() is exact
    V = len(o["node_pos"])
    dist = [None] * V
    par = [-1] * V
    s, t = V - 2, V - 1
    dist[s] = 0
    for _ in range(V):
        ch = False
        for k in range(len(src)):
            u, v = int(src[k]), int(dst[k])
            if dist[u] is not None and (dist[v] is None or dist[u] + wi[k] < dist[v]):
                dist[v] = dist[u] + wi[k]
                par[v] = u
                ch = True
        if not ch:
            break
    if dist[t] is None:
        return None, []
    path = [t]
    while path[-1] != s:
        path.append(par[path[-1]])
    path.reverse()
    sp = path[1:]
    genes = [(int(o["node_pos"][a]), int(o["node_pos"][b]) + 2) for a, b in zip(sp[0::2], sp[1::2])]
    return dist[t], genes


@pytest.mark.parametrize("ncodons,min_limbs", [(5500, 8), (12000, 17)])
def test_very_long_orf_wide_integers_and_far_edges(pa, oracle, ncodons, min_limbs):
    """A 24-48 kb stop-free reading frame rich in start codons: the ORF weights need 512 / 1088-bit integers, the stop
    node has more in-edges (one per start) than the LDS tile holds (untiled path), and most of its start nodes lie
    far outside the LDS distance ring."""
    rng = np.random.RandomState(42)
    sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
    w = np.array([12.0 if c in ("atg", "gtg", "ttg") else 1.0 for c in sense])
    body = "".join(rng.choice(sense, ncodons, p=w / w.sum()))
    seq = pa.synth_contig(900, 4000).decode() + "atg" + body + "taa" + pa.synth_contig(901, 4000).decode()
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    gl = ann.globals(0)
    assert status == 0
    assert gl.n_limbs >= min_limbs
    o = oracle.run(seq, stages=2)  # the oracle's own 256-bit solver would overflow: solve its graph with python ints
    assert o["status"] == 0
    assert np.bincount(o["edge_dst"]).max() > 1024  # more in-edges than SW_ECAP
    dist, want = _py_bellman_ford_genes(o)
    assert [(int(g["left"]), int(g["right"])) for g in genes] == want
    p, d = ann.path(0)
    assert abs(d - dist) <= abs(dist) * 1e-12  # weights agree to ~1e-15, so do the exact sums
    check_exact_distances(ann, 0)  # k_sssp_lds<8> / <17>: bit for bit against the device's own weights
    ann.close()


def test_unreachable_target(pa, oracle):
    """ORFs only in the middle of the contig (more than 2000 bp from both ends): the source has no out-edge
    (functions.py:444-452), there is no path and no gene is reported."""
    dense_stops = "".join("tagctaactgattaa"[i % 15] for i in range(2700))
    seq = dense_stops + pa.synth_contig(77, 1500).decode() + dense_stops
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    o = oracle.run(seq)
    assert o["status"] == 0 and len(o["gene_left"]) == 0 and len(o["path"]) == 0
    assert status == 1 and len(genes) == 0  # PHX_S_NOPATH
    ann.close()


def test_fuzz_small_random_contigs(pa, oracle):
    """400 random contigs in one batch: every length class around the tile / bitmap-word boundaries, IUPAC codes,
    upper case, occasional illegal letters; every status and gene list must equal the oracle's, and all stage taps
    for a sample."""
    rng = np.random.RandomState(2024)
    lens = [6, 7, 8, 9, 20, 21, 22, 63, 64, 65, 89, 90, 91, 92, 93, 191, 192, 193, 575, 576, 577, 1535, 1536, 1537, 1538, 3071, 3072, 3073, 4608]
    seqs = []
    for i in range(400):
        L = lens[i] if i < len(lens) else int(rng.randint(6, 5000))
        gc = rng.uniform(0.25, 0.7)
        s = rng.choice(list("acgt"), L, p=[(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
        if i % 5 == 0:  # sprinkle ambiguity codes
            for p in rng.choice(L, max(1, L // 50), replace=False):
                s[p] = "nryswkmbvdh"[rng.randint(11)]
        if i % 37 == 36:
            s[rng.randint(L)] = "x"
        s = "".join(s)
        if i % 7 == 0:
            s = s.upper()
        if i % 3 == 0 and L > 400:  # low-complexity stretch without stops: long fragments at the contig edges
            s = s[: L // 3] + "gca" * (L // 9) + s[L // 3 + 3 * (L // 9) :]
        seqs.append(s)
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    nbad = 0
    for i, s in enumerate(seqs):
        o = oracle.run(s)
        st, genes = res[i]
        if o["status"] < 0:
            assert st == o["status"], (i, len(s))
            nbad += 1
            continue
        assert st == (1 if len(o["node_pos"]) > 2 and not len(o["path"]) else 0), (i, st)  # PHX_S_NOPATH: a graph (phanotate.py:63) whose target is unreachable
        assert np.array_equal(genes["left"], o["gene_left"]), (i, len(s))
        assert np.array_equal(genes["right"], o["gene_right"]), (i, len(s))
        assert np.array_equal(genes["strand"], o["gene_strand"].astype(np.int32))
        if len(genes):
            np.testing.assert_allclose(genes["score"], o["gene_score"], rtol=WTOL)
        if i % 9 == 0:
            check_contig(ann, i, s, o, genes, st if st != 1 else 0)
    assert nbad >= 5
    ann.close()


def test_large_contig_beyond_lds_parent_table(pa, oracle):
    """A 400 kb contig (~22 k nodes): more nodes than the LDS parent table of the SSSP kernel holds (the final walk then
    chases parent edges in global memory), 260 feature tiles, 2000 bitmap words per frame.  Everything must still
    equal the oracle."""
    seq = pa.synth_contig(31337, 400000)
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    o = oracle.run(seq)
    assert status == 0 and o["status"] == 0
    assert ann.globals(0).n_node > 12000
    check_contig(ann, 0, seq, o, genes, status)
    ann.close()


def test_contig_beyond_two_million_bases(pa, oracle):
    """2.2 Mbp in one contig: node ids no longer fit the 21 bits of the queued overlap-edge record, so k_edges<true> evaluates the
    overlap weights in place (DBatch.defer_overlap = 0); 1430 feature tiles, 11 500 bitmap words per frame, ~94 k nodes solved
    by the global-memory kernel.  Everything must still equal the oracle."""
    seq = pa.synth_contig(4242, 2200000)
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    o = oracle.run(seq)
    assert status == 0 and o["status"] == 0
    assert ann.globals(0).n_node > 80000
    check_contig(ann, 0, seq, o, genes, status)
    ann.close()


@pytest.mark.parametrize("ncodons,p_gtg,expect", [(3000, 0.12, "wave"), (3500, 0.20, "wave"), (5500, 0.27, "handed_back"), (4000, 0.45, "dense")])
def test_gc_rich_long_orf_wavefront_kernel_paths(pa, oracle, ncodons, p_gtg, expect):
    """A 9-12 kb reading frame in a GC-rich contig: p_stop is small, so the path sums still fit 128 bits and the contig
    goes to the wavefront-per-contig kernel.  The ORF's start nodes lie further back than its LDS distance ring holds
    (sources folded in from global memory), and its stop node has hundreds of in-edges (one per in-frame gtg): more than
    a wavefront's lanes (helper lanes + spill list); with more than 1280 the contig is handed to the workgroup kernel when
    the sweep reaches the stop node, and a contig with more than ~60 open nodes per 500 bp is routed there from the start.  Genes must equal the exact solution of the oracle's graph either way."""
    rng = np.random.RandomState(7)
    def gc_rich(n):
        return "".join(rng.choice(list("acgt"), n, p=[0.1, 0.4, 0.4, 0.1]))
    sense = [a + b + c for a in "cg" for b in "acgt" for c in "cg"]  # no stop codon starts with c/g... and none ends in c/g
    sense = [c for c in sense if c != "gtg"]
    body = "".join("gtg" if rng.rand() < p_gtg else sense[rng.randint(len(sense))] for _ in range(ncodons))
    seq = gc_rich(6000) + "atg" + body + "taa" + gc_rich(6000)
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    gl = ann.globals(0)
    assert status == 0
    assert gl.n_limbs == 2
    o = oracle.run(seq, stages=2)
    assert o["status"] == 0
    deg = np.bincount(o["edge_dst"])
    assert deg.max() > 64 * 4  # the stop node needs more than 64 lanes of 4 in-edges
    if expect == "wave":
        assert gl.sssp_kernel in (2, 3) and gl.sssp_handed_back == 0  # the wavefront kernel (3: its roomy configuration)
    elif expect == "handed_back":
        assert deg.max() > 1280  # ... and more than a window's staging area: the wavefront kernel passes the contig on when it gets there
        assert gl.sssp_kernel == 1 and gl.sssp_handed_back in (1, 2)
    else:
        # so many starts that more than 64 open nodes fall within 500 bp: k_edges<false> flags the contig and it never
        # enters the wavefront kernel
        assert gl.sssp_kernel == 1 and gl.sssp_handed_back == 0
    dist, want = _py_bellman_ford_genes(o)
    assert [(int(g["left"]), int(g["right"])) for g in genes] == want
    p, d = ann.path(0)
    assert abs(d - dist) <= abs(dist) * 1e-12
    check_exact_distances(ann, 0)  # k_sssp_wave<2> (helper lanes, spill list, folded sources) or k_sssp_lds<2>
    ann.close()


def test_context_reuse_across_growing_and_changing_batches(pa, oracle):
    """One context, batches of different shapes back to back.  A run enqueues everything against the buffers and the
    solver classes of the previous run and validates on the device: a batch that outgrows the buffers, or that needs an
    integer class / kernel the previous one did not, must transparently run again and still equal the oracle."""
    rng = np.random.RandomState(11)
    sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
    long_orf = pa.synth_contig(900, 3000).decode() + "atg" + "".join(rng.choice(sense, 5500)) + "taa" + pa.synth_contig(901, 3000).decode()
    batches = [
        [pa.synth_contig(500 + i, 3000).decode() for i in range(3)],
        [pa.synth_contig(600 + i, 20000).decode() for i in range(6)],           # outgrows every buffer
        [pa.synth_contig(700, 5000).decode(), long_orf],                        # a wide-integer contig: another solver class
        [pa.synth_contig(800 + i, 2000).decode() for i in range(2)] + ["acgtn", ""],  # smaller again, with degenerate contigs
        [pa.synth_contig(600 + i, 20000).decode() for i in range(6)],           # fits now: no host round trip
    ]
    ann = pa.Annotator()
    saw_wide = False
    for seqs in batches:
        res = ann.annotate(seqs)
        for i, (s, (status, genes)) in enumerate(zip(seqs, res)):
            if s is long_orf:
                saw_wide = ann.globals(i).n_limbs > 2
                dist, want = _py_bellman_ford_genes(oracle.run(s, stages=2))
                assert [(int(g["left"]), int(g["right"])) for g in genes] == want
            else:
                check_contig(ann, i, s, oracle.run(s), genes, status)
    assert saw_wide
    ann.close()


def test_benchmark_batch_slice_equals_oracle(pa, oracle):
    """200 of the benchmark's synthetic 50 kb contigs in one batch (the wavefront kernel's normal load: step-backs, helper
    lanes, spill lists all occur): every gene list equals the oracle's.  (`tools/validate_batch.py` does all 1000.)"""
    seqs = [pa.synth_contig(i, 50000) for i in range(0, 1000, 5)]
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    assert all(ann.globals(i).sssp_kernel in (2, 3) for i in range(len(seqs)))
    for s, (status, genes) in zip(seqs, res):
        o = oracle.run(s)
        assert status == o["status"] == 0
        assert np.array_equal(genes["left"], o["gene_left"]) and np.array_equal(genes["right"], o["gene_right"])
        assert np.array_equal(genes["strand"], o["gene_strand"])
        np.testing.assert_allclose(genes["score"], o["gene_score"], rtol=WTOL)
    for i in (0, 57, 123, 199):
        check_exact_distances(ann, i)  # the benchmark's kernel (k_sssp_wave<2>, 64-bit phases on the relative ring): exact
    ann.close()


def test_mixed_strand_start_rich_stretches_equal_oracle(pa, oracle):
    """Random contigs with stop-free stretches that are rich in start codons of BOTH strands (many close and open nodes within
    500 bp: narrow windows, helper lanes, spill lists, hand-backs to the workgroup kernel) and a wide GC range (ORF weights
    from 2^10 to beyond 2^64 in one contig: the relative distance ring moves its base).  Every distance is exact and every
    gene list is the oracle's."""
    rng = np.random.RandomState(20240928)
    seqs = []
    for _ in range(24):
        L = int(rng.choice([6000, 12000, 25000]))
        gc = rng.uniform(0.25, 0.7)
        s = rng.choice(list("acgt"), L, p=[(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
        for _ in range(int(rng.randint(1, 4))):
            n = int(rng.randint(150, 900))
            q = rng.uniform(0.03, 0.2)
            cod = ["atg", "gtg", "ttg", "cat", "cac", "caa", "gcc", "gac", "ctc", "aaa", "ggc", "acg"]
            text = "".join(rng.choice(cod, n, p=[q / 3] * 6 + [(1 - 2 * q) / 6] * 6))
            at = 3 * int(rng.randint(0, max(1, (L - 3 * n) // 3)))
            text = text[: max(0, L - at)]
            s[at:at + len(text)] = list(text)
        seqs.append("".join(s))
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    kernels = set()
    for i, (s, (status, genes)) in enumerate(zip(seqs, res)):
        o = oracle.run(s)
        if o["status"] == -7:
            continue  # beyond the oracle's 256-bit integers (python-int solves cover such cases elsewhere)
        assert status == o["status"], i
        if status < 0 or not len(o["path"]):
            continue
        kernels.add(ann.globals(i).sssp_kernel)
        assert abs(ann.path(i)[1] - int(o["path_dist"])) <= abs(int(o["path_dist"])) * 1e-12, i
        check_exact_distances(ann, i)
        assert np.array_equal(genes["left"], o["gene_left"]) and np.array_equal(genes["right"], o["gene_right"]) and np.array_equal(genes["strand"], o["gene_strand"]), i
        np.testing.assert_allclose(genes["score"], o["gene_score"], rtol=WTOL)
    assert 2 in kernels  # the wavefront kernel took part
    ann.close()


def test_cabi_annotate_and_struct_download(pa, oracle):
    """The one-call entry point of the C-ABI (phx_annotate -> phx_result[] with library-owned gene arrays, released by
    phx_free_results), called the way INTEGRATION.md's ctypes stub does, equals the flat download and the oracle."""
    import ctypes as C
    from phanotate_amd import _lib
    L = _lib.lib()
    seqs = [pa.synth_contig(40 + i, 9000) for i in range(3)] + [b"acgtnnacgt" * 30, b"acg"]
    ann = pa.Annotator()
    flat = ann.annotate(seqs)
    n = len(seqs)
    arr = (C.c_char_p * n)(*seqs)
    lens = (C.c_int64 * n)(*[len(s) for s in seqs])
    res = (_lib.Result * n)()
    assert L.phx_annotate(ann.h, n, arr, lens, res) == 0
    for i in range(n):
        st, genes = flat[i]
        assert res[i].status == st and res[i].n_genes == len(genes)
        for k in range(res[i].n_genes):
            g = res[i].genes[k]
            assert (g.left, g.right, g.strand, g.frame, g.score) == (genes["left"][k], genes["right"][k], genes["strand"][k], genes["frame"][k], genes["score"][k])
        o = oracle.run(seqs[i])
        assert st == o["status"]
        if st >= 0:
            assert np.array_equal(genes["left"], o["gene_left"]) and np.array_equal(genes["right"], o["gene_right"])
    L.phx_free_results(res, n)
    ann.close()


def test_many_uncovered_runs_bridges(pa, oracle):
    """Forty coding islands separated by 720 bp stop-rich spacers: every spacer is an uncovered run of more than 500 bases
    and gets its bridge edges (functions.py:334-354); far more than the 16 an earlier fixed-size table held."""
    spacer = "".join("tagctaactgattaa"[i % 15] for i in range(720))  # stop codons in all six frames every 15 bp
    seq = spacer.join(pa.synth_contig(300 + i, 1500).decode() for i in range(40))
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    o = oracle.run(seq)
    assert o["status"] == 0 and status == 0
    assert ann.globals(0).n_bridge > 16
    check_contig(ann, 0, seq, o, genes, status)
    ann.close()


def test_equal_length_alternatives_follow_the_reference_relaxation_order(pa, oracle):
    """Contigs on which another path of exactly the same integer length exists (tiny ORF weights: trunc(w*1000) has three
    digits).  The distances leave the choice open; the reference's solver decides by the order in which it relaxes the edges
    (in place, Graph.iteredges order: the oracle and the golden generator restate it).  k_inorder must land on the same genes.
    The contigs are tools/fuzz_gpu.py's (GC 20-80 %, repeats, start-/stop-rich stretches): about 2 % of them are ambiguous."""
    import sys

    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu

    rng = np.random.RandomState(101)
    seqs = [fuzz_gpu.make(rng) for _ in range(300)]
    rng = np.random.RandomState(7)
    seqs += [fuzz_gpu.make(rng) for _ in range(300)]
    ann = pa.Annotator()
    n_tie = n_changed = 0
    for b0 in range(0, len(seqs), 100):
        part = seqs[b0 : b0 + 100]
        res = ann.annotate(part)
        for i, (s, (status, genes)) in enumerate(zip(part, res)):
            o = oracle.run(s)
            if o["status"] == -7:
                continue
            assert status == (o["status"] if o["status"] < 0 else (1 if len(o["node_pos"]) > 2 and not len(o["path"]) else 0)), (b0 + i)
            if status != 0:
                continue
            t = ann.globals(i).tie
            assert t in (0, 1, 2)
            n_tie += t != 0
            n_changed += t == 2
            assert np.array_equal(genes["left"], o["gene_left"]) and np.array_equal(genes["right"], o["gene_right"]) and np.array_equal(genes["strand"], o["gene_strand"]), (b0 + i, t)
            if t:
                nd = ann.nodes(i)
                assert np.array_equal(nd["refidx"][ann.path(i)[0]], o["path"]), (b0 + i, t)
    assert n_tie >= 5, "only %d ambiguous contigs: the test no longer exercises k_inorder" % n_tie
    assert n_changed >= 1, "no contig needed its path replaced"
    ann.close()


def test_solver_alone_ties_follow_the_callers_edge_order(pa):
    """phx_solve (the fastpathz shim): small integer weights, zero-weight edges and zero-length cycles, so that nearly every
    node has several tight in-edges.  The path must be the one an in-place Bellman-Ford over the caller's edge list leaves."""
    import random

    rnd = random.Random(5)
    ann = pa.Annotator()
    checked = 0
    for trial in range(900):
        V = rnd.randint(3, 40 if trial % 10 else 400)
        edges, seen = [], set()
        for _ in range(rnd.randint(V, 4 * V)):
            u, v = rnd.randrange(V), rnd.randrange(V)
            if u == v or (u, v) in seen:
                continue
            seen.add((u, v))
            edges.append((u, v, rnd.choice([0, 0, 1, 1, 2, 3, -1]) if rnd.random() < 0.9 else rnd.randint(-2, 5)))
        s, t = 0, V - 1
        dist, par = inorder_bellman_ford(V, edges, s)
        if dist is None:
            continue  # a negative cycle
        path, d = ann.solve(V, [e[0] for e in edges], [e[1] for e in edges], [e[2] for e in edges], s, t)
        if dist[t] is None:
            assert path == [] and d is None
            continue
        want = [t]
        while want[-1] != s:
            want.append(edges[par[want[-1]]][0])
        want.reverse()
        assert d == dist[t]
        assert path == want, (trial, V)
        checked += 1
    assert checked > 150
    ann.close()


@pytest.mark.parametrize("case", [c for c in golden_cases()])
def test_dump_text_is_byte_exact(case, pa):
    """-d/--dump: the edge text of the reference (edges.py:17-23: repr(src), repr(dst), str(Decimal weight * 1000), in
    Graph.iteredges order), reproduced byte for byte by the library (phx_dump_text: csrc/phx_dec.c replays the reference's Decimal
    operations on the integers the GPU path delivers) and, as its cross-check, by the same replay with Python's own decimal
    (tests/decimal_replay.py).  The fixtures hold the md5 of the reference's own text."""
    import hashlib

    from decimal_replay import dump_lines

    g, name, seq = load_golden(case)
    if str(g["error"]):
        pytest.skip("the reference raises on this input")
    kw = golden_params(g)
    tr = golden_trnas(g)
    ann = pa.Annotator(pa.make_params(**kw))
    (status, genes), = ann.annotate([seq], trnas=None if tr is None else [tr])
    assert status >= 0
    text = ann.dump_text(0)
    assert text.count(b"\n") == len(g["edge_src"])
    assert hashlib.md5(text).hexdigest() == str(g["dump_md5"])
    if len(seq) < 60000 or case == "NC_001416.1":  # (the Python replay of T4 takes a minute)
        lines = dump_lines(ann, 0, seq, kw["start_codons"])
        assert ("".join(line + "\n" for line in lines)).encode() == text
    ann.close()


def test_paths_from_decimal_derived_integers(pa):
    """The reference hands fastpathz trunc(Decimal(w) * 1000) (28 digits), libphx solves on trunc(fp64(w) * 1000): the integers
    differ in the low digits of large weights.  On fuzz contigs the Decimal weights (tests/decimal_replay.py, the replay
    behind the byte-exact --dump) solved with the golden generator's in-order Bellman-Ford in python ints give the node path
    libphx returns (tools/decimal_check.py runs the same over thousands of contigs)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from decimal_check import solve
    from fuzz_gpu import make
    from decimal_replay import decimal_weights

    rng = np.random.RandomState(77)
    seqs = []
    while len(seqs) < 40:
        s = make(rng)
        if len(s) <= 6000 and not set(s.lower()) - set("acgt"):
            seqs.append(s)
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    checked = differing = 0
    for i, (status, genes) in enumerate(res):
        if status < 0 or ann.globals(i).n_node <= 2:
            continue
        nd, ed, w = decimal_weights(ann, i, seqs[i])
        differing += sum(1 for k in range(len(ed)) if int(w[k] * 1000) != int(np.trunc(float(ed["w"][k]) * 1000.0)))
        assert solve(nd, ed, w) == [int(x) for x in ann.path(i)[0]], i
        checked += 1
    assert checked >= 30 and differing > 0  # the two sets of integers do differ; the paths do not
    ann.close()


def test_batch_of_thousands_of_short_contigs(pa, oracle):
    """Batches beyond 1024 / 4096 contigs take other kernels in places: the layout scans in two passes (k_layout*_a / _b), gene records at
    a fixed place per contig + k_gene_pack instead of one shared counter, one-wavefront workgroups in k_inorder / k_score for short
    contigs.  6000 short contigs (tRNA hits on some, a bad letter in a few) in ONE batch give, contig by contig, what batches of 500 give,
    and a sample of them equals the oracle."""
    rng = np.random.RandomState(11)
    seqs, trnas = [], []
    for i in range(6000):
        L = int(rng.choice([150, 400, 900, 1600, 2500, 5000]))
        s = pa.synth_contig(40000 + i, L)
        if i % 997 == 0:
            s = s[:50] + b"x" + s[51:]
        seqs.append(s)
        hits = []
        if i % 7 == 0 and L >= 900:
            a = int(rng.randint(30, L - 200))
            hits.append((a, a + 72) if i % 14 else (a + 72, a))
        trnas.append(hits)
    big = pa.Annotator()
    got = big.annotate(seqs, trnas=trnas)
    small = pa.Annotator()
    want = []
    for k in range(0, len(seqs), 500):
        want += small.annotate(seqs[k:k + 500], trnas=trnas[k:k + 500])
    assert len(got) == len(want) == 6000
    n_genes = 0
    for i, ((gs, gg), (ws, wg)) in enumerate(zip(got, want)):
        assert gs == ws, i
        assert gg.tobytes() == wg.tobytes(), i
        n_genes += len(gg)
    assert n_genes > 6000 and sum(1 for st, _ in got if st < 0) >= 6
    for i in rng.choice(6000, 60, replace=False):
        o = oracle.run(seqs[i].decode(), trnas=trnas[i])
        st, g = got[i]
        if o["status"] == -7:  # beyond the oracle's 256-bit integers
            continue
        if o["status"] < 0:
            assert st == o["status"], i
            continue
        assert st >= 0 and g["left"].tolist() == np.asarray(o["gene_left"]).tolist() and g["right"].tolist() == np.asarray(o["gene_right"]).tolist(), i
    big.close(); small.close()


def test_batches_in_flight_equal_one_after_the_other(pa):
    """phx_run_async / phx_wait and pipeline.Pipeline (two contexts alternating on one GPU): a stream of different batches —
    growing sizes, so that a context finds its buffers too small in phx_wait and runs again; an empty batch; a contig with a bad
    letter — gives exactly what one Annotator gives batch by batch.  Calls on a context with a run in flight wait for it."""
    rng = np.random.RandomState(3)

    def batch(n, length, seed0):
        return [pa.synth_contig(seed0 + k, int(length * rng.uniform(0.6, 1.4))) for k in range(n)]

    batches = [batch(3, 6000, 10), batch(40, 12000, 100), [], batch(1, 50000, 7), batch(150, 3000, 300) + [b"acgtxacgt" * 50], batch(6, 25000, 900),
               batch(60, 15000, 2000), batch(2, 2000, 5)]
    one = pa.Annotator()
    want = [one.annotate_flat(b) for b in batches]
    with pa.Pipeline(depth=2) as pipe:
        got = pipe.annotate_flat(batches)
        assert len(got) == len(want)
        for (ws, wo, wg), (gs, go, gg) in zip(want, got):
            assert np.array_equal(ws, gs) and np.array_equal(wo, go)
            assert wg.tobytes() == gg.tobytes()
        again = list(pipe.run(batches[::-1]))  # the contexts are warm now: every run is asynchronous
        for (ws, wo, wg), (gs, go, gg) in zip(want[::-1], again):
            assert np.array_equal(ws, gs) and np.array_equal(wo, go) and wg.tobytes() == gg.tobytes()
    assert any((w[0] < 0).any() for w in want)  # the bad-letter contig kept its status, its batch its other results
    # a run in flight: taps, downloads and a new upload wait for it; wait() twice is harmless
    a = pa.Annotator()
    a.annotate_flat(batches[1])
    a.upload(batches[5])
    a.run_async()
    n_node = a.globals(0).n_node  # tap on a context with a run in flight
    a.wait(); a.wait()
    st, offs, genes = a.download_flat()
    assert n_node > 2 and genes.tobytes() == want[5][2].tobytes()
    a.run_async()
    a.upload(batches[3])  # replaces the batch under a run in flight: the run is collected first
    a.run_async()
    assert a.download_flat()[2].tobytes() == want[3][2].tobytes()
    a.run_async()
    a.close()  # destroying a context with a run in flight drains its stream first
    one.close()


def test_bench_sharded_path_smoke():
    """bench.py's N > 1 code path (config 5: shard.partition, per-rank shards, timed host-to-host region with the flat gather to
    rank 0) with two ranks on this one GPU over gloo.  The numbers are meaningless; the line must come out and account for every contig."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--contigs", "120", "--length", "20000", "--smoke-single-device", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["contigs_total"] == 120 and d["config"]["contigs_rank0"] == 60
    assert d["config"]["contigs_with_error_status"] == 0 and d["config"]["genes_called_total"] > 120 * 10
    assert d["value"] > 0 and d["host_to_host"]["value"] > 0


def test_many_ambiguous_contigs_outgrow_the_tie_scratch(pa, oracle):
    """k_inorder bump-allocates its scratch (36 B per node + 12 B per edge of a contig with equal-length alternatives) from a
    buffer that starts at 1 MB: a batch full of such contigs overflows it, the run reports how much it wants and is repeated
    with a larger buffer.  Every copy must come out like the first, and like the oracle."""
    import sys

    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu

    rng = np.random.RandomState(101)
    cands = [fuzz_gpu.make(rng) for _ in range(300)]
    ann = pa.Annotator()
    pick = None
    for b0 in range(0, 300, 100):
        res = ann.annotate(cands[b0 : b0 + 100])
        for i, (st, genes) in enumerate(res):
            if st == 0 and ann.globals(i).tie == 2 and ann.globals(i).n_node > 1500:
                pick = cands[b0 + i]
                break
        if pick:
            break
    assert pick is not None, "no large ambiguous contig among the candidates"
    o = oracle.run(pick)
    fresh = pa.Annotator()  # a context whose scratch is still at its initial size
    res = fresh.annotate([pick] * 96)
    for i, (st, genes) in enumerate(res):
        assert st == 0 and fresh.globals(i).tie == 2
        assert np.array_equal(genes["left"], o["gene_left"]) and np.array_equal(genes["right"], o["gene_right"]) and np.array_equal(genes["strand"], o["gene_strand"])
    ann.close()
    fresh.close()


def test_random_trna_hits_against_oracle(pa, oracle):
    """tRNA masking beyond the fixtures: 160 fuzz contigs, each with 0-9 random hits on both strands (some within 2000 bp of an
    end, clusters 10-200 bp apart, hits that share an end position, now and then the same hit twice = the reference's ValueError).
    Status, gene list incl. the tRNA features and, for a sample, every stage tap must equal the oracle's."""
    import sys

    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu

    rng = np.random.RandomState(4242)
    seqs, hits = [], []
    while len(seqs) < 160:
        s = fuzz_gpu.make(rng).lower()
        L = len(s)
        if L < 400 or not set(s) <= set("acgtnryswkmbvdh"):
            continue
        h = []
        p = int(rng.randint(1, L - 120))
        for _ in range(int(rng.randint(0, 10))):
            ln = int(rng.randint(60, 95))
            kind = rng.randint(6)
            if kind == 0:
                p = int(rng.randint(1, L - 120))
            elif kind == 1:
                p = int(rng.randint(1, min(L - 120, 1900)))
            elif kind == 2:
                p = int(rng.randint(max(1, L - 1990), L - 120))
            else:
                p = min(L - 120, p + ln + int(rng.randint(-40, 200)))
            p = max(1, p)
            a, z = p, p + ln
            h.append((z, a) if rng.rand() < 0.5 else (a, z))
        if h and rng.rand() < 0.06:
            h.append(h[int(rng.randint(len(h)))])  # the same hit twice
        seqs.append(s)
        hits.append(h)
    ann = pa.Annotator()
    n_on_path = n_dup = 0
    for b0 in range(0, len(seqs), 40):
        part, ph = seqs[b0 : b0 + 40], hits[b0 : b0 + 40]
        res = ann.annotate(part, trnas=ph)
        for i, (s, h, (status, genes)) in enumerate(zip(part, ph, res)):
            o = oracle.run(s, trnas=h)
            if o["status"] == -7:
                continue
            if o["status"] < 0:
                assert status == o["status"], (b0 + i, status, o["status"])
                n_dup += o["status"] == -6
                continue
            assert status == (1 if len(o["node_pos"]) > 2 and not len(o["path"]) else 0), (b0 + i)
            assert np.array_equal(genes["left"], o["gene_left"]) and np.array_equal(genes["right"], o["gene_right"]), (b0 + i)
            assert np.array_equal(genes["frame"], o["gene_frame"].astype(np.int32)), (b0 + i)
            n_on_path += int((np.abs(genes["frame"]) == 4).sum())
            if i % 8 == 0 and status == 0:
                check_contig(ann, i, s, o, genes, status)
    assert n_on_path >= 20, "only %d tRNA features ended up on a path" % n_on_path
    assert n_dup >= 1
    ann.close()


def _fuzz_contigs(n, seed, max_len=14000):
    import sys

    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu

    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        s = fuzz_gpu.make(rng).lower()
        if 400 <= len(s) <= max_len and set(s) <= set("acgt"):
            out.append(s)
    return out


def test_certificate_kernel_equals_its_python_statement(pa):
    """k_refine + k_certify (phx_refine.inc, phx_certify.inc) against Python.  (1) Soundness of what k_refine states: for every edge the
    reference's integer int(Decimal weight * 1000) — replayed with Python's decimal (dump.decimal_weights) — lies inside the device's
    bounds, and equals the solver's integer wherever the flag is cleared.  (2) k_certify against the prototype it was written from
    (tools/certify_probe.py: python ints, the same tree, sigma / kappa and per-edge test), contig by contig: with the product's bounds —
    where everything certifies — and with the bounds inflated (PHX_CREATE_CERT_TIGHT), where contigs with large ORF weights fail.
    Every solver kernel."""
    import sys

    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import certify_probe
    import decimal_replay as dump

    seqs = _fuzz_contigs(90, 77) + [pa.synth_contig(4000 + k, 30000).decode() for k in range(6)]
    n_fail = {False: 0, True: 0}
    n_flag = {False: [0, 0], True: [0, 0]}
    for flags in ((), ("cert_tight",), ("cert_tight", "solver_no_wave"), ("cert_tight", "solver_global", "cert_wide"), ("cert_wide",)):
        tight = "cert_tight" in flags
        ann = pa.Annotator(flags=flags + ("no_exact",))  # the kernel's own verdicts (no host re-solve behind them)
        for b0 in range(0, len(seqs), 48):
            part = seqs[b0 : b0 + 48]
            ann.upload(part)
            ann.run()
            cert = ann.certified()
            for i in range(len(part)):
                gl = ann.globals(i)
                if gl.status < 0 or gl.n_node <= 2:
                    assert cert[i] == 1
                    continue
                path = [int(x) for x in ann.path(i)[0]]
                if len(path) < 2:
                    assert cert[i] == 1
                    continue
                ed = ann.edges(i)
                n_flag[tight][0] += int(((ed["inexact"] != 0) & (ed["err"] != 0)).sum()); n_flag[tight][1] += len(ed)
                if (b0 + i) % 6 == 0 or flags == ():
                    viol = certify_probe.bounds_hold(ed, dump.decimal_weights(ann, i, part[i])[2])
                    assert not viol, (flags, b0 + i, viol[:3])
                why, ok = certify_probe.certify(ann.nodes(i), ed, ann.dist(i), path, repair="cert_wide" not in flags, ties=gl.tie != 0)
                assert int(cert[i]) == ok == gl.certified, (flags, b0 + i, why, int(cert[i]))
                n_fail[tight] += 1 - ok
        ann.close()
    assert n_fail[False] == 0 and n_fail[True] >= 30, (n_fail, n_flag)
    assert n_flag[False][0] * 200 < n_flag[False][1], n_flag  # with the product's bounds less than half a per cent of the edges keep an eps > 0 (those beyond ~1e22)
    a = pa.Annotator(flags=("no_certify",))
    a.annotate(seqs[:3])
    assert (a.certified() == -1).all()
    a.close()


def test_refine_bounds_hold_on_every_fixture(pa):
    """k_refine's statements against Python's decimal on the reference's own inputs — tRNA nodes (their pairs are scored 'same',
    functions.py:388-399; the tRNA edge is the constant -20), bridges over non-coding runs, terminal edges, non-default codon tables and
    -s weights with 28 digits (the neartie pair), the 1e126 ORF of edge_wide: for every edge of every non-error fixture the reference's
    integer lies inside the device's bounds, and equals the solver's integer wherever the flag is cleared; every fixture is certified
    or solved again, and the genes are the fixture's."""
    import sys

    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import certify_probe
    import decimal_replay as dump

    n_edges = n_flag = 0
    for case in golden_cases():
        g, name, seq = load_golden(case)
        if str(g["error"]) or len(seq) > 60000:
            continue
        kw = golden_params(g)
        tr = golden_trnas(g)
        ann = pa.Annotator(pa.make_params(**kw))
        (status, genes), = ann.annotate([seq], trnas=None if tr is None else [tr])
        assert status >= 0 and ann.certified()[0] in (1, 2), case
        assert np.array_equal(genes["left"], g["gene_left"]) and np.array_equal(genes["right"], g["gene_right"]), case
        ed = ann.edges(0)
        if len(ed):
            viol = certify_probe.bounds_hold(ed, dump.decimal_weights(ann, 0, seq, kw["start_codons"])[2])
            assert not viol, (case, viol[:3])
            n_edges += len(ed); n_flag += int(ed["inexact"].sum())
        ann.close()
    assert n_edges > 200000 and n_flag > 100, (n_edges, n_flag)


def test_uncertified_contigs_are_solved_again_on_the_references_integers(pa, oracle):
    """The host leg of the guarantee, below the C-ABI: a contig the device does not certify is solved again inside phx_download* on
    the reference's Decimal-derived integers (csrc/phx_exact.inc: the Decimal chain replayed by phx_dec.c + the in-order Bellman-Ford
    on exact integers) and its genes are replaced; phx_certified reports 2.  With the bounds inflated (cert_tight) that happens to
    most contigs of a batch: the result must equal the certified run's, byte for byte in the integers, the oracle's, and what the
    same replay gives in Python (dump.python_resolve: decimal.Decimal itself + python ints)."""
    import decimal_replay as dump

    seqs = _fuzz_contigs(40, 91, 9000) + [pa.synth_contig(4100, 20000).decode()]
    plain = pa.Annotator()
    want = plain.annotate_flat(seqs)
    assert plain.resolved == [] and (plain.certified() == 1).all()
    plain.close()
    tight = pa.Annotator(flags=("cert_tight",))
    got = tight.annotate_flat(seqs)
    assert len(tight.resolved) >= 10 and tight.resolved == [int(i) for i in np.nonzero(tight.certified() == 2)[0]]
    assert (tight.certified() != 0).all()
    assert got[0].tobytes() == want[0].tobytes() and got[1].tobytes() == want[1].tobytes()
    for f in ("left", "right", "strand", "frame"):
        assert np.array_equal(got[2][f], want[2][f]), f
    np.testing.assert_allclose(got[2]["score"], want[2]["score"], rtol=1e-12)
    raw = tight.download_flat(exact=False)  # what the device itself reported: the same here
    assert raw[2].tobytes() == want[2].tobytes() and tight.resolved == []
    again = tight.download_flat()           # and the guarantee is back on
    assert again[2].tobytes() == got[2].tobytes() and len(tight.resolved) >= 10
    for i in tight.resolved[:12]:
        o = oracle.run(seqs[i])
        g = got[2][got[1][i] : got[1][i + 1]]
        assert np.array_equal(g["left"], o["gene_left"]) and np.array_equal(g["right"], o["gene_right"]) and np.array_equal(g["frame"], o["gene_frame"].astype(np.int32))
        py = dump.python_resolve(tight, i, seqs[i])
        assert [(int(x["left"]), int(x["right"]), int(x["strand"]), int(x["frame"]), float(x["score"])) for x in g] == py, i
    # the struct-of-pointers download goes through the same substitution
    per = tight.download()
    for i in tight.resolved[:12]:
        assert per[i][1].tobytes() == got[2][got[1][i] : got[1][i + 1]].tobytes()
    tight.close()
    # a context that must not spend host time: the device's lists, cert stays 0
    ne = pa.Annotator(flags=("cert_tight", "no_exact"))
    r = ne.annotate_flat(seqs)
    assert ne.resolved == [] and (ne.certified() == 0).sum() >= 10 and r[2].tobytes() == want[2].tobytes()
    ne.close()


def test_contigs_without_a_graph_are_certified(pa):
    """A batch whose contigs have no ORF at all (two nodes: source and target) launches no k_certify for their integer class: such
    contigs count as certified (nothing to decide) and are never handed to the host re-solve."""
    seqs = ["acgtacgtac" * 3, "t" * 80, "ca" * 30]
    ann = pa.Annotator()
    st, offs, genes = ann.annotate_flat(seqs)
    assert st.tolist() == [0, 0, 0] and len(genes) == 0
    assert ann.certified().tolist() == [1, 1, 1] and ann.resolved == []
    assert [ann.globals(i).n_node for i in range(3)] == [2, 2, 2]
    ann.close()


def test_fp64_and_decimal_integers_disagree_on_the_neartie_pair(pa):
    """A constructed contig whose fp64-integer path and Decimal-integer path differ (tests/golden/make_neartie.py, generated through the
    reference): the same 24 kb contig with two -s weights for gtg that are 5e-28 apart and between which the reference's path changes —
    a 3.5e21 ORF decides.  A double reads both flags as the same number, so the device's first pass gives ONE path for both; the final
    genes must be the reference's for both.  For the one it got wrong the certificate fails (k_refine knows the reference's integers,
    the path is not optimal for them), the contig is solved again on the host in the reference's own arithmetic (phx_exact.inc), and the
    result differs from what the device alone reported."""
    import decimal_replay as dump

    n_changed = 0
    raw_paths = []
    for case in ("neartie_lo", "neartie_hi"):
        g, name, seq = load_golden(case)
        kw = golden_params(g)
        ann = pa.Annotator(pa.make_params(**kw))
        st, offs, genes = ann.annotate_flat([seq])
        cert = int(ann.certified()[0])
        assert st[0] == 0 and cert in (1, 2) and ann.globals(0).certified == cert
        assert np.array_equal(genes["left"], g["gene_left"]) and np.array_equal(genes["right"], g["gene_right"]) and np.array_equal(genes["strand"], g["gene_strand"].astype(np.int32)), case
        np.testing.assert_allclose(genes["score"], g["gene_score"], rtol=1e-6)
        py = dump.python_resolve(ann, 0, seq, kw["start_codons"])  # decimal.Decimal itself + python ints
        assert [(int(x["left"]), int(x["right"]), int(x["strand"])) for x in genes] == [t[:3] for t in py]
        raw = ann.download_flat(exact=False)[2]
        raw_paths.append([(int(x["left"]), int(x["right"])) for x in raw])
        if raw.tobytes() != genes.tobytes():
            n_changed += 1
            assert cert == 2
            assert not np.array_equal(raw["left"], genes["left"]) or not np.array_equal(raw["right"], genes["right"])
        # the whole C-ABI path (struct of pointers) delivers the same
        (st2, g2), = ann.annotate([seq])
        assert g2.tobytes() == genes.tobytes()
        ann.close()
    assert raw_paths[0] == raw_paths[1], "fp64 reads both weights as one number"
    assert n_changed == 1, "the host re-solve must change exactly one of the two results"


def test_a_cycle_of_negative_length_is_reported(pa, oracle):
    """Overlap edges point backwards, so the ORF graph can hold a cycle, and on a contig of many short overlapping frames its length can
    be negative (tools/fuzz_gpu.py, seed 949, contig 176: 2000 bases, 123 nodes — the first such contig in 110 000): the relaxation never
    settles.  What fastpathz does then is unknown (the dependency is absent); libphx reports PHX_S_NEGCYCLE (-9) for that contig and goes
    on, the oracle now says the same instead of walking parents that run in a circle."""
    import sys

    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu

    rng = np.random.RandomState(949)
    seqs = [fuzz_gpu.make(rng) for _ in range(177)]
    cyc = seqs[176]
    assert oracle.run(cyc)["status"] == -9
    for flags in ((), ("solver_no_wave",), ("solver_global",)):
        ann = pa.Annotator(flags=flags)
        res = ann.annotate([seqs[175], cyc, seqs[174]])
        assert [r[0] for r in res] == [oracle.run(seqs[175])["status"], -9, oracle.run(seqs[174])["status"]] and len(res[1][1]) == 0, flags
        assert ann.certified()[1] == 1
        ann.close()


def test_path_sums_beyond_1088_bits_go_to_the_host_and_leave_the_batch_alone(pa):
    """The reference's solver has no width limit (GMP, CHANGELOG.md:11-13); libphx's device kernels stop at 1088-bit integers.  A contig
    that needs more (an open reading frame of 24 000 codons without a stop, ~340 in-frame starts: fp64 itself overflows) is solved on
    the host in the reference's arithmetic when exactness is on (round 5; certified == 2) and reported as PHX_S_OVERFLOW (-7), without
    genes, when it is off; the rest of the batch is not touched either way."""
    rng = np.random.RandomState(12)
    sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
    body = "".join(sense[i] for i in rng.randint(0, len(sense), 24000))
    huge = pa.synth_contig(320, 2000).decode() + "atg" + body + "taa" + pa.synth_contig(321, 2000).decode()
    other = [pa.synth_contig(322 + k, 9000) for k in range(3)]
    ann = pa.Annotator()
    want = ann.annotate_flat(other)
    ann.upload([other[0], huge, other[1], other[2]])
    ann.run()
    st, offs, genes = ann.download_flat(exact=False)
    assert st.tolist() == [0, -7, 0, 0] and offs[2] == offs[1]
    keep = np.concatenate([genes[offs[0]:offs[1]], genes[offs[2]:]])
    assert keep.tobytes() == want[2].tobytes()
    st, offs, genes = ann.download_flat()
    assert st.tolist() == [0, 0, 0, 0] and offs[2] > offs[1]
    keep = np.concatenate([genes[offs[0]:offs[1]], genes[offs[2]:]])
    assert keep.tobytes() == want[2].tobytes()
    mine = genes[offs[1]:offs[2]]
    assert (np.diff(mine["left"]) > 0).all() and mine["right"].max() > 70000 and not np.isfinite(mine["score"]).all()  # the long ORF is called: its score is -INF in '%E' as in the reference
    assert ann.certified().tolist() == [1, 2, 1, 1]
    ann.close()


def test_create_flags_pick_the_solver_kernel_not_the_result(pa):
    """PHX_CREATE_SOLVER_GLOBAL / _NO_WAVE / _NO_GRAPH / _SIZE_EVERY_RUN (phx_create_ex; they replace round 2's environment switches):
    the same genes, byte for byte, whichever shortest-path kernel solves the contigs and however the run is enqueued."""
    seqs = [pa.synth_contig(300 + i, 9000 + 1300 * i) for i in range(12)]
    base = pa.Annotator()
    want = base.annotate_flat(seqs)
    assert all(base.globals(i).sssp_kernel in (2, 3) for i in range(len(seqs)))
    base.close()
    for flags, kern in ((("solver_global",), 0), (("solver_no_wave",), 1), (("no_graph", "size_every_run"), 2), (("no_duo",), 2)):
        a = pa.Annotator(flags=flags)
        got = a.annotate_flat(seqs)
        for _ in range(3):
            a.run()
        again = a.download_flat()
        assert all(a.globals(i).sssp_kernel in ((2, 3) if kern == 2 else (kern,)) for i in range(len(seqs))), flags
        for x, y, z in zip(want, got, again):
            assert x.tobytes() == y.tobytes() == z.tobytes(), flags
        a.close()
    import ctypes as C
    from phanotate_amd import _lib
    h = C.c_void_p()
    assert _lib.lib().phx_create_ex(C.byref(pa.make_params()), 0, None, 1 << 20, C.byref(h)) == -1  # unknown flag


def test_trna_hit_outside_the_contig_fails_that_contig_only(pa, oracle):
    """phx_set_trnas: a hit with an end beyond 1..L (a finder run on a circular topology, a truncated parse) gives that contig the
    status PHX_S_BADTRNA; the other contigs of the batch run with their own hits (it used to fail the whole batch, and a
    forward start at L + 1 slipped through one position past the node bitmaps)."""
    seqs = [pa.synth_contig(900 + i, 7000 + 500 * i) for i in range(5)]
    L = [len(s) for s in seqs]
    hits = [[(1200, 1275)], [(L[1] - 40, L[1] + 33)], [(L[2] + 1, L[2] + 2)], [(80, -5)], [(3080, 3000), (5000, 5080)]]
    ann = pa.Annotator()
    res = ann.annotate(seqs, trnas=hits)
    assert [r[0] for r in res] == [0, -4, -4, -4, 0]
    assert all(len(res[i][1]) == 0 for i in (1, 2, 3))
    for i in (0, 4):
        o = oracle.run(seqs[i], trnas=hits[i])
        check_contig(ann, i, seqs[i], o, res[i][1], res[i][0])
    res2 = ann.annotate(seqs, trnas=[[], [], [], [], []])  # the statuses do not stick to the context
    assert [r[0] for r in res2] == [0] * 5
    ann.close()


@pytest.mark.parametrize("ncodons", [2200, 3000, 3800])
def test_256_bit_contigs_on_the_wavefront_kernel(pa, oracle, ncodons):
    """A 6.6-11 kb stop-free reading frame in an otherwise ordinary 40 kb contig: the path sums need more than 128 bits (a
    phage tail-fibre-sized gene is enough).  Such contigs are solved by k_sssp_wave<4> (the same wavefront kernel on 256-bit
    distances) instead of dropping to the workgroup kernel: distances exact, genes = the exact solution of the oracle's graph."""
    rng = np.random.RandomState(ncodons)
    sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
    seqs = []
    for k in range(6):
        body = "".join(rng.choice(sense, ncodons))
        seqs.append(pa.synth_contig(900 + k, 20000).decode() + "atg" + body + "taa" + pa.synth_contig(1900 + k, 20000).decode())
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    n4 = 0
    for i, (s, (status, genes)) in enumerate(zip(seqs, res)):
        gl = ann.globals(i)
        assert status == 0
        if gl.n_limbs == 4:
            n4 += 1
            assert gl.sssp_kernel in (2, 3) and gl.sssp_handed_back == 0, (i, gl.sssp_kernel, gl.sssp_handed_back)
        o = oracle.run(s, stages=2)
        dist, want = _py_bellman_ford_genes(o)
        assert [(int(g["left"]), int(g["right"])) for g in genes] == want, i
        check_exact_distances(ann, i)
    assert n4 >= 1, "no contig of this batch needed 256 bits"
    ann.close()


def test_wavefront_kernel_in_both_configurations_equals_the_oracle(pa, oracle):
    """k_sssp_wave<2> exists twice: a tight configuration (29 KB of LDS per wavefront, so that another batch's workgroups fit beside
    it) and a roomy one for the 128-bit contigs whose windows need more staged in-edges or spill entries (k_wave_plan decides,
    phx_globals.sssp_kernel 2 / 3).  The generator's dense 25-40 kb contigs need the roomy one: same genes as the oracle either way."""
    seqs = _fuzz_contigs(60, 4, max_len=50000) + _fuzz_contigs(60, 5, max_len=50000)
    ann = pa.Annotator()
    res = ann.annotate(seqs)
    kinds = {}
    for i in range(len(seqs)):
        gl = ann.globals(i)
        if gl.n_node > 2 and gl.n_limbs == 2:
            kinds.setdefault(gl.sssp_kernel, []).append(i)
    assert kinds.get(3) and kinds.get(2), {k: len(v) for k, v in kinds.items()}
    for i in kinds[3] + kinds[2][:12]:
        o = oracle.run(seqs[i])
        status, genes = res[i]
        assert status == o["status"] == 0, i
        check_exact_distances(ann, i)
        assert np.array_equal(genes["left"], o["gene_left"]) and np.array_equal(genes["right"], o["gene_right"]) and np.array_equal(genes["strand"], o["gene_strand"]), i
    ann.close()
    # Batches of up to 800 contigs launch the tight solver BESIDE its planner and let it follow the planner's progress counter
    # (DMeta.plan_prog).  A contig the tight planner gives up half-way is the hard case: its solver has consumed windows already when
    # the counter turns to -1, must leave the contig alone, and the roomy pair takes over.  Alone in a batch (the planner then runs far
    # longer than the edge fill, the solver really waits on the counter), on repeated runs of the captured graph, and with the side
    # streams folded into one (one_stream: nothing is streamed) the records must be the batch's.
    lone = pa.Annotator()
    serial = pa.Annotator(flags=("one_stream",))
    for i in kinds[3][:4] + kinds[2][:2]:
        for a in (lone, serial):
            (st, g), = a.annotate([seqs[i]])
            assert st == res[i][0] and g.tobytes() == res[i][1].tobytes(), i
            a.run(); a.run()
            (st2, g2), = a.download()
            assert st2 == st and g2.tobytes() == g.tobytes(), i
    lone.close(); serial.close()


def test_features_behind_the_upload_equal_features_inside_the_run(pa):
    """phx_upload launches k_features piece by piece behind its copies and the first run after it starts at the ORF scan; a repeated
    run, a run after phx_set_trnas, a profiled context (stage timers) and phx_attach do the whole path inside the run: same records."""
    import torch

    seqs = [pa.synth_contig(700 + i, 30000 + 997 * i) for i in range(12)] + [b"acgtnnacgtryk" * 40, b"acg", b"acgtxacgt" * 30]
    a = pa.Annotator()
    first = a.annotate_flat(seqs)            # features behind the upload
    a.run(); again = a.download_flat()       # the whole path inside the run
    a.upload(seqs); a.set_trnas([[] for _ in seqs]); a.run(); with_trna_call = a.download_flat()
    a.upload(seqs); a.set_trnas(None); a.run(); no_finder = a.download_flat()
    b = pa.Annotator(); b.set_profiling(True)
    prof = b.annotate_flat(seqs)
    cat = b"".join(seqs)
    t = torch.frombuffer(bytearray(cat), dtype=torch.uint8).cuda()
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
    c = pa.Annotator(stream=torch.cuda.current_stream().cuda_stream)
    c.attach(t.data_ptr(), offs); c._keep = (seqs,); c.run(); att = c.download_flat()
    for other in (again, with_trna_call, no_finder, prof, att):
        for x, y in zip(first, other):
            assert x.tobytes() == y.tobytes()
    assert first[0].tolist()[-3:] == [0, -3, -2]
    for i in (0, 5, 12):  # per-position records as well (bins, bitmaps -> ORF table)
        assert a.orfs(i).tobytes() == b.orfs(i).tobytes() == c.orfs(i).tobytes()
    a.close(); b.close(); c.close()


@pytest.mark.parametrize("ncodons", [7500, 9500])
def test_512_bit_contigs_on_the_wavefront_kernel(pa, oracle, ncodons):
    """A 22-28 kb stop-free reading frame with about a hundred in-frame starts: the path sums need 512 bits.  k_sssp_wave<8> (the
    wavefront kernel's number type is generic in its limb count; its cached in-edge weights stay 64 bits wide and the wide ones sit
    on the LDS side list) takes such a contig as long as its stop node's in-edges fit the lanes and the spill list: distances
    exact against python ints, genes = the exact solution of the oracle's graph."""
    rng = np.random.RandomState(ncodons)
    sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
    quiet = [c for c in sense if c not in ("atg", "gtg", "ttg")]
    body = ["atg" if rng.rand() < 0.01 else quiet[rng.randint(len(quiet))] for _ in range(ncodons)]
    seq = pa.synth_contig(900, 15000).decode() + "atg" + "".join(body) + "taa" + pa.synth_contig(1900, 15000).decode()
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    gl = ann.globals(0)
    assert status == 0
    assert gl.n_limbs == 8 and gl.sssp_kernel == 2 and gl.sssp_handed_back == 0, (gl.n_limbs, gl.sssp_kernel, gl.sssp_handed_back)
    o = oracle.run(seq, stages=2)
    dist, want = _py_bellman_ford_genes(o)
    assert [(int(g["left"]), int(g["right"])) for g in genes] == want
    check_exact_distances(ann, 0)
    assert ann.certified()[0] in (0, 1)
    ann.close()


def test_512_bit_contig_whose_stop_node_has_450_in_edges(pa, oracle):
    """The same with ordinary codon usage in the long frame: 5 % of its 9000 codons are starts, the stop node behind it collects ~450
    ORF in-edges — more than a window's lanes (64 x 4) hold; the rest goes to the spill list, which k_sssp_wave<8> has room for since
    round 4 (448 entries; the contig used to fall to k_sssp_lds<8>: 1.93 ms of solver instead of 1.53).  Exact distances, exact genes."""
    rng = np.random.RandomState(42)
    sense = [a + b + c for a in "acgt" for b in "acgt" for c in "acgt" if a + b + c not in ("taa", "tag", "tga")]
    seq = pa.synth_contig(900, 20000).decode() + "atg" + "".join(rng.choice(sense, 9000)) + "taa" + pa.synth_contig(1900, 20000).decode()
    ann = pa.Annotator()
    (status, genes), = ann.annotate([seq])
    gl = ann.globals(0)
    assert status == 0 and gl.n_limbs == 8 and gl.sssp_kernel == 2 and gl.sssp_handed_back == 0, (gl.n_limbs, gl.sssp_kernel, gl.sssp_handed_back)
    o = oracle.run(seq, stages=2)
    dist, want = _py_bellman_ford_genes(o)
    assert [(int(g["left"]), int(g["right"])) for g in genes] == want
    check_exact_distances(ann, 0)
    ann.close()
