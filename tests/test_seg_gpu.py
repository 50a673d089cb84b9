"""Segments (phx_sssp_seg.inc): in batches of up to 64 contigs a contig's shortest path (fastpathz, phanotate.py:56-64) is swept by up to 16
wavefront pairs side by side, each in a frame of its own, and k_seg_merge joins the frames by a constant each and PROVES the result (no edge
improves its head, every reached node has a tight parent, the parents lead to the source).  Bar: whatever is delivered is bit-equal to the
one-sweep solver (PHX_CREATE_NO_SEG) — records, distances, parents' path — and to exact python-int distances; a run that cannot be proven
is repeated without segments inside phx_run and delivers the same."""
import os

import numpy as np
import pytest

from conftest import exact_dist_from_device_edges, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import phanotate_amd

    return phanotate_amd


def _exact(ann, i):
    want = exact_dist_from_device_edges(ann, i)
    got = ann.dist(i)
    assert len(got) == len(want)
    bad = [v for v in range(len(want)) if got[v] != want[v]]
    assert not bad, "contig %d: %d of %d distances differ, first at node %d" % (i, len(bad), len(want), bad[0])


def _run(pa, seqs, flags, runs, sample, env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        a = pa.Annotator(flags=flags)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    flat = a.annotate_flat(seqs)
    for _ in range(runs - 1):
        a.run()
    if runs > 1:
        flat = a.download_flat()
    dist = {i: list(a.dist(i)) for i in sample}
    path = {i: (a.path(i)[0].tolist(), int(a.path(i)[1])) for i in sample}
    for i in sample:
        _exact(a, i)
    segs = a.seg_runs()
    a.close()
    return flat, dist, path, segs


def _same(x, y):
    for p, q in zip(x[0], y[0]):
        assert p.tobytes() == q.tobytes()
    assert x[1] == y[1] and x[2] == y[2]


@pytest.mark.parametrize("case", ["NC_001416.1", "NC_000866.1", "phiX174", "synth50k_0", "synth50k_5", "synth6k_101", "edge_bridge", "edge_nrun"])
def test_lone_contig_in_segments_equals_one_sweep(pa, case):
    seq = load_golden(case)[2]
    s = _run(pa, [seq], (), 3, (0,))
    w = _run(pa, [seq], ("no_seg",), 3, (0,))
    _same(s, w)
    assert w[3] == 0
    assert s[3] != 0  # segments were used (negative: tried, could not be proven, repeated with one sweep: allowed, the result stands)
    if case in ("NC_001416.1", "NC_000866.1", "synth50k_0", "synth50k_5"):
        assert s[3] == 3, s[3]  # these are proven at the default margin, every run


def test_small_batches_in_segments(pa):
    """8 and 40 contigs of 3-60 kb (16 segments x 64 contigs fill the chip), graph replay included."""
    rng = np.random.RandomState(11)
    for n in (8, 40):
        seqs = [pa.synth_contig(900 + 13 * i + n, int(rng.randint(3000, 60000))) for i in range(n)]
        s = _run(pa, seqs, (), 3, (0, n // 2, n - 1))
        w = _run(pa, seqs, ("no_seg",), 2, (0, n // 2, n - 1))
        _same(s, w)
        assert s[3] != 0


def test_a_margin_too_short_is_caught_and_the_run_repeated(pa):
    """PHX_SEG_MARGIN_BP=1: 16 nodes (~400 bp) of margin — the frames have not run together; k_seg_merge must refuse (an offset that is not
    one constant, an edge that improves its head, a node without a tight parent), phx_run repeats the batch with one sweep per contig, and
    the context stays that way."""
    seqs = [pa.synth_contig(40 + i, 50000) for i in range(6)]
    s = _run(pa, seqs, (), 2, (0, 5), env={"PHX_SEG_MARGIN_BP": "1"})
    w = _run(pa, seqs, ("no_seg",), 1, (0, 5))
    _same(s, w)
    assert s[3] < 0, s[3]


def test_margins_between_too_short_and_default(pa):
    """Whatever the margin, the delivered result is the one-sweep solver's (Lambda and T4; 500 bp ... 12 kb)."""
    for case in ("NC_001416.1", "NC_000866.1"):
        seq = load_golden(case)[2]
        w = _run(pa, [seq], ("no_seg",), 1, (0,))
        for bp in ("500", "1500", "3000", "12000"):
            s = _run(pa, [seq], (), 2, (0,), env={"PHX_SEG_MARGIN_BP": bp})
            _same(s, w)
