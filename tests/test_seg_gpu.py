"""Segments (phx_sssp_seg.inc): in batches of up to 64 contigs a contig's shortest path (fastpathz, phanotate.py:56-64) is swept by up to 16
wavefront pairs side by side, each in a frame of its own, and k_seg_merge joins the frames by a constant each and PROVES the result (no edge
improves its head, every reached node has a tight parent, the parents lead to the source).  Bar: whatever is delivered is bit-equal to the
one-sweep solver (PHX_CREATE_NO_SEG) — records, distances, parents' path — and to exact python-int distances; a contig that cannot be
proven is solved by one sweep in the same run and delivers the same."""
import os
import subprocess
import sys

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
    fb = a.seg_fallbacks()
    a.close()
    return flat, dist, path, segs, fb


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
    assert s[3] == (3 if len(seq) >= 12000 else 0)  # segments were used in every run (a contig too short for two of them keeps the one sweep, which follows its planner)
    if case in ("NC_001416.1", "NC_000866.1", "synth50k_0", "synth50k_5"):
        assert s[4] == 0, s[4]  # these are proven at the default margin


def test_small_batches_in_segments(pa):
    """8 and 30 contigs of 3-60 kb (segments are used in batches of up to 32 contigs), graph replay included."""
    rng = np.random.RandomState(11)
    for n in (8, 30):
        seqs = [pa.synth_contig(900 + 13 * i + n, int(rng.randint(3000, 60000))) for i in range(n)]
        s = _run(pa, seqs, (), 3, (0, n // 2, n - 1))
        w = _run(pa, seqs, ("no_seg",), 2, (0, n // 2, n - 1))
        _same(s, w)
        assert s[3] != 0


def test_staged_front_end_and_solvers_behind_their_planners(pa):
    """Up to 4 contigs run their front end as one launch (k_front) and the segments' solvers beside their planner wavefronts; PHX_CREATE_NO_FUSE
    keeps the staged kernels and launches the solvers behind the planners: the same records either way."""
    seqs = [pa.synth_contig(70 + i, 11000 + 2000 * i) for i in range(3)]  # (39 kb in all: the fused front end takes up to 40 kb)
    f = _run(pa, seqs, (), 4, (0, 2))
    s = _run(pa, seqs, ("no_fuse",), 4, (0, 2))
    w = _run(pa, seqs, ("no_seg",), 2, (0, 2))
    _same(f, w)
    _same(s, w)
    assert f[3] == 4 and s[3] == 4


def test_a_margin_too_short_is_caught_and_one_sweep_solves_the_contig(pa):
    """PHX_SEG_MARGIN_BP=300: the frames have not run together; k_seg_join must refuse (an edge that improves its head, a node without a
    tight parent) and the one-sweep kernels behind it solve the flagged contigs in the same run; later runs of the batch go without segments."""
    seqs = [pa.synth_contig(40 + i, 50000) for i in range(6)]
    s = _run(pa, seqs, (), 2, (0, 5), env={"PHX_SEG_MARGIN_BP": "300"})
    w = _run(pa, seqs, ("no_seg",), 1, (0, 5))
    _same(s, w)
    assert s[3] == 1 and s[4] >= 3, (s[3], s[4])  # the first run used segments and most contigs fell back; the second took the one-sweep kernels


def test_margins_between_too_short_and_default(pa):
    """Whatever the margin, the delivered result is the one-sweep solver's (Lambda and T4; 500 bp ... 12 kb)."""
    for case in ("NC_001416.1", "NC_000866.1"):
        seq = load_golden(case)[2]
        w = _run(pa, [seq], ("no_seg",), 1, (0,))
        for bp in ("500", "1500", "3000", "12000"):
            s = _run(pa, [seq], (), 2, (0,), env={"PHX_SEG_MARGIN_BP": bp})
            _same(s, w)


SEG_WORKER = """
import os, sys, time
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import phanotate_amd as pa
from conftest import load_golden
rank = int(sys.argv[1])
seq = load_golden("NC_001416.1" if rank %% 2 == 0 else "NC_000866.1")[2]
ref = pa.Annotator(flags=("no_seg",))
want = ref.annotate_flat([seq])
ref.close()
ann = pa.Annotator()
first = ann.annotate_flat([seq])
assert all(a.tobytes() == b.tobytes() for a, b in zip(first, want)), "rank %%d: segments differ from one sweep" %% rank
worst = 0.0
for r in range(150):
    t0 = time.perf_counter()
    ann.run()
    worst = max(worst, (time.perf_counter() - t0) * 1e3)
    if r %% 10 == 0:
        got = ann.download_flat()
        assert all(a.tobytes() == b.tobytes() for a, b in zip(got, want)), "rank %%d run %%d differs" %% (rank, r)
print("SEG_OK", rank, ann.seg_runs(), ann.seg_fallbacks(), ann.plan_timeouts(), "%%.2f" %% worst, flush=True)
"""


def test_six_processes_of_lone_genomes_share_the_gpu(tmp_path):
    """Six processes, each running Lambda or T4 alone 150 times: segments with their solvers beside their planner wavefronts while other
    processes hold SIMDs.  Every downloaded result equals the one-sweep solver's; a planner time-out (28 ms, once) may switch a context
    back to launching solvers behind planners, nothing more."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "seg_worker.py"
    script.write_text(SEG_WORKER % (root, root))
    procs = [subprocess.Popen([sys.executable, str(script), str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(6)]
    outs = [p.communicate(timeout=900) for p in procs]
    lines = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-2000:]
        lines += [l.split() for l in so.splitlines() if l.startswith("SEG_OK")]
    assert len(lines) == 6
    print("runs in segments, fallbacks, planner time-outs, worst run (ms):", [(int(l[2]), int(l[3]), int(l[4]), float(l[5])) for l in lines])
    assert all(int(l[2]) > 0 for l in lines)
    assert max(float(l[5]) for l in lines) < 400.0
