"""Whole batches against the CPU oracle (oracle/phx_oracle.c, the restatement of functions.py:143-454 + the in-place Bellman-Ford of
phanotate.py:56-64): what tools/validate_batch.py and the fuzz rounds did by hand in rounds 1-5, as tests.

* all 1000 contigs of the benchmark batch (BASELINE config 4) in ONE 1000-contig batch — the only place the non-streamed 1000-contig launch
  order (k_sssp_duo behind k_wave_plan, coded gap edges, the captured graph) is compared with the oracle contig by contig;
* a slice of the fuzz generator's contigs and of random synthetic ones, lone and in batches of 2-32, i.e. through the segments
  (phx_sssp_seg.inc) — against the oracle, not against the one-sweep solver.

The oracle side runs in worker processes started with `spawn` (this process holds the GPU runtime)."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _orc(seq):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import oracle

    o = oracle.run(seq)
    return int(o["status"]), np.asarray(o["gene_left"], np.int64).tolist(), np.asarray(o["gene_right"], np.int64).tolist(), np.asarray(o["gene_strand"], np.int64).tolist()


def _oracle_all(seqs):
    n = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    with mp.get_context("spawn").Pool(n) as pool:
        return pool.map(_orc, seqs, chunksize=max(1, len(seqs) // (8 * n)))


def _same(res, want, what):
    bad = []
    for i, ((st, genes), (ost, gl, gr, gs)) in enumerate(zip(res, want)):
        if not (st == ost and genes["left"].tolist() == gl and genes["right"].tolist() == gr and genes["strand"].tolist() == gs):
            bad.append(i)
    assert not bad, "%s: %d of %d contigs differ from the oracle, first %s" % (what, len(bad), len(res), bad[:5])


@pytest.fixture(scope="module")
def pa():
    import phanotate_amd

    return phanotate_amd


def test_the_1000_benchmark_contigs_in_one_batch_equal_the_oracle(pa):
    seqs = [pa.synth_contig(i, 50000) for i in range(1000)]
    ann = pa.Annotator()
    first = ann.annotate(seqs)  # the sizing run
    kern = [ann.globals(i).sssp_kernel for i in range(0, 1000, 50)]
    assert set(kern) <= {2, 3}  # the wavefront kernels (k_sssp_duo / its roomy fallback), not the workgroup kernel
    flat1 = ann.download_flat()
    for _ in range(3):  # steady state: the third run on a layout replays the captured graph
        ann.run()
    flat2 = ann.download_flat()
    assert all(a.tobytes() == b.tobytes() for a, b in zip(flat1, flat2))
    assert (ann.certified() == 1).all()
    want = _oracle_all(seqs)
    _same(first, want, "1000 x 50 kb in one batch")
    ann.close()


def test_fuzz_slice_lone_and_small_batches_against_the_oracle(pa):
    """>= 500 contigs through the segments' launch order: 150 lone contigs (10-60 kb synthetic, and the fuzz generator's: GC 20-80 %, repeats,
    start- / stop-rich stretches) and 24 batches of 2-32.  Three runs each (sizing, steady state, replay): equal to each other and to the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu

    rng = np.random.RandomState(606)
    pool = []
    while len(pool) < 120:
        s = fuzz_gpu.make(rng).lower()
        if 2000 <= len(s) <= 60000 and set(s) <= set("acgt"):
            pool.append(s.encode())
    lone = [pa.synth_contig(880000 + i, int(rng.uniform(10000, 60000))) for i in range(90)] + pool[:60]
    batches = []
    for b in range(24):
        n = int(rng.randint(2, 33))
        batches.append([pa.synth_contig(890000 + 40 * b + k, int(rng.uniform(5000, 60000))) if rng.rand() < 0.7 else pool[int(rng.randint(len(pool)))] for k in range(n)])
    jobs = [[s] for s in lone] + batches
    flat = [s for j in jobs for s in j]
    assert len(flat) >= 500
    want = _oracle_all(flat)
    ann = pa.Annotator()
    seg_before = ann.seg_runs()
    k = 0
    for j in jobs:
        res = ann.annotate(j)
        f1 = ann.download_flat()
        ann.run(); ann.run()
        f2 = ann.download_flat()
        assert all(a.tobytes() == b.tobytes() for a, b in zip(f1, f2)), "runs of one batch differ (%d contigs)" % len(j)
        _same(res, want[k:k + len(j)], "batch of %d at contig %d" % (len(j), k))
        k += len(j)
    assert ann.seg_runs() > seg_before  # the segments took part
    ann.close()
