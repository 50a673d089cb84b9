"""k_sssp_duo (phx_sssp_duo.inc): the 128-bit contigs' shortest path (fastpathz, phanotate.py:56-64) by a feeder and a solver wavefront
per contig, against k_sssp_wave<2> (PHX_CREATE_NO_DUO: one wavefront per contig, the kernel of rounds 2-4) and against exact python-int
distances.  Bar: every record byte-equal, every distance bit-equal."""
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


def _both(pa, seqs, runs=1, sample=()):
    out = {}
    for flags in ((), ("no_duo",)):
        a = pa.Annotator(flags=flags)
        flat = a.annotate_flat(seqs)
        for _ in range(runs - 1):
            a.run()
        if runs > 1:
            flat = a.download_flat()
        kern = [a.globals(i).sssp_kernel for i in range(len(seqs))]
        back = [a.globals(i).sssp_handed_back for i in range(len(seqs))]
        dist = {i: a.dist(i) for i in sample}
        for i in sample:
            _exact(a, i)
        out[flags] = (flat, kern, back, dist)
        a.close()
    d, w = out[()], out[("no_duo",)]
    for x, y in zip(d[0], w[0]):
        assert x.tobytes() == y.tobytes()
    for i in sample:
        assert list(d[3][i]) == list(w[3][i]), i
    return d, w


def test_streamed_batch_equals_the_one_wavefront_kernel(pa):
    """120 of the benchmark's 50 kb contigs (a batch of up to 800 contigs: both kernels follow the planner's progress counter; step-backs —
    a rejected pack — occur about once per contig), three runs each (the third replays the captured graph)."""
    seqs = [pa.synth_contig(i, 50000) for i in range(3, 1000, 8)][:120]
    d, w = _both(pa, seqs, runs=3, sample=(0, 17, 63, 119))
    assert all(k == 2 for k in d[1]) and not any(d[2])


def test_large_batch_of_short_contigs_not_streamed(pa):
    """900 contigs (beyond PHX_PLAN_STREAM_MAX: the solver is launched behind its planner) of 3-9 kb."""
    rng = np.random.RandomState(5)
    seqs = [pa.synth_contig(7000 + i, int(rng.randint(3000, 9000))) for i in range(900)]
    d, w = _both(pa, seqs, runs=2, sample=(0, 450, 899))
    assert sum(k == 2 for k in d[1]) > 850


def test_folded_sources_side_list_and_spill(pa):
    """GC-rich contigs with a 9-10 kb reading frame (test_gpu_parity.py::test_gc_rich_long_orf...): ORF weights beyond 2^51 (the side list),
    start nodes further back than the distance ring holds (folded in from global memory by the feeder), a stop node with hundreds of
    in-edges (spill list; the tight configuration cannot hold it: the roomy one, which stays with k_sssp_wave<2, 1>, or a hand-back);
    and ordinary contigs around them.  Whatever kernel ends up with a contig: exact distances, equal records."""
    rng = np.random.RandomState(7)

    def gc_rich(n):
        return "".join(rng.choice(list("acgt"), n, p=[0.1, 0.4, 0.4, 0.1]))

    sense = [a + b + c for a in "cg" for b in "acgt" for c in "cg" if a + b + c != "gtg"]
    seqs = []
    for ncod, p in ((1200, 0.02), (3000, 0.03), (3000, 0.12), (2500, 0.06)):
        body = "".join("gtg" if rng.rand() < p else sense[rng.randint(len(sense))] for _ in range(ncod))
        seqs.append(gc_rich(6000) + "atg" + body + "taa" + gc_rich(6000))
    seqs += [pa.synth_contig(40 + i, 30000).decode() for i in range(4)]
    d, w = _both(pa, seqs, runs=2, sample=tuple(range(len(seqs))))
    assert 2 in d[1]


def test_lone_genomes(pa):
    """Lambda and T4 alone in a batch: the feeder really waits on the planner's counter (k_sssp_duo<0, true>)."""
    for name in ("NC_001416.1", "NC_000866.1"):
        g, nm, seq = load_golden(name)
        d, w = _both(pa, [seq], runs=3, sample=(0,))
        assert d[1] == [2]
