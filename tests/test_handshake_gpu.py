"""The planner / solver hand-shake of small batches (k_sssp_wave<2,0,true> is launched beside k_wave_plan<2,0> and follows
its progress counter; a solver that sees no progress for ~20 ms hands its contig to the workgroup kernel) under contention:
two contexts in flight on one GPU, and eight processes sharing it.  The results must be what an undisturbed context gives; the
time-outs are counted (phx_plan_timeouts) and a context that saw one goes back to launching the solver behind its planner — so a
stall can happen at most once per context, and not at all in the cases below unless the counter says so."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _batch(n, seed0):
    import phanotate_amd as pa

    rng = np.random.RandomState(seed0)
    return [pa.synth_contig(seed0 + i, int(rng.choice([3000, 20000, 50000, 90000]))) for i in range(n)]


@pytest.mark.parametrize("n", [8, 300, 800])
def test_two_contexts_in_flight_follow_their_planners(n):
    import phanotate_amd as pa

    seqs = [_batch(n, 9000), _batch(n, 29000)]
    ref = []
    solo_ms = []
    for s in seqs:  # each batch alone: the records to compare with, and the time of an undisturbed run
        a = pa.Annotator(device=0)
        ref.append(a.annotate_flat(s))
        for _ in range(3):
            a.run()
        t0 = time.perf_counter()
        for _ in range(10):
            a.run()
        solo_ms.append((time.perf_counter() - t0) / 10 * 1e3)
        assert a.plan_timeouts() == 0
        a.close()
    pipe = pa.Pipeline(device=0, depth=2)
    anns = pipe.anns
    for a, s in zip(anns, seqs):
        a.annotate_flat(s)
    worst = 0.0
    runs = 60
    for r in range(-1, runs):  # (round -1 is not timed: the first asynchronous run of a context captures its graph and loads kernels no synchronous run has used: ~7 ms)
        t0 = time.perf_counter()
        for a in anns:
            a.run_async()
        for k, a in enumerate(anns):
            a.wait()
            got = a.download_flat(exact=False)
            assert all(x.tobytes() == y.tobytes() for x, y in zip(got, ref[k])), (n, r, k)
        if r >= 0:
            worst = max(worst, (time.perf_counter() - t0) * 1e3)
    to = [a.plan_timeouts() for a in anns]
    # no solver may have run into its 20 ms time-out; and no round of the two runs + downloads may have taken anywhere near it
    assert to == [0, 0], "planner time-outs with two contexts in flight: %r (n = %d)" % (to, n)
    assert worst < sum(solo_ms) + 12.0, "a round of two runs took %.2f ms (alone: %.2f + %.2f ms)" % (worst, solo_ms[0], solo_ms[1])
    for a in anns:
        for i in (0, n // 2, n - 1):
            assert a.globals(i).sssp_handed_back != 5
    pipe.close()


WORKER = """
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import phanotate_amd as pa
rank = int(sys.argv[1])
seqs = [pa.synth_contig(40000 + 500 * rank + i, 50000) for i in range(500)]
ann = pa.Annotator(device=0)
first = ann.annotate_flat(seqs)
worst = 0.0
for r in range(25):
    t0 = time.perf_counter()
    ann.run()
    worst = max(worst, (time.perf_counter() - t0) * 1e3)
    got = ann.download_flat(exact=False)
    assert all(a.tobytes() == b.tobytes() for a, b in zip(got, first)), "rank %%d run %%d differs from the first" %% (rank, r)
print("HANDSHAKE_OK", rank, ann.plan_timeouts(), "%%.2f" %% worst, flush=True)
"""


def test_eight_processes_of_500_contigs_on_one_gpu(tmp_path):
    """Eight processes x 500 contigs on one GPU: 4000 planner + 4000 solver wavefronts compete for 1024 SIMDs, so a solver may well
    start before its planner is resident.  What must hold: every run's records equal the first run's; a process sees at most ONE run
    with time-outs (the context then stops launching the solver beside the planner), so no process stalls twice."""
    script = tmp_path / "hs_worker.py"
    script.write_text(WORKER % ROOT)
    procs = [subprocess.Popen([sys.executable, str(script), str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(8)]
    outs = [p.communicate(timeout=900) for p in procs]
    lines = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-2000:]
        lines += [l.split() for l in so.splitlines() if l.startswith("HANDSHAKE_OK")]
    assert len(lines) == 8
    timeouts = [int(l[2]) for l in lines]
    worst = [float(l[3]) for l in lines]
    print("time-outs per process:", timeouts, "worst run (ms):", worst)
    # a time-out costs ~28 ms once; 25 runs of 500 contigs with 8 processes taking turns stay far below 25 such stalls
    assert max(worst) < 400.0
