"""k_front (phx_front.inc): the front end of a small batch (ORF count ... edge fill, functions.py:143-454) as one launch with grid barriers.
Its safety net — a workgroup that waits too long at a barrier raises DTotals.front_abort, the host repeats the run with the staged kernels
and keeps them on the context — is forced here (PHX_FRONT_SPINS=-1: the first wait at a barrier counts as too long), and the kernel is run
under contention: six processes share the GPU, every run goes through k_front."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import phanotate_amd

    return phanotate_amd


def _ann(pa, flags=(), env=None):
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        return pa.Annotator(flags=flags)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _records(a, n):
    flat = a.download_flat()
    taps = []
    for i in range(n):
        gl = a.globals(i)
        taps.append((gl.n_orf, gl.n_node, gl.n_edge, a.orfs(i).tobytes(), a.nodes(i).tobytes(), a.edges(i).tobytes()))
    return [x.tobytes() for x in flat], taps


def test_a_stalled_grid_barrier_repeats_the_run_with_the_staged_kernels(pa):
    """PHX_FRONT_SPINS=-1: the first workgroup that has to wait at a grid barrier gives up.  The run that was to be the first fused one is
    repeated with the staged kernels inside phx_run, delivers the records of a PHX_CREATE_NO_FUSE context byte for byte, and the context
    never launches k_front again (phx_front_runs reports -(runs) - 1 = -1 from then on)."""
    seqs = [pa.synth_contig(900 + i, 9000 + 1500 * i) for i in range(3)]  # three workgroups per barrier at least: some workgroup waits
    want = None
    for mode in ("staged", "stalled", "fused"):
        a = _ann(pa, ("no_fuse",) if mode == "staged" else (), {"PHX_FRONT_SPINS": "-1"} if mode == "stalled" else None)
        a.upload(seqs)
        got = []
        for r in range(4):
            a.run()
            got.append(_records(a, len(seqs)))
            if mode == "stalled":
                assert a.front_runs() == (0 if r == 0 else -1), (r, a.front_runs())  # run 0 sizes the buffers (staged); run 1 meets the stall
        if mode == "fused":
            assert a.front_runs() == 3
        if mode == "staged":
            assert a.front_runs() == 0
            want = got[0]
        for g in got:
            assert g == want, mode
        a.close()


FRONT_WORKER = """
import os, sys, time
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import phanotate_amd as pa
from conftest import load_golden
rank = int(sys.argv[1])
seq = load_golden("phiX174")[2] if rank %% 2 == 0 else pa.synth_contig(5000 + rank, 30000)
ref = pa.Annotator(flags=("no_fuse",))
want = ref.annotate_flat([seq])
ref.close()
ann = pa.Annotator()
first = ann.annotate_flat([seq])
assert all(a.tobytes() == b.tobytes() for a, b in zip(first, want)), "rank %%d: first run differs" %% rank
worst = 0.0
for r in range(150):
    t0 = time.perf_counter()
    ann.run()
    worst = max(worst, (time.perf_counter() - t0) * 1e3)
    if r %% 10 == 0:
        got = ann.download_flat()
        assert all(a.tobytes() == b.tobytes() for a, b in zip(got, want)), "rank %%d run %%d differs" %% (rank, r)
print("FRONT_OK", rank, ann.front_runs(), "%%.2f" %% worst, flush=True)
"""


def test_six_processes_run_small_batches_through_k_front(tmp_path):
    """Six processes, each running phiX174 or a 30 kb contig alone 150 times: every steady-state run is ONE k_front launch whose grid barriers
    need all of its workgroups resident while five other processes hold CUs.  Every downloaded result equals the staged kernels'; a context
    may fall back to the staged kernels (front_runs < 0), a run may not take long."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "front_worker.py"
    script.write_text(FRONT_WORKER % (root, root))
    procs = [subprocess.Popen([sys.executable, str(script), str(k)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(6)]
    outs = [p.communicate(timeout=900) for p in procs]
    lines = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-2000:]
        lines += [l.split() for l in so.splitlines() if l.startswith("FRONT_OK")]
    assert len(lines) == 6
    print("fused runs (negative: fell back to the staged kernels), worst run (ms):", [(int(l[2]), float(l[3])) for l in lines])
    assert all(int(l[2]) != 0 for l in lines)  # k_front ran (or stalled and was replaced) in every process
    assert max(float(l[3]) for l in lines) < 400.0
