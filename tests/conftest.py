import gzip
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def read_fasta_gz(path):
    with gzip.open(path, "rt") as f:
        lines = f.read().split("\n")
    return lines[0][1:].split()[0], "".join(lines[1:])


def golden_cases():
    return sorted(fn[:-4] for fn in os.listdir(GOLDEN) if fn.endswith(".npz"))


def load_golden(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    name, seq = read_fasta_gz(os.path.join(GOLDEN, case + ".fasta.gz"))
    return g, name, seq


def golden_trnas(g):
    """The tRNA hit list of a fixture as functions.add_trnas holds it ([(start, stop)], reversed for complement hits), or None."""
    if "trna_start" not in g:
        return None
    return list(zip(g["trna_start"].tolist(), g["trna_stop"].tolist()))


def golden_params(g):
    """(start_codons, stop_codons, minlen) strings as the CLI would take them."""
    return dict(start_codons=str(g["params_start"]), stop_codons=str(g["params_stop"]), minlen=int(g["params_minlen"]))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


# ---- the solver boundary, restated for the tests (the golden generator's rule, tests/golden/make_golden.py:100-133) ----
def inorder_bellman_ford(V, edges, s):
    """In-place Bellman-Ford over `edges` = [(u, v, w int)] in the given order with a strict '<' (what the reference feeds
    fastpathz: Graph.iteredges order, phanotate.py:56-64).  Returns (dist, parent edge index per node) or (None, None)
    when it does not settle in V rounds."""
    dist = [None] * V
    par = [-1] * V
    dist[s] = 0
    for _ in range(V + 1):
        ch = False
        for i, (u, v, w) in enumerate(edges):
            du = dist[u]
            if du is None:
                continue
            nd = du + w
            if dist[v] is None or nd < dist[v]:
                dist[v] = nd
                par[v] = i
                ch = True
        if not ch:
            return dist, par
    return None, None


def exact_dist_from_device_edges(ann, i):
    """Exact distances (python ints) over the graph libphx built for contig i, from ITS OWN fp64 weights
    (phx_tap_edges): weight = trunc(w*1000), edges.py:22.  The device's 128..1088-bit sums must equal these bit for bit."""
    import math

    ed = ann.edges(i)
    V = int(ann.globals(i).n_node)
    src = ed["src"].tolist()
    dst = ed["dst"].tolist()
    w = [int(math.trunc(float(x) * 1000.0)) for x in ed["w"]]
    dist = [None] * V
    dist[V - 2] = 0
    for _ in range(V + 1):  # edges are grouped by destination in position order: a handful of sweeps
        ch = False
        for k in range(len(src)):
            du = dist[src[k]]
            if du is not None and (dist[dst[k]] is None or du + w[k] < dist[dst[k]]):
                dist[dst[k]] = du + w[k]
                ch = True
        if not ch:
            break
    return dist
