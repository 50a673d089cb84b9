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


def golden_params(g):
    """(start_codons, stop_codons, minlen) strings as the CLI would take them."""
    return dict(start_codons=str(g["params_start"]), stop_codons=str(g["params_stop"]), minlen=int(g["params_minlen"]))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc
