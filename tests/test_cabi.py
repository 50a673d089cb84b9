"""CPU-side checks of the C-ABI library and the host logic (no compute calls: there is no GPU here)."""
import ctypes as C
import gzip
import hashlib
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, read_fasta_gz

import phanotate_amd as pa
from phanotate_amd import _lib


def header_functions():
    txt = open(os.path.join(ROOT, "include", "phx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(phx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.SO)
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "include/phx.h declares %s but libphx.so does not export it" % n
    assert set(_lib.EXPORTS) == set(names)


def test_version_and_errors():
    lib = _lib.lib()
    assert lib.phx_version() == 400
    assert b"no CPU path" in lib.phx_strerror(-10)
    assert lib.phx_strerror(-2)


def test_no_cpu_fallback():
    """Without a HIP device the product must fail loudly, never compute on the host."""
    lib = _lib.lib()
    if lib.phx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(pa.PhxError) as e:
        pa.Annotator()
    assert e.value.code == -10


def test_bad_params_rejected():
    lib = _lib.lib()
    h = C.c_void_p()
    bads = [pa.make_params()]
    bads[0].minlen = 3
    for field, text in (("start", b"anx"), ("stop", b"ta"), ("start", b"atgc")):  # a bad letter, a short codon, a 4-letter codon without NUL
        q = pa.make_params()
        C.memmove(C.addressof(getattr(q, field)[0]), text + b"\0" * (4 - len(text)), 4)
        bads.append(q)
    for bad in bads:
        assert lib.phx_create(C.byref(bad), 0, None, C.byref(h)) == -14
        assert lib.phx_create_ex(C.byref(bad), 0, None, 1, C.byref(h)) == -14
    assert lib.phx_create_ex(C.byref(pa.make_params()), 0, None, 1 << 20, C.byref(h)) == -1  # unknown flag
    # the binding refuses them before they reach the library (the reference would keep such codons as keys that never match)
    for kw in (dict(start_codons="anx:1"), dict(stop_codons="ta"), dict(start_codons="atgc:1"), dict(start_codons="atg"), dict(minlen=3)):
        with pytest.raises(ValueError):
            pa.make_params(**kw)


def test_default_params_match_reference_flags():
    p = _lib.Params()
    _lib.lib().phx_default_params(C.byref(p))
    q = pa.make_params()
    assert p.minlen == q.minlen == 90
    assert [bytes(p.start[i].value) for i in range(3)] == [b"atg", b"gtg", b"ttg"]
    assert [bytes(p.stop[i].value) for i in range(3)] == [b"tag", b"tga", b"taa"]
    assert list(p.start_w[:3]) == list(q.start_w[:3]) == [1.0, 0.10 / 0.85, 0.05 / 0.85]


def test_synth_generator_is_pinned():
    """The committed golden FASTA of seed 0..4 must be reproduced bit for bit."""
    for s in range(5):
        name, seq = read_fasta_gz(os.path.join(GOLDEN, "synth50k_%d.fasta.gz" % s))
        assert pa.synth_contig(s, 50000) == seq.encode()
    a = pa.synth_contig(7, 1000)
    assert a == pa.synth_contig(7, 50000)[:1000] or len(a) == 1000
    assert set(a) <= set(b"acgt")


def test_rbs_motif_table_equals_reference_chain(oracle):
    """The position kernel scores a 21-mer as max over (offset, k-mer) table hits; the reference is an
    if/elif chain (functions.py:48-138).  Check the host-built tables against the oracle's chain."""
    t6 = np.zeros(4096, np.uint32); t5 = np.zeros(1024, np.uint32); t4 = np.zeros(256, np.uint32); t3 = np.zeros(64, np.uint32)
    _lib.lib().phx_rbs_table(*[t.ctypes.data_as(C.c_void_p) for t in (t6, t5, t4, t3)])
    tabs = {6: t6, 5: t5, 4: t4, 3: t3}
    code = {"a": 0, "c": 1, "t": 2, "g": 3}
    cls_of = lambda o: 0 if o <= 4 else 1 if o <= 10 else 2 if o <= 12 else 3
    rng = np.random.RandomState(3)

    def table_score(seq):
        s = seq[::-1]
        best = 0
        for o in range(3, 16):
            v = min(6, len(s) - o)
            if v < 3:
                break
            kc = sum(code[s[o + k]] << (2 * k) for k in range(v))
            best = max(best, (int(tabs[v][kc]) >> (8 * cls_of(o))) & 0xFF)
        return best

    seen = set()
    for it in range(6000):
        n = 21 if it % 4 else rng.randint(1, 22)
        if it % 3 == 0:  # plant a purine-rich core so that high bins are exercised
            core = "".join(rng.choice(list("ag"), rng.randint(3, 9), p=[0.35, 0.65]))
            pad = "".join(rng.choice(list("acgt"), 21))
            k = rng.randint(0, 15)
            seq = (pad[:k] + core + pad)[:n]
        else:
            seq = "".join(rng.choice(list("acgt"), n))
        want = oracle.score_rbs(seq)
        assert table_score(seq) == want, seq
        seen.add(want)
    assert len(seen) >= 20, "test windows reached only bins %s" % sorted(seen)


def test_partition_is_balanced_and_complete():
    from phanotate_amd.shard import partition

    rng = np.random.RandomState(0)
    lens = list(rng.randint(1000, 200000, 97))
    for world in (1, 2, 3, 8):
        parts = partition(lens, world)
        assert sorted(i for p in parts for i in p) == list(range(97))
        loads = [sum(lens[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(lens)
    assert partition([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]
