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
    assert lib.phx_version() == 410
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


def test_host_packer_writes_the_record_layout():
    """phx_upload's staging pass (phx_pack_planes, phx_host.c: AVX2 + pext, scalar tail) against the numpy statement of the layout the
    kernels read (tests/test_feat_core_host.py::pack_records): residue-split bit planes, 36 bytes per 96 bases."""
    from test_feat_core_host import pack_records

    L_ = _lib.lib()
    L_.phx_pack_planes.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
    L_.phx_pack_planes.restype = None
    rnd = np.random.RandomState(9)
    for n in (0, 1, 2, 3, 95, 96, 97, 191, 192, 1000, 96 * 40, 96 * 40 + 17, 12345):
        seq = bytes(rnd.choice(list(b"acgtACGTnryswkmbvdhNRSxX-"), n, p=[0.2] * 4 + [0.02] * 4 + [0.12 / 17] * 17).astype(np.uint8))
        nrec = (n + 95) // 96 + 2
        got = np.full((nrec, 3, 3), 0xDEADBEEF, np.uint32)
        L_.phx_pack_planes(seq, n, got.ctypes.data, nrec)
        want = pack_records(seq, nrec)[1:-1]
        assert np.array_equal(got, want), n


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
