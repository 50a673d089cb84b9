"""The bit-sliced feature code of k_features (phanotate_amd/csrc/phx_feat_core.h), compiled for the host over a 64-lane array type
(tests/feat_core_host.cpp), against the oracle: per-position RBS bins, background histogram, codon classes, GC-frame classes, g+c.
The GPU runs the same source with V = uint32_t and DPP for the neighbouring lanes; this is the CPU-side check of its logic.
Also: the host packer of phx_upload (letters -> residue-split bit planes) against a numpy statement of the layout."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CODE = {ord("a"): 0, ord("c"): 1, ord("t"): 2, ord("g"): 3}
AMB_G = b"sbv"
AMB_A = b"nrywkmdh"


def codes_of(seq):
    """per letter: b0 | b1 << 1 | amb << 2 (a letter outside the alphabet: 6)"""
    lut = np.full(256, 6, np.uint8)
    for ch, c in CODE.items():
        lut[ch] = c
        lut[ch - 32] = c
    for ch in AMB_G:
        lut[ch] = 7
        lut[ch - 32] = 7
    for ch in AMB_A:
        lut[ch] = 4
        lut[ch - 32] = 4
    return lut[np.frombuffer(seq, np.uint8)]


def pack_records(seq, nrec):
    """records [nrec + 2][stream][b0, b1, amb] of 32-bit words; record 0 and nrec + 1 are pads, positions >= L read as (0, 0, 1)"""
    L = len(seq)
    n = nrec * 96
    c = np.full(n, 4, np.uint8)
    c[:L] = codes_of(seq)
    rec = np.zeros((nrec + 2, 3, 3), np.uint32)
    rec[0, :, 2] = 0xFFFFFFFF
    rec[nrec + 1, :, 2] = 0xFFFFFFFF
    c = c.reshape(nrec, 32, 3)  # [record][bit][stream]
    w = (1 << np.arange(32, dtype=np.uint64))
    for r in range(3):
        for pl in range(3):
            bits = ((c[:, :, r] >> pl) & 1).astype(np.uint64)
            rec[1:-1, r, pl] = (bits * w).sum(1).astype(np.uint32)
    return rec


@pytest.fixture(scope="module")
def host():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "feat_core_host.so")
    src = os.path.join(HERE, "feat_core_host.cpp")
    hdr = os.path.join(ROOT, "phanotate_amd", "csrc", "phx_feat_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    lib.feat_core_host.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rbs_bin_linear_host.argtypes = [C.c_void_p, C.c_int]
    return lib


def codon_sets(starts, stops):
    comp = {"a": "t", "c": "g", "g": "c", "t": "a"}
    sets = [0, 0, 0, 0]
    for c0 in "actg":
        for c1 in "actg":
            for c2 in "actg":
                cod = c0 + c1 + c2
                rc = comp[c2] + comp[c1] + comp[c0]
                cl = 1 if cod in starts else 2 if rc in starts else 3 if cod in stops else 4 if rc in stops else 0
                if cl:
                    sets[cl - 1] |= 1 << ("actg".index(c0) | "actg".index(c1) << 2 | "actg".index(c2) << 4)
    return np.array(sets, np.uint64)


def run_core(lib, seq, starts=("atg", "gtg", "ttg"), stops=("tag", "tga", "taa"), defcod=False, tap=True):
    L = len(seq)
    nrec = (L + 95) // 96 + 1
    rec = np.ascontiguousarray(pack_records(seq, nrec))
    planes = np.zeros((12, 3, nrec), np.uint32)
    cnt = np.zeros(28, np.uint32)
    gc = np.zeros(1, np.uint32)
    bad = np.zeros(1, np.uint32)
    tp = np.zeros((nrec, 2, 3, 5), np.uint32)
    sets = codon_sets(starts, stops)
    lib.feat_core_host(rec.ctypes.data, nrec, L, int(defcod), sets.ctypes.data, planes.ctypes.data, cnt.ctypes.data, gc.ctypes.data, bad.ctypes.data,
                       tp.ctypes.data if tap else None)
    return planes, cnt, int(gc[0]), int(bad[0]), tp


def unbits(words):
    """[..., nrec] words -> [..., nrec * 32] bits"""
    return ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(words.shape[:-1] + (-1,)).astype(np.uint8)


def per_position(planes3):
    """[3][nrec] (bit k of frame / stream f <-> position f + 3 k) -> per position"""
    b = unbits(planes3)  # [3][K]
    return b.T.reshape(-1)  # position 3 k + f


def _mx(a, b, c):
    return np.where(a > b, np.where(a > c, 1, 3), np.where(b > c, 2, 3))


def _mn(a, b, c):
    return np.where(a > b, np.where(b > c, 3, 2), np.where(a > c, 3, 1))


def check(lib, seq, oracle, starts=("atg", "gtg", "ttg"), stops=("tag", "tga", "taa"), defcod=False, params=None):
    o = oracle.run(seq, params=params, stages=1)
    assert o["status"] == 0
    L = len(seq)
    planes, cnt, gc, bad, tp = run_core(lib, seq, starts, stops, defcod)
    assert bad == 0
    # RBS bins of every window, from the tap planes
    binF = sum(per_position(tp[:, 0, :, b].T) << b for b in range(5))[:L]
    binR = sum(per_position(tp[:, 1, :, b].T) << b for b in range(5))[:L]
    assert np.array_equal(binF[20:], o["binF"][20:])
    assert np.array_equal(binR, o["binR"])
    # background histogram without the 20 right-truncated forward windows (k_orf adds those: tail windows)
    want = np.bincount(o["binR"], minlength=28) + np.bincount(o["binF"][20:], minlength=28)
    assert np.array_equal(cnt, want)
    # codon classes
    from test_gpu_parity import codon_classes
    cls = np.zeros(L, np.uint8)
    for c in range(4):
        cls = np.where(per_position(planes[c])[:L] == 1, c + 1, cls)
    assert np.array_equal(cls, codon_classes(seq, starts, stops))
    s = seq.decode().lower()
    atgF = per_position(planes[10])[:L]
    atgR = per_position(planes[11])[:L]
    assert np.array_equal(np.flatnonzero(atgF), np.array([p for p in range(L - 2) if s[p:p + 3] == "atg"], np.int64))
    assert np.array_equal(np.flatnonzero(atgR), np.array([p for p in range(L - 2) if s[p:p + 3] == "cat"], np.int64))
    # GC frame classes
    gcf = o["gc_pos_freq"][1:].astype(int)
    A, B, Cc = (per_position(planes[4 + i])[: len(gcf)] for i in range(3))
    assert np.array_equal(A, (gcf[:, 0] > gcf[:, 1]).astype(np.uint8))
    assert np.array_equal(B, (gcf[:, 1] > gcf[:, 2]).astype(np.uint8))
    assert np.array_equal(Cc, (gcf[:, 0] > gcf[:, 2]).astype(np.uint8))
    A, B, Cc = (per_position(planes[7 + i])[: len(gcf)] for i in range(3))
    assert np.array_equal(A, (gcf[:, 2] > gcf[:, 1]).astype(np.uint8))
    assert np.array_equal(B, (gcf[:, 1] > gcf[:, 0]).astype(np.uint8))
    assert np.array_equal(Cc, (gcf[:, 2] > gcf[:, 0]).astype(np.uint8))
    c = codes_of(seq)
    assert gc == int((c & 1).sum())
    # without the tap planes the counters are the same
    _, cnt2, gc2, _, _ = run_core(lib, seq, starts, stops, defcod, tap=False)
    assert np.array_equal(cnt2, cnt) and gc2 == gc


def test_bit_sliced_features_equal_the_oracle(host, oracle):
    import phanotate_amd as pa

    for seed, L in ((1, 4000), (2, 6143), (3, 96 * 62), (4, 96 * 62 + 1), (5, 25000), (6, 97), (7, 21), (8, 200)):
        seq = pa.synth_contig(seed, L)
        check(host, seq, oracle)
        check(host, seq, oracle, defcod=True)


def test_bit_sliced_features_on_ambiguity_codes_and_other_codon_tables(host, oracle):
    import phanotate_amd as pa
    from oracle import oracle as om

    rnd = np.random.RandomState(5)
    seq = bytearray(pa.synth_contig(11, 9000))
    for p in rnd.randint(0, len(seq), 300):
        seq[p] = rnd.choice(list(b"nryswkmbvdhNRS"))
    seq[-40:] = b"N" * 40
    seq[:3] = b"SBV"
    check(host, bytes(seq), oracle)
    check(host, bytes(seq).upper(), oracle, defcod=True)
    starts, stops = ("atg", "ctg", "ata", "gtg"), ("tag", "taa")
    p = om.make_params("atg:0.5,ctg:0.2,ata:0.2,gtg:0.1", "tag,taa", 90)
    check(host, bytes(seq), oracle, starts, stops, params=p)
    _, _, _, bad, _ = run_core(host, b"acgtxacgt" * 30)
    assert bad == 1


def test_score_rbs_on_linear_masks_equals_the_rule_chain(host):
    from oracle import oracle as om

    rnd = np.random.RandomState(0)
    motifs = [b"ggagga", b"ggagg", b"gagga", b"ggacga", b"ggcgga", b"ggag", b"agga", b"ggtgg", b"agg", b"gaaga", b"gga", b"gag"]
    for it in range(30000):
        n = int(rnd.choice([21, 21, 21, 20, 17, 9, 5, 3, 1]))
        w = bytearray(rnd.choice(list(b"acgt"), n).tobytes())
        if it % 2:
            m = motifs[rnd.randint(len(motifs))]
            o = rnd.randint(0, 18)
            w[o:o + len(m)] = m
            w = w[:n]
        if it % 7 == 0 and n > 2:
            w[rnd.randint(n)] = ord("n")
        # score_rbs reverses its argument: s = seq[::-1]
        s = bytes(w)
        codes = np.array([CODE.get(ch, 4) for ch in s], np.uint8)
        got = host.rbs_bin_linear_host(codes.ctypes.data, len(codes))
        assert got == om.score_rbs(s[::-1]), (s, got)
