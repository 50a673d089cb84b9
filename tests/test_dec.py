"""csrc/phx_dec.c against the thing it restates: Python's decimal module (libmpdec at the default context, prec 28, ROUND_HALF_EVEN)
is the reference's number type (functions.py:1 `from decimal import Decimal`), so every operation the weights go through is compared
with decimal itself, text for text (coefficient AND exponent: str(weight * 1000) is the --dump text, edges.py:17-23).  CPU only."""
import ctypes as C
import random
from decimal import ROUND_HALF_EVEN, Context, Decimal as D, localcontext

import pytest

from phanotate_amd import _lib


@pytest.fixture(scope="module")
def ev():
    L = _lib.lib()
    buf = C.create_string_buffer(8192)

    def f(op, a, b="", prec=28):
        n = L.phx_dec_eval(op.encode(), str(a).encode(), str(b).encode(), prec, buf, 8192)
        return buf.value.decode() if n >= 0 else "ERR%d" % n

    return f


def _rdec(rnd, maxd=28, emin=-40, emax=10):
    nd = rnd.randint(1, maxd)
    c = rnd.randint(0, 10 ** nd - 1)
    if rnd.random() < 0.1:
        c = c // 10 ** rnd.randint(0, nd) * 10 ** rnd.randint(0, 3)
    return D("%s%dE%d" % ("-" if rnd.random() < 0.3 else "", c, rnd.randint(emin, emax)))


def test_add_sub_mul_div_round_like_decimal(ev):
    rnd = random.Random(1)
    for emin, emax in ((-40, 10), (-5, 5), (-300, 300)):
        for _ in range(6000):
            a, b = _rdec(rnd, 28, emin, emax), _rdec(rnd, 28, emin, emax)
            assert ev("add", a, b) == str(a + b), (a, b)
            assert ev("sub", a, b) == str(a - b), (a, b)
            assert ev("mul", a, b) == str(a * b), (a, b)
            if b:
                assert ev("div", a, b) == str(a / b), (a, b)
    # the shapes the reference produces (functions.py:26-46,174-178, orfs.py:122-127,162-173)
    assert ev("div", 1, "0.05") == "2E+1" and ev("div", 6, 3) == "2" and ev("div", 1, 4) == "0.25"
    assert ev("add", "1E+30", 1) == str(D("1E+30") + D(1)) and ev("sub", 1, "0.0") == "1.0"
    for n, d in ((17, 300), (0, 77), (123456, 50000 * 2), (1, 3)):
        assert ev("div", n, d) == str(D(n) / D(d))


def test_ln_exp_are_correctly_rounded(ev):
    rnd = random.Random(2)
    for prec in (28, 51):
        ctx = Context(prec=prec, rounding=ROUND_HALF_EVEN)
        for _ in range(1500):
            a = abs(_rdec(rnd, 28, -30, 3))
            if a:
                assert ev("ln", a, "", prec) == str(ctx.ln(a)), (prec, a)
            x = _rdec(rnd, 28, -30, -27 + rnd.randint(0, 2))
            assert ev("exp", x, "", prec) == str(ctx.exp(x)), (prec, x)
        for a in ("0.9999999999999999999999999999", "1.000000000000000000000000001", "0.1", "10", "1E-20", "123456789.123456789"):
            assert ev("ln", a, "", prec) == str(ctx.ln(D(a))), (prec, a)
    assert ev("ln", 1) == "0" and ev("exp", 0) == "1"


def test_pow_follows_mpd_qpow(ev):
    """Integer exponents by square-and-multiply at prec + digits + 2 (score_overlap's Decimal(length), score_gap's Decimal(100) and
    Decimal(length / 3) for lengths divisible by 3), all others as exp(b ln a) at 51 digits (pos_max / pos_min, length / 3)."""
    rnd = random.Random(3)
    for _ in range(2500):
        base = D(1) - abs(_rdec(rnd, 28, -30, -29))  # 1 - pstop
        if base <= 0:
            continue
        for e in (D(rnd.randint(1, 502)), D(100), D(rnd.randint(-2, 300) / 3), abs(_rdec(rnd, 28, -30, -28)), D(1), D(rnd.randint(1, 40)) / D(rnd.randint(41, 99))):
            assert ev("pow", base, e) == str(base ** e), (base, e)
    assert ev("pow", 1, 5) == "1" and ev("pow", 1, "0.5") == str(D(1) ** D("0.5")) and ev("pow", "1.00", 2) == "1.0000"
    assert ev("pow", "0.95", "0") == "1" and ev("pow", 2, 3) == "8" and ev("pow", "0.95", -3) == str(D("0.95") ** D(-3))


def test_floats_enter_like_decimal_and_repr(ev):
    rnd = random.Random(4)
    for _ in range(20000):
        x = rnd.choice([rnd.random(), rnd.uniform(-1e3, 1e3), rnd.randint(-2, 300) / 3, 10.0 ** rnd.randint(-20, 20) * rnd.random(), float(rnd.randint(0, 10 ** rnd.randint(1, 18)))])
        assert ev("float", repr(x)) == str(D(x)), x          # Decimal(length / 3), functions.py:43
        assert ev("repr", repr(x)) == repr(x), x              # Decimal(str(weight_rbs)), orfs.py:126
        assert ev("str", repr(x)) == str(D(repr(x))), x
    for x in (0.0, 1.0, 1e16, 1e-5, 123456789012345678.0, 0.0001, 5e-324, 1.7976931348623157e308, 0.1 + 0.2):
        assert ev("repr", repr(x)) == repr(x)


def test_truncation_and_double_double(ev):
    rnd = random.Random(5)
    for _ in range(5000):
        a = _rdec(rnd, 28, -30, 40)
        v = int(ev("trunc1000", a), 16)
        if v >> (64 * 18 - 1):
            v -= 1 << (64 * 18)
        assert v == int(a * 1000), a                          # int(weight * 1000) as fastpathz reads it, edges.py:22
    with localcontext() as ctx:
        ctx.prec = 60
        for _ in range(5000):
            a = _rdec(rnd, 28, -40, 20)
            if a:
                h, l = map(float, ev("dd", a).split())
                assert abs((D(h) + D(l) - a) / a) < D(2) ** -100, a


def test_params_from_flags_keeps_the_weights_as_written():
    """phx_params_from_flags = file_handling.py:51-66: codon:weight pairs, dict semantics for a repeated codon, weights / max as
    doubles and — for the exact arithmetic — as the texts the user wrote."""
    L = _lib.lib()
    p = _lib.Params()
    assert L.phx_params_from_flags(b"ATG:0.85,gtg:0.10,ttg:0.05,gtg:0.2", b"tag,TGA,taa", 90, C.byref(p)) == 0
    assert p.n_start == 3 and [p.start[i].value for i in range(3)] == [b"atg", b"gtg", b"ttg"]
    assert [p.start_w_text[i].value for i in range(3)] == [b"0.85", b"0.2", b"0.05"]
    assert [p.start_w[i] for i in range(3)] == [0.85 / 0.85, 0.2 / 0.85, 0.05 / 0.85]
    assert p.n_stop == 3 and p.stop[1].value == b"tga" and p.minlen == 90
    q = _lib.Params()
    L.phx_default_params(C.byref(q))
    assert bytes(q) == bytes(_defaults_via_flags(L))
    for bad in (b"atg", b"atgg:1", b"atn:1", b"atg:x", b"atg:1:2", b""):
        assert L.phx_params_from_flags(bad, None, 90, C.byref(p)) == -14, bad
    assert L.phx_params_from_flags(None, b"ta", 90, C.byref(p)) == -14 and L.phx_params_from_flags(None, None, 5, C.byref(p)) == -14


def _defaults_via_flags(L):
    p = _lib.Params()
    assert L.phx_params_from_flags(b"atg:0.85,gtg:0.10,ttg:0.05", b"tag,tga,taa", 90, C.byref(p)) == 0
    return p


def test_double_double_of_k_refine_against_60_digit_decimals():
    """csrc/phx_dd.h (k_refine's arithmetic; the same header compiles for the host): + - * / to 2^-104, exp to (1 + |x|) 2^-104 relative
    over the exponent range of the weights, ln(1 - p) to 2^-103 absolute, and repr(double) — the shortest digits that read back, the
    text Decimal(str(weight_rbs)) is built from (orfs.py:126) — identical to Python's on 100 000 doubles."""
    import math

    L = _lib.lib()

    def ev(op, a, b=(0.0, 0.0)):
        rh, rl = C.c_double(), C.c_double()
        assert L.phx_dd_eval(op.encode(), a[0], a[1], b[0], b[1], C.byref(rh), C.byref(rl)) == 0
        return D(rh.value) + D(rl.value)

    def todd(x):
        h = float(x)
        return (h, float(x - D(h)))

    rnd = random.Random(6)
    with localcontext() as ctx:
        ctx.prec = 70
        worst = {}
        for _ in range(4000):
            a = D(rnd.uniform(-10, 10)) * D(10) ** rnd.randint(-5, 5) + D(rnd.random()) * D(10) ** -20
            b = D(rnd.uniform(-10, 10)) * D(10) ** rnd.randint(-5, 5) + D(rnd.random()) * D(10) ** -21
            da, db = todd(a), todd(b)
            A, B = D(da[0]) + D(da[1]), D(db[0]) + D(db[1])
            for op, ref in (("add", A + B), ("sub", A - B), ("mul", A * B), ("div", A / B)):
                worst[op] = max(worst.get(op, 0), abs((ev(op, da, db) - ref) / ref))
            x = D(rnd.uniform(-2, 690)) + D(rnd.random()) * D(10) ** -18
            dx = todd(x)
            X = D(dx[0]) + D(dx[1])
            worst["exp"] = max(worst.get("exp", 0), abs((ev("exp", dx) - X.exp()) / X.exp()) / (1 + abs(X)))
            p = D(rnd.uniform(1e-9, 0.149))
            db1 = todd(D(1) - p)
            B1 = D(db1[0]) + D(db1[1])
            worst["log"] = max(worst.get("log", 0), abs(ev("log", db1) - B1.ln()))
        for op in ("add", "sub", "mul", "div", "exp"):
            assert worst[op] < D(2) ** -104, (op, worst[op])
        assert worst["log"] < D(2) ** -103, worst["log"]
        bad = 0
        for _ in range(100000):
            x = rnd.choice([rnd.random(), rnd.uniform(0, 1e3), 10.0 ** rnd.uniform(-10, 10), (1 + rnd.randint(0, 5000)) / (28 + rnd.randint(1, 5000)) / ((1 + rnd.randint(0, 100000)) / (28 + rnd.randint(50000, 100000))),
                            2.0 ** rnd.randint(-30, 30), float(rnd.randint(1, 10 ** rnd.randint(1, 10)))])
            if not (1e-10 <= x <= 1e10):
                continue
            dg, e = C.c_uint64(), C.c_int32()
            nd = L.phx_dd_shortest(x, C.byref(dg), C.byref(e))
            ref = D(repr(x))
            assert nd > 0 and D(dg.value) * D(10) ** e.value == ref and len(str(dg.value)) == nd, (repr(x), dg.value, e.value)
            assert abs((ev("repr", (x, 0.0)) - ref) / ref) < D(2) ** -102
        assert L.phx_dd_shortest(1e-11, C.byref(dg), C.byref(e)) == 0 and L.phx_dd_shortest(1e11, C.byref(dg), C.byref(e)) == 0


def test_number_text_does_not_follow_the_host_locale(ev):
    """repr(float) / the -s weights went through snprintf("%.*e") and strtod, which follow LC_NUMERIC: inside a host
    application with a comma-decimal locale every RBS weight of the host re-solve read as 0.  They now use the "C" locale whatever
    the process has set.  (Skipped where no such locale is installed — this image has only C / POSIX.)"""
    import locale

    old = locale.setlocale(locale.LC_NUMERIC)
    for name in ("de_DE.UTF-8", "de_DE.utf8", "fr_FR.UTF-8", "fr_FR.utf8", "nl_NL.UTF-8", "ru_RU.UTF-8", "de_DE", "fr_FR"):
        try:
            locale.setlocale(locale.LC_NUMERIC, name)
            break
        except locale.Error:
            continue
    else:
        pytest.skip("no comma-decimal locale installed")
    try:
        assert locale.localeconv()["decimal_point"] == ","
        for x in (1.5, 0.1 + 0.2, 2.5e-7, 123456.789):
            assert ev("repr", repr(x)) == repr(x)
            assert ev("float", repr(x)) == str(D(x))
        L = _lib.lib()
        from phanotate_amd._lib import phx_params  # noqa: F401

        p = phx_params()
        assert L.phx_params_from_flags(b"atg:0.85,gtg:0.10,ttg:0.05", b"tag,tga,taa", 90, C.byref(p)) == 0
        assert abs(p.start_w[1] - 0.10 / 0.85) < 1e-15
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)
