// phx_dd.h — double-double arithmetic (~106 bits) and the shortest decimal of a double, for k_refine (phx_refine.inc): the edge weights
// the fp64 pipeline cannot place between two integers are evaluated again to ~1e-30 relative, so that what is left between the device's
// integer and the reference's trunc(Decimal(w) * 1000) is the rounding of the reference's OWN 28-digit chain.
// Compiles for the device (hipcc) and for the host (the CPU tests call it through phx_dd_eval): no HIP-only constructs.
// Error-free transformations after Dekker / Knuth; exp and log after the QD library's scheme (argument reduction by ln 2 and 2^-9,
// Taylor for expm1, nine doublings; one Newton step on exp for log).  Needs FMA and no contraction of the other operations
// (-ffp-contract=off, as the whole library is built).
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define DD_FN __host__ __device__ __forceinline__
#else
#define DD_FN static inline
#endif

struct dd_t { double hi, lo; };

DD_FN dd_t dd_make(double h, double l) { dd_t r; r.hi = h; r.lo = l; return r; }
DD_FN dd_t dd_two_sum(double a, double b) { const double s = a + b, bb = s - a; return dd_make(s, (a - (s - bb)) + (b - bb)); }
DD_FN dd_t dd_quick_two_sum(double a, double b) { const double s = a + b; return dd_make(s, b - (s - a)); } // |a| >= |b|
DD_FN dd_t dd_two_prod(double a, double b) { const double p = a * b; return dd_make(p, fma(a, b, -p)); }
DD_FN dd_t dd_from(double a) { return dd_make(a, 0.0); }
DD_FN dd_t dd_neg(dd_t a) { return dd_make(-a.hi, -a.lo); }
DD_FN dd_t dd_add(dd_t a, dd_t b) { // (IEEE-style: accurate to 2^-105 relative also under cancellation)
    dd_t s = dd_two_sum(a.hi, b.hi), t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi; s = dd_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo; return dd_quick_two_sum(s.hi, s.lo);
}
DD_FN dd_t dd_sub(dd_t a, dd_t b) { return dd_add(a, dd_neg(b)); }
DD_FN dd_t dd_add_d(dd_t a, double b) { dd_t s = dd_two_sum(a.hi, b); s.lo += a.lo; return dd_quick_two_sum(s.hi, s.lo); }
DD_FN dd_t dd_mul(dd_t a, dd_t b) { dd_t p = dd_two_prod(a.hi, b.hi); p.lo += a.hi * b.lo + a.lo * b.hi; return dd_quick_two_sum(p.hi, p.lo); }
DD_FN dd_t dd_mul_d(dd_t a, double b) { dd_t p = dd_two_prod(a.hi, b); p.lo += a.lo * b; return dd_quick_two_sum(p.hi, p.lo); }
DD_FN dd_t dd_sqr(dd_t a) { dd_t p = dd_two_prod(a.hi, a.hi); p.lo += 2.0 * a.hi * a.lo; return dd_quick_two_sum(p.hi, p.lo); }
DD_FN dd_t dd_div(dd_t a, dd_t b) { // three quotient digits
    const double q1 = a.hi / b.hi;
    dd_t r = dd_sub(a, dd_mul_d(b, q1));
    const double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul_d(b, q2));
    const double q3 = r.hi / b.hi;
    dd_t q = dd_quick_two_sum(q1, q2);
    return dd_add_d(q, q3);
}
DD_FN dd_t dd_ldexp(dd_t a, int e) { return dd_make(ldexp(a.hi, e), ldexp(a.lo, e)); }
DD_FN dd_t dd_from_u64(uint64_t x) { const double h = (double)(x & ~0x7ffull), l = (double)(x & 0x7ffull); return dd_two_sum(h, l); } // (both halves exact)

// e^a, |a| <= ~1500.  a = k ln 2 + r, |r| <= ln2 / 2; s = expm1(r / 512) by Taylor; nine times s <- 2 s + s^2; e^a = (1 + s) 2^k.
DD_FN dd_t dd_exp(dd_t a) {
    const dd_t LN2 = dd_make(6.931471805599452862e-01, 2.319046813846299558e-17);
    const double k = floor(a.hi / LN2.hi + 0.5);
    // k ln 2 with the product's low words kept (k <= 2^11: k * LN2.hi is exact to 2^-106 in two_prod)
    dd_t r = dd_sub(a, dd_mul_d(LN2, k));
    r = dd_ldexp(r, -9);
    // Taylor: r + r^2/2! + ... (|r| < 6.8e-4: the term of order 11 is below 1e-42)
    const double inv_fact[9] = {1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880, 1.0 / 3628800, 1.0 / 39916800};
    const double inv_fact_lo[9] = {9.25185853854297e-18, 2.3129646346357427e-18, 1.1564823173178714e-19, -5.300543954373577e-20, 1.7209558293420705e-22, 2.1511947866775882e-23, -1.858393274046472e-22, 2.3767714622250297e-23, -1.448814070935912e-24};
    dd_t p = dd_sqr(r);
    dd_t s = dd_add(r, dd_ldexp(p, -1));
    for (int i = 0; i < 9; i++) {
        p = dd_mul(p, r);
        const dd_t c = dd_make(inv_fact[i], inv_fact_lo[i]);
        s = dd_add(s, dd_mul(p, c));
    }
    for (int i = 0; i < 9; i++) s = dd_add(dd_ldexp(s, 1), dd_sqr(s));
    s = dd_add_d(s, 1.0);
    return dd_ldexp(s, (int)k);
}
// ln a, a > 0: y0 = log(a.hi) in fp64 (a few ulp: |y0 - ln a| <= 2^-50 |y0| + 2^-53), then one Newton step y = y0 + a e^-y0 - 1: with
// d = ln a - y0 the step returns y0 + e^d - 1 = ln a + d^2 / 2 + ...: for the arguments used here (a in [0.85, 1], |ln a| <= 0.17)
// d^2 / 2 < 2^-106; what remains is the error of dd_exp and of the final operations (measured: tests/test_dec.py)
DD_FN dd_t dd_log(dd_t a) {
    const dd_t y = dd_from(log(a.hi));
    const dd_t e = dd_exp(dd_neg(y));
    return dd_add(y, dd_add_d(dd_mul(a, e), -1.0));
}

// floor of |a| and the distance of |a| to the nearest integer; a as an exact pair.  fl as two doubles (fh + fl, both integers): |a| may
// exceed 2^63.
struct dd_floor_t { double fh, fl, frac; };
DD_FN dd_floor_t dd_floor_abs(dd_t a) {
    if (a.hi < 0 || (a.hi == 0 && a.lo < 0)) a = dd_neg(a);
    dd_floor_t r;
    const double h = floor(a.hi), fh = a.hi - h; // fh in [0, 1), exact
    const double l = floor(a.lo), fl = a.lo - l; // fl in [0, 1), exact
    double f = fh + fl;                          // in [0, 2); exact unless both have bits 53 places apart (then the rounding is far below any bound used)
    double carry = 0.0;
    if (f >= 1.0) { f -= 1.0; carry = 1.0; }
    r.fh = h; r.fl = l + carry; r.frac = f;
    return r;
}

// ---- repr(double): the shortest decimal that reads back as x, the closest such one (Python float_repr_style 'short' = David Gay's
// dtoa mode 0), as digits * 10^exp10.  Free-format digit generation after Steele & White / Burger & Dybvig in 128-bit integers:
// good for 1e-10 <= x <= 1e10 (returns 0 outside; the caller then treats the value as known to fp64 only).
typedef unsigned __int128 dd_u128;
DD_FN int dd_shortest(double x, uint64_t *digits, int *exp10) {
    if (!(x >= 1e-10 && x <= 1e10)) return 0;
    int e2; const double fr = frexp(x, &e2);
    const uint64_t m = (uint64_t)ldexp(fr, 53); e2 -= 53; // x = m 2^e2, 2^52 <= m < 2^53, -87 <= e2 <= -19
    const bool even = (m & 1) == 0, boundary = m == (1ull << 52);
    // r / s = x, m+ / s and m- / s the half gaps to the neighbours
    dd_u128 r = (dd_u128)m << (boundary ? 2 : 1), s = (dd_u128)1 << ((boundary ? 2 : 1) - e2), mp = boundary ? 2 : 1, mm = 1;
    // k = ceil(log10((r + m+) / s)) by estimate and fix-up
    int k = (int)ceil(log10(x) - 1e-10);
    if (k >= 0) { for (int i = 0; i < k; i++) s *= 10; } else { for (int i = 0; i < -k; i++) { r *= 10; mp *= 10; mm *= 10; } }
    auto high_ok = [&]() { return even ? (r + mp >= s) : (r + mp > s); };
    if (high_ok()) { s *= 10; k++; }
    else { // the estimate may also be one too high
        while (true) { const dd_u128 r10 = r * 10, mp10 = mp * 10; if (even ? (r10 + mp10 >= s) : (r10 + mp10 > s)) break; r = r10; mp = mp10; mm *= 10; k--; }
    }
    uint64_t D = 0; int nd = 0;
    for (;;) {
        r *= 10; mp *= 10; mm *= 10;
        unsigned d = 0; while (r >= s) { r -= s; d++; } // (a digit: at most nine subtractions; no 128-bit division on the device)
        const bool tc1 = even ? (r <= mm) : (r < mm);
        const bool tc2 = even ? (r + mp >= s) : (r + mp > s);
        k--;
        if (!tc1 && !tc2) { D = D * 10 + d; nd++; if (nd >= 17) break; continue; }
        if (tc1 && tc2) { const dd_u128 r2 = r << 1; if (r2 > s || (r2 == s && (d & 1))) d++; }
        else if (tc2) d++;
        D = D * 10 + d; nd++;
        break;
    }
    *digits = D; *exp10 = k;
    return nd;
}
// 10^k for 0 <= k <= 45, exact in double-double
DD_FN dd_t dd_pow10(int k) {
    dd_t r = dd_from(1.0), b = dd_from(10.0);
    while (k) { if (k & 1) r = dd_mul(r, b); k >>= 1; if (k) b = dd_sqr(b); }
    return r;
}
// the decimal Decimal(repr(x)) holds, as a double-double (relative error <= 2^-103); ok = 0: x outside dd_shortest's range
DD_FN dd_t dd_repr_value(double x, int *ok) {
    uint64_t D; int e10;
    const int nd = dd_shortest(x, &D, &e10);
    *ok = nd > 0;
    if (!nd) return dd_from(x);
    dd_t v = dd_from_u64(D);
    if (e10 > 0) v = dd_mul(v, dd_pow10(e10)); else if (e10 < 0) v = dd_div(v, dd_pow10(-e10));
    return v;
}
