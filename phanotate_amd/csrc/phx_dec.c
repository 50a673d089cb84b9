/* phx_dec.c — Python's decimal.Decimal (libmpdec, prec 28, ROUND_HALF_EVEN) restated for the operations PHANOTATE's weights go
 * through.  See phx_dec.h.  Coefficients are little-endian base-1e9 integers; ln / exp work in decimal fixed point and are rounded
 * correctly (the Python context sets allcr: mpd_qln / mpd_qexp return the correctly rounded value) by a Ziv loop.
 * Checked against Python's own decimal module operation by operation: tests/test_dec.py. */
#define _GNU_SOURCE /* newlocale / uselocale / strtod_l */
#include "phx_dec.h"

#include <locale.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Number text in the "C" locale whatever LC_NUMERIC the host application runs under (with a comma-decimal locale
 * "%.*e" prints "1,5e+00", dec_from_str rejected it and the replay went on with 0).  uselocale() switches the calling thread only. */
static locale_t c_locale(void) {
    static locale_t loc; /* (a lost race leaks one small object) */
    locale_t l = __atomic_load_n(&loc, __ATOMIC_ACQUIRE);
    if (!l) {
        l = newlocale(LC_ALL_MASK, "C", (locale_t)0);
        locale_t expect = (locale_t)0;
        if (l && !__atomic_compare_exchange_n(&loc, &expect, l, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) { freelocale(l); l = expect; }
    }
    return l;
}
double phx_strtod_c(const char *s, char **end) {
    locale_t l = c_locale();
    return l ? strtod_l(s, end, l) : strtod(s, end);
}
int phx_snprintf_c(char *buf, size_t cap, const char *fmt, ...) {
    locale_t l = c_locale(), old = l ? uselocale(l) : (locale_t)0;
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(buf, cap, fmt, ap);
    va_end(ap);
    if (old) uselocale(old);
    return n;
}

#define BASE 1000000000u
static const uint32_t P10[10] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u, 1000000000u};

/* ------------------------------------------------------------------------------------------------ big integers, base 1e9 */
static int bn_trim(const uint32_t *d, int n) { while (n > 0 && d[n - 1] == 0) n--; return n; }
static int bn_cmp(const uint32_t *a, int na, const uint32_t *b, int nb) {
    if (na != nb) return na < nb ? -1 : 1;
    for (int i = na - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
static int bn_mul_small(uint32_t *d, int n, uint64_t m) { /* m <= 2^32; returns the new length */
    uint64_t c = 0;
    if (m == 0) return 0;
    for (int i = 0; i < n; i++) { const uint64_t t = (uint64_t)d[i] * m + c; d[i] = (uint32_t)(t % BASE); c = t / BASE; }
    while (c) { d[n++] = (uint32_t)(c % BASE); c /= BASE; }
    return n;
}
static uint64_t bn_divmod_small(uint32_t *d, int *n, uint64_t m) { /* m <= 2^32; quotient in place, returns the remainder */
    uint64_t r = 0;
    for (int i = *n - 1; i >= 0; i--) { const uint64_t t = r * BASE + d[i]; d[i] = (uint32_t)(t / m); r = t % m; }
    *n = bn_trim(d, *n);
    return r;
}
static int bn_add(uint32_t *r, const uint32_t *a, int na, const uint32_t *b, int nb) { /* r may alias a or b */
    uint32_t c = 0;
    const int n = na > nb ? na : nb;
    for (int i = 0; i < n; i++) {
        uint32_t t = (i < na ? a[i] : 0u) + (i < nb ? b[i] : 0u) + c;
        c = t >= BASE; if (c) t -= BASE;
        r[i] = t;
    }
    int m = n;
    if (c) r[m++] = 1u;
    return m;
}
static int bn_sub(uint32_t *r, const uint32_t *a, int na, const uint32_t *b, int nb) { /* a >= b; r may alias a */
    int32_t c = 0;
    for (int i = 0; i < na; i++) {
        int64_t t = (int64_t)a[i] - (i < nb ? (int64_t)b[i] : 0) - c;
        c = t < 0; if (c) t += BASE;
        r[i] = (uint32_t)t;
    }
    return bn_trim(r, na);
}
static int bn_mul(uint32_t *r, const uint32_t *a, int na, const uint32_t *b, int nb) { /* r must not alias */
    if (na == 0 || nb == 0) return 0;
    memset(r, 0, sizeof(uint32_t) * (size_t)(na + nb));
    for (int i = 0; i < na; i++) {
        uint64_t c = 0;
        const uint64_t ai = a[i];
        if (!ai) continue;
        for (int j = 0; j < nb; j++) { const uint64_t t = ai * b[j] + r[i + j] + c; r[i + j] = (uint32_t)(t % BASE); c = t / BASE; }
        for (int k = i + nb; c; k++) { const uint64_t t = r[k] + c; r[k] = (uint32_t)(t % BASE); c = t / BASE; }
    }
    return bn_trim(r, na + nb);
}
static int bn_ndigits(const uint32_t *d, int n) {
    if (n == 0) return 1;
    int k = 1; uint32_t t = d[n - 1];
    while (t >= 10) { t /= 10; k++; }
    return (n - 1) * 9 + k;
}
static int bn_shl10(uint32_t *d, int n, int k) { /* times 10^k */
    if (n == 0 || k == 0) return n;
    const int q = k / 9, r = k % 9;
    if (r) n = bn_mul_small(d, n, P10[r]);
    if (q) { memmove(d + q, d, sizeof(uint32_t) * (size_t)n); memset(d, 0, sizeof(uint32_t) * (size_t)q); n += q; }
    return n;
}
/* floor(d / 10^k); *dtop = the most significant dropped digit, *sticky = the dropped digits below it are not all zero */
static int bn_shr10(uint32_t *d, int n, int k, int *dtop, int *sticky) {
    *dtop = 0; *sticky = 0;
    if (k <= 0) return n;
    if (bn_ndigits(d, n) < k || n == 0) { /* everything goes; the top dropped digit is a leading zero unless exactly k digits */
        int st = 0; for (int i = 0; i < n; i++) st |= d[i] != 0;
        if (n && bn_ndigits(d, n) == k) { /* top digit of d is dtop */
            uint32_t t = d[n - 1]; while (t >= 10) { if (t % 10) *sticky = 1; t /= 10; }
            *dtop = (int)t;
            for (int i = 0; i < n - 1; i++) if (d[i]) *sticky = 1;
        } else *sticky = st;
        return 0;
    }
    const int k1 = k - 1; /* first drop k-1 digits (sticky), then one more (dtop) */
    const int q = k1 / 9, r = k1 % 9;
    for (int i = 0; i < q; i++) if (d[i]) *sticky = 1;
    if (q) { memmove(d, d + q, sizeof(uint32_t) * (size_t)(n - q)); n -= q; }
    if (r) { if (bn_divmod_small(d, &n, P10[r])) *sticky = 1; }
    *dtop = (int)bn_divmod_small(d, &n, 10);
    return n;
}
/* q = floor(a / b), rem = a mod b; b != 0; arrays of at least na + 2 limbs */
static void bn_divmod(uint32_t *q, int *nq, uint32_t *rem, int *nr, const uint32_t *a, int na, const uint32_t *b, int nb) {
    if (bn_cmp(a, na, b, nb) < 0) { *nq = 0; memcpy(rem, a, sizeof(uint32_t) * (size_t)na); *nr = na; return; }
    if (nb == 1) {
        memcpy(q, a, sizeof(uint32_t) * (size_t)na); *nq = na;
        const uint64_t r = bn_divmod_small(q, nq, b[0]);
        rem[0] = (uint32_t)r; *nr = r ? 1 : 0;
        return;
    }
    static __thread uint32_t u[2 * DEC_LIMBS + 8], v[2 * DEC_LIMBS + 8];
    const uint64_t f = BASE / ((uint64_t)b[nb - 1] + 1);
    memcpy(u, a, sizeof(uint32_t) * (size_t)na); int nu = na;
    memcpy(v, b, sizeof(uint32_t) * (size_t)nb); int nv = nb;
    if (f > 1) { nu = bn_mul_small(u, nu, f); nv = bn_mul_small(v, nv, f); }
    if (nu == na) u[nu++] = 0; /* one extra high limb */
    const int m = nu - nb - 1;  /* quotient limbs - 1 */
    for (int j = m; j >= 0; j--) {
        const uint64_t num = (uint64_t)u[j + nb] * BASE + u[j + nb - 1];
        uint64_t qh = num / v[nb - 1], rh = num % v[nb - 1];
        while (qh >= BASE || qh * v[nb - 2] > rh * BASE + u[j + nb - 2]) { qh--; rh += v[nb - 1]; if (rh >= BASE) break; }
        int64_t borrow = 0; uint64_t carry = 0;
        for (int i = 0; i < nb; i++) {
            const uint64_t p = qh * v[i] + carry; carry = p / BASE;
            int64_t t = (int64_t)u[i + j] - (int64_t)(p % BASE) - borrow;
            borrow = t < 0; if (borrow) t += BASE;
            u[i + j] = (uint32_t)t;
        }
        int64_t t = (int64_t)u[j + nb] - (int64_t)carry - borrow;
        if (t < 0) { /* qh was one too large: add back */
            qh--;
            uint32_t c = 0;
            for (int i = 0; i < nb; i++) { uint32_t s = u[i + j] + v[i] + c; c = s >= BASE; if (c) s -= BASE; u[i + j] = s; }
            t += (int64_t)BASE; t = (t + c) % (int64_t)BASE;
        }
        u[j + nb] = (uint32_t)t;
        q[j] = (uint32_t)qh;
    }
    *nq = bn_trim(q, m + 1);
    int n = bn_trim(u, nb);
    if (f > 1) (void)bn_divmod_small(u, &n, f);
    memcpy(rem, u, sizeof(uint32_t) * (size_t)n); *nr = n;
}

/* ------------------------------------------------------------------------------------------------ decimals */
int dec_digits(const dec_t *a) { return bn_ndigits(a->d, a->n); }
int dec_is_zero(const dec_t *a) { return a->n == 0; }
void dec_from_i64(dec_t *r, int64_t v) {
    r->sign = v < 0; r->exp = 0; r->n = 0;
    uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
    while (u) { r->d[r->n++] = (uint32_t)(u % BASE); u /= BASE; }
}
int dec_from_str(dec_t *r, const char *s) {
    r->sign = 0; r->exp = 0; r->n = 0;
    while (*s == ' ') s++;
    if (*s == '-') { r->sign = 1; s++; } else if (*s == '+') s++;
    int nd = 0, frac = 0, seen_dot = 0, any = 0;
    for (; *s; s++) {
        if (*s >= '0' && *s <= '9') {
            any = 1;
            if (r->n >= DEC_LIMBS - 2) return -1;
            r->n = bn_mul_small(r->d, r->n, 10);
            if (*s != '0') { uint32_t one[1] = {(uint32_t)(*s - '0')}; r->n = bn_add(r->d, r->d, r->n, one, 1); }
            nd++; if (seen_dot) frac++;
        } else if (*s == '.' && !seen_dot) seen_dot = 1;
        else break;
    }
    if (!any) return -1;
    int e = 0;
    if (*s == 'e' || *s == 'E') {
        s++;
        int es = 1;
        if (*s == '-') { es = -1; s++; } else if (*s == '+') s++;
        if (!(*s >= '0' && *s <= '9')) return -1;
        for (; *s >= '0' && *s <= '9'; s++) { e = e * 10 + (*s - '0'); if (e > 100000000) return -1; }
        e *= es;
    }
    while (*s == ' ') s++;
    if (*s) return -1;
    r->exp = e - frac;
    return 0;
}
void dec_from_double(dec_t *r, double x) { /* Decimal.from_float: n / 2^k in lowest terms -> n * 5^k, exponent -k */
    r->sign = signbit(x) ? 1 : 0; r->exp = 0; r->n = 0;
    x = fabs(x);
    if (x == 0.0) return;
    int e; const double fr = frexp(x, &e);
    uint64_t m = (uint64_t)ldexp(fr, 53); e -= 53;
    while (!(m & 1)) { m >>= 1; e++; }
    uint64_t t = m;
    while (t) { r->d[r->n++] = (uint32_t)(t % BASE); t /= BASE; }
    if (e >= 0) { for (int i = 0; i < e; i++) r->n = bn_mul_small(r->d, r->n, 2); }
    else { for (int i = 0; i < -e; i++) r->n = bn_mul_small(r->d, r->n, 5); r->exp = e; }
}
static int coeff_str(const dec_t *a, char *buf) { /* decimal digits of the coefficient, no leading zeros; returns the count */
    if (a->n == 0) { buf[0] = '0'; buf[1] = 0; return 1; }
    int k = sprintf(buf, "%u", a->d[a->n - 1]);
    for (int i = a->n - 2; i >= 0; i--) k += sprintf(buf + k, "%09u", a->d[i]);
    return k;
}
int dec_to_str(const dec_t *a, char *out, int cap) { /* Decimal.__str__ */
    char dg[DEC_LIMBS * 9 + 2];
    const int nd = coeff_str(a, dg);
    const int leftdigits = a->exp + nd;
    const int dotplace = (a->exp <= 0 && leftdigits > -6) ? leftdigits : 1;
    char *p = out, *end = out + cap - 24;
#define PUT(c) do { if (p >= end) return -1; *p++ = (c); } while (0)
    if (a->sign) PUT('-');
    if (dotplace <= 0) { PUT('0'); PUT('.'); for (int i = 0; i < -dotplace; i++) PUT('0'); for (int i = 0; i < nd; i++) PUT(dg[i]); }
    else if (dotplace >= nd) { for (int i = 0; i < nd; i++) PUT(dg[i]); for (int i = nd; i < dotplace; i++) PUT('0'); }
    else { for (int i = 0; i < dotplace; i++) PUT(dg[i]); PUT('.'); for (int i = dotplace; i < nd; i++) PUT(dg[i]); }
#undef PUT
    if (leftdigits != dotplace) p += sprintf(p, "E%+d", leftdigits - dotplace);
    *p = 0;
    return (int)(p - out);
}
void dec_round(dec_t *a, int prec) {
    const int nd = dec_digits(a);
    if (a->n == 0 || nd <= prec) return;
    const int k = nd - prec;
    int dtop, sticky;
    a->n = bn_shr10(a->d, a->n, k, &dtop, &sticky);
    a->exp += k;
    const int odd = a->n ? (int)(a->d[0] & 1u) : 0;
    if (dtop > 5 || (dtop == 5 && (sticky || odd))) {
        const uint32_t one[1] = {1u};
        a->n = bn_add(a->d, a->d, a->n, one, 1);
        if (bn_ndigits(a->d, a->n) > prec) { (void)bn_divmod_small(a->d, &a->n, 10); a->exp++; } /* 99..9 + 1 */
    }
}
int dec_cmp(const dec_t *a, const dec_t *b) {
    if (a->n == 0 && b->n == 0) return 0;
    if (a->n == 0) return b->sign ? 1 : -1;
    if (b->n == 0) return a->sign ? -1 : 1;
    if (a->sign != b->sign) return a->sign ? -1 : 1;
    const int s = a->sign ? -1 : 1;
    const int la = dec_digits(a) + a->exp, lb = dec_digits(b) + b->exp;
    if (la != lb) return la < lb ? -s : s;
    static __thread dec_t x, y;
    x = *a; y = *b;
    if (x.exp > y.exp) { x.n = bn_shl10(x.d, x.n, x.exp - y.exp); } else if (y.exp > x.exp) { y.n = bn_shl10(y.d, y.n, y.exp - x.exp); }
    return s * bn_cmp(x.d, x.n, y.d, y.n);
}
int dec_is_integer(const dec_t *a) {
    if (a->n == 0 || a->exp >= 0) return 1;
    if (-a->exp >= dec_digits(a)) return 0;
    dec_t t = *a; int dtop, sticky;
    t.n = bn_shr10(t.d, t.n, -a->exp, &dtop, &sticky);
    return dtop == 0 && !sticky;
}

static void addsub(dec_t *r, const dec_t *a, const dec_t *b, int bneg, int prec) {
    static __thread dec_t hi_s, lo_s;
    const int sb = b->sign ^ bneg;
    const dec_t *hi = a, *lo = b; int shi = a->sign, slo = sb;
    if (a->exp < b->exp) { hi = b; lo = a; shi = sb; slo = a->sign; }
    hi_s = *hi; lo_s = *lo;
    int shift = hi_s.exp - lo_s.exp;
    if (hi_s.n && shift > 0) {
        const int ndh = dec_digits(&hi_s);
        int k = prec + 3 - ndh; if (k < 0) k = 0;
        if (shift > k && lo_s.n && lo_s.exp + dec_digits(&lo_s) <= hi_s.exp - k && shift > 200) {
            /* the small operand lies wholly below the digits that can matter: it only says "not exact" and which way */
            hi_s.n = bn_shl10(hi_s.d, hi_s.n, k); hi_s.exp -= k;
            lo_s.n = 1; lo_s.d[0] = 1; lo_s.exp = hi_s.exp;
        } else { hi_s.n = bn_shl10(hi_s.d, hi_s.n, shift); hi_s.exp = lo_s.exp; }
    } else if (!hi_s.n) hi_s.exp = lo_s.exp;
    r->exp = lo_s.exp;
    if (shi == slo) { r->n = bn_add(r->d, hi_s.d, hi_s.n, lo_s.d, lo_s.n); r->sign = shi; }
    else {
        const int c = bn_cmp(hi_s.d, hi_s.n, lo_s.d, lo_s.n);
        if (c >= 0) { r->n = bn_sub(r->d, hi_s.d, hi_s.n, lo_s.d, lo_s.n); r->sign = c ? shi : 0; }
        else { r->n = bn_sub(r->d, lo_s.d, lo_s.n, hi_s.d, hi_s.n); r->sign = slo; }
    }
    dec_round(r, prec);
}
void dec_add(dec_t *r, const dec_t *a, const dec_t *b, int prec) { addsub(r, a, b, 0, prec); }
void dec_sub(dec_t *r, const dec_t *a, const dec_t *b, int prec) { addsub(r, a, b, 1, prec); }
void dec_mul(dec_t *r, const dec_t *a, const dec_t *b, int prec) {
    static __thread uint32_t t[2 * DEC_LIMBS + 2];
    int n = bn_mul(t, a->d, a->n, b->d, b->n);
    const int sign = a->sign ^ b->sign, exp = a->exp + b->exp;
    if (n > DEC_LIMBS) { /* (operands far beyond the working precisions used here) drop low limbs, keeping a sticky bit in the lowest */
        const int dropl = n - DEC_LIMBS; int st = 0;
        for (int i = 0; i < dropl; i++) st |= t[i] != 0;
        memmove(t, t + dropl, sizeof(uint32_t) * (size_t)DEC_LIMBS); n = DEC_LIMBS;
        if (st && (t[0] % 10 == 0 || t[0] % 10 == 5)) t[0] += 1;
        r->exp = exp + 9 * dropl;
    } else r->exp = exp;
    memcpy(r->d, t, sizeof(uint32_t) * (size_t)n);
    r->n = n; r->sign = n ? sign : sign;
    dec_round(r, prec);
}
int dec_div(dec_t *r, const dec_t *a, const dec_t *b, int prec) { /* _mpd_qdiv */
    if (b->n == 0) return -1;
    const int ideal = a->exp - b->exp;
    r->sign = a->sign ^ b->sign;
    if (a->n == 0) { r->n = 0; r->exp = ideal; return 0; }
    static __thread uint32_t num[2 * DEC_LIMBS + 8], q[2 * DEC_LIMBS + 8], rem[2 * DEC_LIMBS + 8];
    int shift = dec_digits(b) - dec_digits(a) + prec + 1;
    if (shift < 0) shift = 0;
    memcpy(num, a->d, sizeof(uint32_t) * (size_t)a->n);
    int nn = bn_shl10(num, a->n, shift);
    int nq, nr;
    bn_divmod(q, &nq, rem, &nr, num, nn, b->d, b->n);
    int exp = ideal - shift;
    if (nr) { if (nq == 0) { q[0] = 1; nq = 1; } else if (q[0] % 10 == 0 || q[0] % 10 == 5) q[0] += 1; } /* inexact: the remainder as a sticky digit */
    else { /* exact: towards the ideal exponent */
        while (exp < ideal && nq && q[0] % 10 == 0) { (void)bn_divmod_small(q, &nq, 10); exp++; }
    }
    if (nq > DEC_LIMBS) return -1;
    memcpy(r->d, q, sizeof(uint32_t) * (size_t)nq); r->n = nq; r->exp = exp;
    dec_round(r, prec);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ fixed point: value = m / 10^(9 FL) */
#define FXMAX 40 /* limbs of fraction at most */
typedef struct { int n; uint32_t d[2 * FXMAX + 8]; } fx_t;
static void fx_set_small(fx_t *r, uint32_t v, int FL) { memset(r->d, 0, sizeof(uint32_t) * (size_t)(FL + 1)); r->d[FL] = v; r->n = v ? FL + 1 : 0; }
static void fx_mul(fx_t *r, const fx_t *a, const fx_t *b, int FL) {
    static __thread uint32_t t[4 * FXMAX + 16];
    const int n = bn_mul(t, a->d, a->n, b->d, b->n);
    if (n <= FL) { r->n = 0; return; }
    memcpy(r->d, t + FL, sizeof(uint32_t) * (size_t)(n - FL)); r->n = n - FL;
}
static void fx_div(fx_t *r, const fx_t *a, const fx_t *b, int FL) { /* a / b */
    static __thread uint32_t num[4 * FXMAX + 16], q[4 * FXMAX + 16], rem[4 * FXMAX + 16];
    memset(num, 0, sizeof(uint32_t) * (size_t)FL);
    memcpy(num + FL, a->d, sizeof(uint32_t) * (size_t)a->n);
    int nq, nr;
    bn_divmod(q, &nq, rem, &nr, num, a->n ? a->n + FL : 0, b->d, b->n);
    memcpy(r->d, q, sizeof(uint32_t) * (size_t)nq); r->n = nq;
}
/* e^x for a fixed-point x in [0, ~3): x / 2^12, Taylor, twelve squarings; error below 1e7 units of the last place */
static void fx_exp(fx_t *r, const fx_t *x, int FL) {
    fx_t y = *x, term, sum;
    for (int i = 0; i < 3; i++) (void)bn_divmod_small(y.d, &y.n, 16);
    fx_set_small(&sum, 1, FL);
    sum.n = bn_add(sum.d, sum.d, sum.n, y.d, y.n);
    term = y;
    for (uint32_t k = 2; term.n; k++) {
        fx_t t2; fx_mul(&t2, &term, &y, FL); term = t2;
        (void)bn_divmod_small(term.d, &term.n, k);
        sum.n = bn_add(sum.d, sum.d, sum.n, term.d, term.n);
    }
    for (int i = 0; i < 12; i++) { fx_t t2; fx_mul(&t2, &sum, &sum, FL); sum = t2; }
    *r = sum;
}
/* 2 atanh(1 / m) = ln((m + 1) / (m - 1)) */
static void fx_atanh2(fx_t *r, uint32_t m, int FL) {
    fx_t p, acc;
    fx_set_small(&p, 1, FL); (void)bn_divmod_small(p.d, &p.n, m);
    acc = p;
    for (uint32_t k = 3; p.n; k += 2) {
        (void)bn_divmod_small(p.d, &p.n, (uint64_t)m * m);
        fx_t t = p; (void)bn_divmod_small(t.d, &t.n, k);
        acc.n = bn_add(acc.d, acc.d, acc.n, t.d, t.n);
    }
    acc.n = bn_mul_small(acc.d, acc.n, 2);
    *r = acc;
}
static void fx_ln10(fx_t *r, int FL) { /* ln 10 = 3 ln 2 + ln 1.25 = 3 * 2 atanh(1/3) + 2 atanh(1/9); cached per FL */
    static __thread fx_t cache[FXMAX + 1]; static __thread uint8_t have[FXMAX + 1];
    if (!have[FL]) {
        fx_t a, b; fx_atanh2(&a, 3, FL); fx_atanh2(&b, 9, FL);
        a.n = bn_mul_small(a.d, a.n, 3);
        a.n = bn_add(a.d, a.d, a.n, b.d, b.n);
        cache[FL] = a; have[FL] = 1;
    }
    *r = cache[FL];
}
/* |a| as fixed point (truncated) */
static int fx_from_dec(fx_t *r, const dec_t *a, int FL) {
    const int F = 9 * FL;
    static __thread dec_t t; t = *a;
    const int sh = F + a->exp;
    if (sh >= 0) { if (dec_digits(a) + sh > 9 * (2 * FXMAX)) return -1; t.n = bn_shl10(t.d, t.n, sh); }
    else { int dt, st; t.n = bn_shr10(t.d, t.n, -sh, &dt, &st); }
    if (t.n > 2 * FXMAX + 4) return -1;
    memcpy(r->d, t.d, sizeof(uint32_t) * (size_t)t.n); r->n = t.n;
    return 0;
}
/* Ziv: does coeff +- err (units of 10^exp) round to one prec-digit value?  If so it is left in r. */
static int ziv_round(dec_t *r, const uint32_t *c, int n, int exp, int sign, uint64_t err, int prec) {
    static __thread dec_t lo, hi;
    uint32_t e[3]; int ne = 0; uint64_t t = err;
    while (t) { e[ne++] = (uint32_t)(t % BASE); t /= BASE; }
    if (n > DEC_LIMBS - 2) return 0;
    if (bn_cmp(c, n, e, ne) <= 0) return 0;
    lo.sign = hi.sign = sign; lo.exp = hi.exp = exp;
    lo.n = bn_sub(lo.d, c, n, e, ne);
    hi.n = bn_add(hi.d, c, n, e, ne);
    dec_round(&lo, prec); dec_round(&hi, prec);
    if (lo.exp != hi.exp || bn_cmp(lo.d, lo.n, hi.d, hi.n) != 0) return 0;
    *r = lo;
    return 1;
}
int dec_exp(dec_t *r, const dec_t *a, int prec) {
    if (a->n == 0) { dec_from_i64(r, 1); return 0; }
    if (dec_digits(a) + a->exp > 7) return -1; /* |a| >= 1e7: not restated */
    for (int FL = (prec + 24) / 9 + 2; FL <= FXMAX; FL += 2) {
        fx_t x, l10, q, rr, e;
        if (fx_from_dec(&x, a, FL)) return -1;
        fx_ln10(&l10, FL);
        /* |a| = k ln 10 + rr, 0 <= rr < ln 10 */
        int nq, nr;
        static __thread uint32_t qq[4 * FXMAX + 16], rem[4 * FXMAX + 16];
        bn_divmod(qq, &nq, rem, &nr, x.d, x.n, l10.d, l10.n);
        const int64_t k = nq ? (int64_t)qq[0] + (nq > 1 ? (int64_t)qq[1] * BASE : 0) : 0;
        memcpy(rr.d, rem, sizeof(uint32_t) * (size_t)nr); rr.n = nr;
        (void)q;
        fx_exp(&e, &rr, FL); /* in [1, 10) */
        int exp10;
        if (!a->sign) exp10 = -9 * FL + (int)k;
        else { fx_t one, inv; fx_set_small(&one, 1, FL); fx_div(&inv, &one, &e, FL); e = inv; exp10 = -9 * FL - (int)k; }
        /* errors: truncation of x (1 unit), k * (error of ln 10) <= 1e7 units in rr, fx_exp 1e7 units, all relative to e in [0.1, 10]:
         * 1e10 units bound them with room */
        if (ziv_round(r, e.d, e.n, exp10, 0, 10000000000ull, prec)) return 0;
    }
    return -1;
}
int dec_ln(dec_t *r, const dec_t *a, int prec) {
    if (a->n == 0 || a->sign) return -1;
    { dec_t one; dec_from_i64(&one, 1); if (dec_cmp(a, &one) == 0) { r->sign = 0; r->exp = 0; r->n = 0; return 0; } }
    const int nd = dec_digits(a);
    const int64_t e10 = (int64_t)a->exp + nd; /* a = m * 10^e10, m in [0.1, 1) */
    for (int FL = (prec + 24) / 9 + 2; FL <= FXMAX; FL += 2) {
        const int F = 9 * FL;
        fx_t m; { dec_t t = *a; t.exp = -nd; if (fx_from_dec(&m, &t, FL)) return -1; }
        /* y0 ~ ln m on a 1e-17 grid (exact in fixed point) */
        double md = 0; { char dg[DEC_LIMBS * 9 + 2]; coeff_str(a, dg); char buf[40]; snprintf(buf, sizeof buf, "0.%.20s", dg); md = phx_strtod_c(buf, NULL); }
        const double y0 = log(md);
        const int64_t qg = llround(y0 * 1e17);
        fx_t Y0; { dec_t t; dec_from_i64(&t, qg < 0 ? -qg : qg); t.exp = -17; if (fx_from_dec(&Y0, &t, FL)) return -1; }
        fx_t E, T, one; fx_exp(&E, &Y0, FL); fx_set_small(&one, 1, FL);
        if (qg <= 0) fx_mul(&T, &m, &E, FL); else fx_div(&T, &m, &E, FL); /* m * exp(-y0) = 1 + t */
        int tneg; fx_t t;
        if (bn_cmp(T.d, T.n, one.d, one.n) >= 0) { tneg = 0; t.n = bn_sub(t.d, T.d, T.n, one.d, one.n); }
        else { tneg = 1; t.n = bn_sub(t.d, one.d, one.n, T.d, T.n); }
        /* log1p(t) = t - t^2/2 + t^3/3 - ...  (|t| ~ 1e-16) as a positive and a negative sum */
        fx_t pos, neg, p = t; pos.n = 0; neg.n = 0;
        for (uint32_t k = 1; p.n; k++) {
            fx_t term = p; if (k > 1) (void)bn_divmod_small(term.d, &term.n, k);
            const int negative = tneg ? 1 : ((k & 1) == 0);
            if (negative) neg.n = bn_add(neg.d, neg.d, neg.n, term.d, term.n); else pos.n = bn_add(pos.d, pos.d, pos.n, term.d, term.n);
            fx_t p2; fx_mul(&p2, &p, &t, FL); p = p2;
        }
        if (qg >= 0) pos.n = bn_add(pos.d, pos.d, pos.n, Y0.d, Y0.n); else neg.n = bn_add(neg.d, neg.d, neg.n, Y0.d, Y0.n);
        { fx_t l10; fx_ln10(&l10, FL); const uint64_t ae = (uint64_t)(e10 < 0 ? -e10 : e10);
          if (ae > 4000000000ull) return -1;
          l10.n = bn_mul_small(l10.d, l10.n, ae);
          if (e10 >= 0) pos.n = bn_add(pos.d, pos.d, pos.n, l10.d, l10.n); else neg.n = bn_add(neg.d, neg.d, neg.n, l10.d, l10.n); }
        int sign; fx_t S;
        if (bn_cmp(pos.d, pos.n, neg.d, neg.n) >= 0) { sign = 0; S.n = bn_sub(S.d, pos.d, pos.n, neg.d, neg.n); }
        else { sign = 1; S.n = bn_sub(S.d, neg.d, neg.n, pos.d, pos.n); }
        /* errors: m truncated (1 unit, relative to m >= 0.1: 10 units in ln), fx_exp 1e7, the product / quotient 2, the series a few,
         * |e10| * (error of ln 10 < 1e3 units): 1e10 + |e10| * 1e3 units bound them */
        const uint64_t err = 10000000000ull + (uint64_t)(e10 < 0 ? -e10 : e10) * 1000ull;
        if (ziv_round(r, S.d, S.n, -F, sign, err, prec)) return 0;
    }
    return -1;
}

static void pow_uint(dec_t *r, const dec_t *base, uint64_t n, int wprec) { /* _mpd_qpow_uint */
    static __thread dec_t t;
    *r = *base;
    int top = 63; while (top > 0 && !((n >> top) & 1)) top--;
    for (int b = top - 1; b >= 0; b--) {
        dec_mul(&t, r, r, wprec); *r = t;
        if ((n >> b) & 1) { dec_mul(&t, r, base, wprec); *r = t; }
    }
}
int dec_pow(dec_t *r, const dec_t *a, const dec_t *b, int prec) { /* mpd_qpow for a > 0 */
    if (a->n == 0 || a->sign) return -1;
    if (b->n == 0) { dec_from_i64(r, 1); return 0; }
    const int intexp = dec_is_integer(b);
    if (intexp) { /* _mpd_qpow_int */
        static __thread dec_t tb, q;
        int wprec = prec + dec_digits(b) + b->exp + 2;
        q = *b; if (q.exp > 0) { q.n = bn_shl10(q.d, q.n, q.exp); q.exp = 0; } else if (q.exp < 0) { int dt, st; q.n = bn_shr10(q.d, q.n, -q.exp, &dt, &st); q.exp = 0; }
        if (q.n > 2) return -1;
        const uint64_t n = (uint64_t)q.d[0] + (q.n > 1 ? (uint64_t)q.d[1] * BASE : 0);
        { /* the result's exponent must stay inside Emin / Emax of the reference's context (+-999999): beyond that libmpdec rounds to
             subnormals or signals Overflow, which is not restated here (and the int exponents below would wrap) */
            const double adj = (double)a->exp + (double)dec_digits(a) - 1.0;
            if (fabs(adj) * (double)n > 999999.0 || n > 100000000ull) return -1;
        }
        if (b->sign) { wprec += 1; dec_t one; dec_from_i64(&one, 1); if (dec_div(&tb, &one, a, wprec)) return -1; } else tb = *a;
        pow_uint(r, &tb, n, wprec);
        dec_round(r, prec);
        return 0;
    }
    { /* |a| == 1 with a non-integer exponent: _qcheck_pow_one */
        dec_t one; dec_from_i64(&one, 1);
        if (dec_cmp(a, &one) == 0) { dec_from_i64(r, 1); r->n = bn_shl10(r->d, r->n, prec - 1); r->exp = -(prec - 1); return 0; }
    }
    /* _mpd_qpow_real: exp(b ln a) at prec + 4 + MPD_EXPDIGITS (19), both functions correctly rounded */
    static __thread dec_t l, y;
    const int nda = dec_digits(a);
    const int wprec = (nda > prec ? nda : prec) + 4 + 19;
    if (dec_ln(&l, a, wprec)) return -1;
    dec_mul(&y, &l, b, wprec);
    if (dec_exp(r, &y, wprec)) return -1;
    { dec_t one; dec_from_i64(&one, 1); if (dec_cmp(r, &one) == 0) { dec_from_i64(r, 1); r->n = bn_shl10(r->d, r->n, prec - 1); r->exp = -(prec - 1); } }
    dec_round(r, prec);
    return 0;
}

int dec_trunc_limbs(const dec_t *a, int shift10, uint64_t *out, int nl) {
    static __thread dec_t t; t = *a;
    const int e = a->exp + shift10;
    if (e > 0) { if (dec_digits(a) + e > 9 * (DEC_LIMBS - 2)) return -1; t.n = bn_shl10(t.d, t.n, e); }
    else if (e < 0) { int dt, st; t.n = bn_shr10(t.d, t.n, -e, &dt, &st); }
    memset(out, 0, sizeof(uint64_t) * (size_t)nl);
    for (int w = 0; w < 2 * nl && t.n; w++) {
        const uint64_t r32 = bn_divmod_small(t.d, &t.n, 4294967296ull);
        out[w / 2] |= r32 << (32 * (w & 1));
    }
    if (t.n) return -1;
    if (out[nl - 1] >> 63) return -1;
    if (a->sign) { uint64_t c = 1; for (int i = 0; i < nl; i++) { out[i] = ~out[i] + c; c = (c && out[i] == 0) ? 1 : 0; } }
    return 0;
}

/* ------------------------------------------------------------------------------------------------ double-double, repr */
static void two_prod(double a, double b, double *p, double *e) { *p = a * b; *e = fma(a, b, -*p); }
static void two_sum(double a, double b, double *s, double *e) { *s = a + b; const double bb = *s - a; *e = (a - (*s - bb)) + (b - bb); }
static void dd_mul(double ah, double al, double bh, double bl, double *rh, double *rl) {
    double p, e; two_prod(ah, bh, &p, &e); e += ah * bl + al * bh;
    two_sum(p, e, rh, rl);
}
static void dd_div(double ah, double al, double bh, double bl, double *rh, double *rl) {
    const double q1 = ah / bh;
    double ph, pl; dd_mul(bh, bl, q1, 0.0, &ph, &pl);
    double s, e; two_sum(ah, -ph, &s, &e); e += al - pl;
    const double q2 = (s + e) / bh;
    dd_mul(bh, bl, q2, 0.0, &ph, &pl);
    double s2, e2; two_sum(s, -ph, &s2, &e2); e2 += e - pl;
    const double q3 = (s2 + e2) / bh;
    double h, l; two_sum(q1, q2, &h, &l); l += q3;
    two_sum(h, l, rh, rl);
}
static void dd_pow10(int k, double *h, double *l) { /* exact for k <= 45 */
    *h = 1.0; *l = 0.0;
    double bh = 10.0, bl = 0.0;
    while (k) { if (k & 1) dd_mul(*h, *l, bh, bl, h, l); k >>= 1; if (k) dd_mul(bh, bl, bh, bl, &bh, &bl); }
}
void dec_to_dd(const dec_t *a, double *hi, double *lo) {
    *hi = 0; *lo = 0;
    if (a->n == 0) return;
    static __thread dec_t t; t = *a;
    const int nd = dec_digits(a);
    if (nd > 34) { int dt, st; t.n = bn_shr10(t.d, t.n, nd - 34, &dt, &st); t.exp += nd - 34; }
    unsigned __int128 N = 0;
    for (int i = t.n - 1; i >= 0; i--) N = N * BASE + t.d[i];
    double h = (double)N;
    double l = (double)((__int128)N - (__int128)h);
    two_sum(h, l, &h, &l);
    if (t.exp > 0) { double ph, pl; dd_pow10(t.exp, &ph, &pl); dd_mul(h, l, ph, pl, &h, &l); }
    else if (t.exp < 0) { double ph, pl; dd_pow10(-t.exp, &ph, &pl); dd_div(h, l, ph, pl, &h, &l); }
    if (a->sign) { h = -h; l = -l; }
    *hi = h; *lo = l;
}

int phx_repr_double(double x, char *out, int cap) { /* float_repr_style 'short' (Python/pystrtod.c format_float_short, 'r') */
    char buf[40];
    int p;
    for (p = 1; p <= 17; p++) { phx_snprintf_c(buf, sizeof buf, "%.*e", p - 1, x); if (phx_strtod_c(buf, NULL) == x) break; }
    if (p > 17) p = 17;
    /* buf = [-]d[.ddd]e[+-]XX */
    char dg[24]; int nd = 0; const char *s = buf; int neg = 0;
    if (*s == '-') { neg = 1; s++; }
    for (; *s && *s != 'e'; s++) if (*s != '.') dg[nd++] = *s;
    const int e10 = atoi(s + 1);
    while (nd > 1 && dg[nd - 1] == '0') nd--; /* (p is minimal, so no trailing zeros except for 0 itself) */
    char *q = out, *end = out + cap - 8;
    if (cap < 32) return -1;
    if (neg) *q++ = '-';
    if (e10 >= -4 && e10 < 16) { /* fixed notation, at least one digit after the point */
        if (e10 < 0) { *q++ = '0'; *q++ = '.'; for (int i = 0; i < -e10 - 1; i++) *q++ = '0'; for (int i = 0; i < nd; i++) *q++ = dg[i]; }
        else {
            for (int i = 0; i <= e10; i++) *q++ = i < nd ? dg[i] : '0';
            *q++ = '.';
            if (nd > e10 + 1) for (int i = e10 + 1; i < nd; i++) *q++ = dg[i]; else *q++ = '0';
        }
    } else {
        *q++ = dg[0];
        if (nd > 1) { *q++ = '.'; for (int i = 1; i < nd; i++) *q++ = dg[i]; }
        q += snprintf(q, (size_t)(end - q + 8), "e%c%02d", e10 < 0 ? '-' : '+', e10 < 0 ? -e10 : e10);
    }
    *q = 0;
    return (int)(q - out);
}

int dec_start_weights(int n, const char (*texts)[32], const double *w, dec_t *out) {
    static __thread dec_t t[16];
    int bad = 0;
    if (n > 16) n = 16;
    int mx = 0;
    for (int i = 0; i < n; i++) {
        char buf[48];
        if (texts && texts[i][0]) { if (dec_from_str(&t[i], texts[i])) { dec_from_i64(&t[i], 0); bad = -1; } }
        else { phx_repr_double(w[i], buf, sizeof buf); if (dec_from_str(&t[i], buf)) { dec_from_i64(&t[i], 0); bad = -1; } }
        if (dec_cmp(&t[i], &t[mx]) > 0) mx = i;
    }
    for (int i = 0; i < n; i++) if (dec_div(&out[i], &t[i], &t[mx], DEC_PREC)) { dec_from_i64(&out[i], 0); bad = -1; }
    return bad;
}

/* ------------------------------------------------------------------------------------------------ test hook (tests/test_dec.py) */
/* op: "add" "sub" "mul" "div" "pow" "ln" "exp" "str" (a alone) "float" (a = repr of a double -> Decimal(float)) "repr" (repr(float(a)))
 * "trunc1000" (int(a * 1000) as decimal text); result text in out.  Returns the length or a negative error. */
int phx_dec_eval(const char *op, const char *a, const char *b, int prec, char *out, int cap) {
    static __thread dec_t x, y, r;
    if (!strcmp(op, "repr")) return phx_repr_double(phx_strtod_c(a, NULL), out, cap);
    if (!strcmp(op, "float")) { dec_from_double(&r, phx_strtod_c(a, NULL)); return dec_to_str(&r, out, cap); }
    if (dec_from_str(&x, a)) return -2;
    if (b && *b && dec_from_str(&y, b)) return -2;
    int rc = 0;
    if (!strcmp(op, "add")) dec_add(&r, &x, &y, prec);
    else if (!strcmp(op, "sub")) dec_sub(&r, &x, &y, prec);
    else if (!strcmp(op, "mul")) dec_mul(&r, &x, &y, prec);
    else if (!strcmp(op, "div")) rc = dec_div(&r, &x, &y, prec);
    else if (!strcmp(op, "pow")) rc = dec_pow(&r, &x, &y, prec);
    else if (!strcmp(op, "ln")) rc = dec_ln(&r, &x, prec);
    else if (!strcmp(op, "exp")) rc = dec_exp(&r, &x, prec);
    else if (!strcmp(op, "str")) r = x;
    else if (!strcmp(op, "dd")) { double h, l; dec_to_dd(&x, &h, &l); return phx_snprintf_c(out, (size_t)cap, "%.17g %.17g", h, l); }
    else if (!strcmp(op, "trunc1000")) {
        uint64_t w[18];
        if (dec_trunc_limbs(&x, 3, w, 18)) return -3;
        /* print as hex words, most significant first (the test turns them into a python int) */
        int k = 0;
        for (int i = 17; i >= 0; i--) k += snprintf(out + k, (size_t)(cap - k), "%016llx", (unsigned long long)w[i]);
        return k;
    } else return -4;
    if (rc) return -5;
    return dec_to_str(&r, out, cap);
}
