/* phx_dec.h — the reference's number type on the host: Python's decimal.Decimal (libmpdec) at the default context, prec = 28,
 * ROUND_HALF_EVEN, restated for the operations PHANOTATE's weights go through (functions.py:26-46,174-178,281-301, orfs.py:122-127,
 * 162-173, edges.py:17-23): construction from text / integers / floats, + - * /, ** (integer and real exponents) and str().
 *
 * Why it exists: libphx solves on integers derived in fp64 / double-double on the device; the reference solves on
 * trunc(Decimal(w) * 1000).  Wherever the device cannot prove the two equal, the host replays the Decimal chain itself
 * (phx_exact.c) — below the C-ABI, without Python.  Results are coefficient-and-exponent identical to libmpdec's (tests/test_dec.py
 * compares against Python's decimal on random operands), because str(weight * 1000) — the --dump text — shows the representation.
 *
 * Not a general decimal library: no NaN / infinities / subnormals / traps; exponents are ints.  Internal header (not part of the ABI).
 */
#ifndef PHX_DEC_H
#define PHX_DEC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEC_LIMBS 80 /* base 1e9: 720 digits */
#define DEC_PREC 28

typedef struct dec_t {
    int sign;              /* 0 +, 1 - */
    int exp;               /* value = (-1)^sign * coeff * 10^exp */
    int n;                 /* limbs in use (little-endian, base 1e9); n == 0: coefficient 0 */
    uint32_t d[DEC_LIMBS];
} dec_t;

void dec_from_i64(dec_t *r, int64_t v);
int dec_from_str(dec_t *r, const char *s);   /* "[-]ddd[.ddd][E[+-]dd]"; 0 ok, -1 syntax / too long */
void dec_from_double(dec_t *r, double x);    /* exact, as Decimal(float) */
int dec_to_str(const dec_t *a, char *out, int cap); /* str(Decimal): scientific string of the spec; returns length or -1 */
int dec_digits(const dec_t *a);              /* decimal digits of the coefficient (1 for zero) */
int dec_is_zero(const dec_t *a);
int dec_cmp(const dec_t *a, const dec_t *b); /* by value */
int dec_is_integer(const dec_t *a);
void dec_round(dec_t *a, int prec);          /* half-even to at most prec digits */

void dec_add(dec_t *r, const dec_t *a, const dec_t *b, int prec);
void dec_sub(dec_t *r, const dec_t *a, const dec_t *b, int prec);
void dec_mul(dec_t *r, const dec_t *a, const dec_t *b, int prec);
int dec_div(dec_t *r, const dec_t *a, const dec_t *b, int prec); /* -1: division by zero */
/* a ** b as mpd_qpow: integer exponents by square-and-multiply at prec + digits + 2, others as exp(b ln a) through correctly
 * rounded ln / exp at prec + 23 digits; a > 0.  -1: outside what is restated here (a <= 0, huge exponents) */
int dec_pow(dec_t *r, const dec_t *a, const dec_t *b, int prec);
int dec_ln(dec_t *r, const dec_t *a, int prec);  /* correctly rounded (allcr); a > 0 */
int dec_exp(dec_t *r, const dec_t *a, int prec); /* correctly rounded (allcr) */

/* trunc(a * 10^shift10) toward zero as a two's-complement integer of nl 64-bit words (little-endian).  -1: does not fit. */
int dec_trunc_limbs(const dec_t *a, int shift10, uint64_t *out, int nl);
/* nearest double-double (hi + lo, |lo| <= ulp(hi)/2) of a: relative error <= 2^-104 */
void dec_to_dd(const dec_t *a, double *hi, double *lo);

/* repr(float) of Python 3 (float_repr_style 'short': the shortest digit string that rounds back to x, the closest such one), as the
 * text Decimal(str(x)) is built from (orfs.py:126).  Finite x only.  Returns the length. */
int phx_repr_double(double x, char *out, int cap);
/* file_handling.py:58-62: Decimal(weight) / max over n start codons.  texts[i]: the weight as written, or "" = the shortest decimal
 * that reads back as w[i].  out[i]: the 28-digit quotients. */
int dec_start_weights(int n, const char (*texts)[32], const double *w, dec_t *out); /* -1: a text that is no number, or every weight 0 (the outputs are then 0: nothing may be claimed from them) */
/* strtod / snprintf in the "C" locale, whatever LC_NUMERIC the host application has set (thread-safe: uselocale) */
double phx_strtod_c(const char *s, char **end);
int phx_snprintf_c(char *buf, size_t cap, const char *fmt, ...);
int phx_dec_eval(const char *op, const char *a, const char *b, int prec, char *out, int cap); /* test hook, see include/phx.h */

#ifdef __cplusplus
}
#endif
#endif
