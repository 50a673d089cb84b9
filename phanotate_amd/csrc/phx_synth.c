/* phx_synth.c — deterministic synthetic phage-like contig generator (host utility).
 *
 * Not part of the reference (PHANOTATE ships no generator); this is the workload
 * definition for BASELINE.json configs 4-5 ("synthetic 50 kb phage contigs"),
 * following the spec in SURVEY.md §8(d).  Integer PRNG = splitmix64 so the same
 * (seed, L) gives the same contig everywhere; used by bench.py, the tests and the
 * golden-vector generator (tests/golden/make_golden.py) through the C-ABI symbol
 * phx_synth_contig (include/phx.h).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } sm64;

static uint64_t sm_next(sm64 *r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double sm_unif(sm64 *r) { return (double)(sm_next(r) >> 11) * (1.0 / 9007199254740992.0); }

static char draw_base(sm64 *r, double gc) {
    double u = sm_unif(r);
    if (u < gc) return (u < gc * 0.5) ? 'g' : 'c';
    u = (u - gc) / (1.0 - gc);
    return (u < 0.5) ? 'a' : 't';
}
static double clip(double x) { return x < 0.05 ? 0.05 : (x > 0.95 ? 0.95 : x); }
static char comp(char c) { return c == 'a' ? 't' : c == 't' ? 'a' : c == 'g' ? 'c' : 'g'; }

typedef struct { char *p; int64_t n, cap; } sbuf;
static void sb_put(sbuf *b, char c) {
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 4096; b->p = (char *)realloc(b->p, (size_t)b->cap); }
    b->p[b->n++] = c;
}

/* Writes exactly L lower-case acgt characters to out (no terminator). Returns 0, or -1 on bad args. */
int phx_synth_contig(uint64_t seed, int64_t L, char *out) {
    if (L <= 0 || !out) return -1;
    sm64 r = { 0x9E3779B97F4A7C15ULL ^ seed };
    double gc = 0.35 + 0.20 * sm_unif(&r);
    double gcp[3] = { clip(gc + 0.05), clip(gc - 0.05), clip(gc + 0.10) };
    sbuf b = { 0, 0, 0 };
    sbuf g = { 0, 0, 0 };
    int rev = 0;
    while (b.n < L) {
        /* spacer */
        int sl = (int)floor(-20.0 * log(1.0 - sm_unif(&r)));
        if (sl > 150) sl = 150;
        int64_t sp0 = b.n;
        for (int i = 0; i < sl; i++) sb_put(&b, draw_base(&r, gc));
        if (sm_unif(&r) < 0.1) rev = !rev;
        /* gene length, log-normal around 200 codons */
        double u1 = 1.0 - sm_unif(&r), u2 = sm_unif(&r);
        double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        int n = (int)floor(200.0 * exp(0.8 * z));
        if (n < 30) n = 30;
        if (n > 1500) n = 1500;
        /* sense strand of the gene */
        g.n = 0;
        double us = sm_unif(&r);
        const char *sc = us < 0.85 ? "atg" : (us < 0.95 ? "gtg" : "ttg");
        sb_put(&g, sc[0]); sb_put(&g, sc[1]); sb_put(&g, sc[2]);
        for (int c = 0; c < n; c++) {
            char x, y, w;
            do {
                x = draw_base(&r, gcp[0]); y = draw_base(&r, gcp[1]); w = draw_base(&r, gcp[2]);
            } while (x == 't' && ((y == 'a' && (w == 'a' || w == 'g')) || (y == 'g' && w == 'a')));
            sb_put(&g, x); sb_put(&g, y); sb_put(&g, w);
        }
        double ut = sm_unif(&r);
        const char *tc = ut < 1.0 / 3 ? "taa" : (ut < 2.0 / 3 ? "tga" : "tag");
        sb_put(&g, tc[0]); sb_put(&g, tc[1]); sb_put(&g, tc[2]);
        int plant = sm_unif(&r) < 0.6;
        if (!rev) {
            /* Shine-Dalgarno 'aggagg' ending 7 nt upstream of the start codon */
            if (plant && sl >= 13) memcpy(b.p + sp0 + sl - 13, "aggagg", 6);
            for (int64_t i = 0; i < g.n; i++) sb_put(&b, g.p[i]);
        } else {
            for (int64_t i = g.n - 1; i >= 0; i--) sb_put(&b, comp(g.p[i]));
            int64_t t0 = b.n;
            for (int i = 0; i < 13; i++) sb_put(&b, draw_base(&r, gc));
            if (plant) memcpy(b.p + t0 + 7, "cctcct", 6);
        }
    }
    memcpy(out, b.p, (size_t)L);
    free(b.p); free(g.p);
    return 0;
}
