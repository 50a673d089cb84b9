#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "phx_dec.h"
static unsigned long long s = 88172645463325252ull;
static unsigned long long rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static void rnum(char *o, int maxdig, int emin, int emax, int neg) {
    int nd = 1 + (int)(rnd() % (unsigned)maxdig), p = 0;
    if (neg && (rnd() & 1)) o[p++] = '-';
    for (int i = 0; i < nd; i++) o[p++] = (char)('0' + (i == 0 ? 1 + rnd() % 9 : rnd() % 10));
    if (rnd() % 3 == 0) { int k = (int)(rnd() % (unsigned)nd); memmove(o + p - k + 1, o + p - k, (size_t)k); o[p - k] = '.'; p++; }
    int e = emin + (int)(rnd() % (unsigned)(emax - emin + 1));
    p += sprintf(o + p, "E%d", e);
    o[p] = 0;
}
int main(void) {
    char a[200], b[200], out[4096];
    const char *ops[] = {"add", "sub", "mul", "div", "str", "dd", "trunc1000"};
    long n = 0;
    for (int it = 0; it < 300000; it++) {
        rnum(a, 60, -60, 60, 1); rnum(b, 60, -60, 60, 1);
        int prec = (it % 5 == 0) ? 1 + (int)(rnd() % 60) : 28;
        phx_dec_eval(ops[it % 7], a, b, prec, out, sizeof out); n++;
    }
    for (int it = 0; it < 20000; it++) {
        rnum(a, 30, -40, 6, 0);
        phx_dec_eval("ln", a, "0", 28, out, sizeof out);
        rnum(a, 20, -30, 2, 1);
        phx_dec_eval("exp", a, "0", 28, out, sizeof out);
        rnum(a, 28, -28, 0, 0); rnum(b, 17, -17, 3, 1);
        phx_dec_eval("pow", a, b, 28, out, sizeof out);
        sprintf(b, "%d", (int)(rnd() % 3000));
        phx_dec_eval("pow", a, b, 28, out, sizeof out);
        double x = ldexp((double)(rnd() >> 11), (int)(rnd() % 200) - 150);
        sprintf(a, "%.17g", x);
        phx_dec_eval("repr", a, "0", 28, out, sizeof out);
        phx_dec_eval("float", a, "0", 28, out, sizeof out);
        n += 6;
    }
    /* start weights */
    for (int it = 0; it < 2000; it++) {
        char texts[8][32]; double w[8]; dec_t o8[8];
        int k = 1 + (int)(rnd() % 8);
        for (int i = 0; i < k; i++) { w[i] = (double)(1 + rnd() % 1000) / 1000.0; if (rnd() & 1) texts[i][0] = 0; else snprintf(texts[i], 32, "%.6f", w[i]); }
        dec_start_weights(k, (const char (*)[32])texts, w, o8);
    }
    printf("ok %ld calls, last %s\n", n, out);
    return 0;
}
