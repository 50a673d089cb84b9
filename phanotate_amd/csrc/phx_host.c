/* phx_host.c — host-side I/O of the drop-in CLI, in C so that reading a multi-contig FASTA and writing the gene table
 * do not cost more than the GPU path (include/phx.h, "host utilities").  No device code here.
 *
 * What it stands in for in the reference: the FASTA reader of the external `genbank` package (phanotate_modules/file.py:1-5,
 * phanotate.py:32-35: every record's name = first token of the header, README.md:45) and the tabular writer
 * phanotate_modules/locus.py:39-56.
 */
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "../../include/phx.h"

struct phx_fasta {
    char *buf;        /* the file, compacted in place: the sequences back to back */
    char *names;      /* NUL-terminated names back to back */
    int64_t *seq_off; /* n + 1 */
    int64_t *name_off;
    int32_t n;
};

static int is_space(int c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

/* whole file (plain or gzip: zlib reads both) into one buffer */
static int slurp(const char *path, char **out, int64_t *len) {
    gzFile g = gzopen(path, "rb");
    if (!g) return PHX_E_IO;
    gzbuffer(g, 1 << 20);
    int64_t cap = 1 << 22, n = 0;
    if (gzdirect(g)) { /* not compressed: size the buffer from the file */
        FILE *f = fopen(path, "rb");
        if (f) { if (fseek(f, 0, SEEK_END) == 0) { long s = ftell(f); if (s > 0) cap = (int64_t)s + 16; } fclose(f); }
    }
    char *b = (char *)malloc((size_t)cap);
    if (!b) { gzclose(g); return PHX_E_NOMEM; }
    for (;;) {
        if (n == cap) {
            cap += cap / 2 + (1 << 20);
            char *nb = (char *)realloc(b, (size_t)cap);
            if (!nb) { free(b); gzclose(g); return PHX_E_NOMEM; }
            b = nb;
        }
        const int64_t want = cap - n > (1 << 30) ? (1 << 30) : cap - n;
        const int got = gzread(g, b + n, (unsigned)want);
        if (got < 0) { free(b); gzclose(g); return PHX_E_IO; }
        if (got == 0) break;
        n += got;
    }
    gzclose(g);
    *out = b; *len = n;
    return PHX_OK;
}

int phx_fasta_read(const char *path, phx_fasta **out) {
    if (!path || !out) return PHX_E_ARG;
    *out = NULL;
    char *b = NULL;
    int64_t len = 0;
    int rc = slurp(path, &b, &len);
    if (rc) return rc;
    /* pass 1: records and the room their names need */
    int64_t nrec = 0, name_bytes = 0;
    for (int64_t p = 0; p < len;) {
        const char *nl = (const char *)memchr(b + p, '\n', (size_t)(len - p));
        const int64_t e = nl ? nl - b : len;
        if (b[p] == '>') { nrec++; name_bytes += e - p; }
        p = e + 1;
    }
    phx_fasta *f = (phx_fasta *)calloc(1, sizeof(*f));
    if (!f) { free(b); return PHX_E_NOMEM; }
    f->buf = b;
    f->names = (char *)malloc((size_t)name_bytes + 1);
    f->seq_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrec + 1));
    f->name_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrec + 1));
    if (!f->names || !f->seq_off || !f->name_off || nrec > 0x7fffffff) { phx_fasta_free(f); return PHX_E_NOMEM; }
    /* pass 2: names out, sequence lines stripped and moved down (the write position never passes the read position) */
    int64_t w = 0, nw = 0;
    int32_t k = -1;
    for (int64_t p = 0; p < len;) {
        const char *nl = (const char *)memchr(b + p, '\n', (size_t)(len - p));
        const int64_t e = nl ? nl - b : len;
        if (b[p] == '>') {
            k++;
            f->seq_off[k] = w;
            f->name_off[k] = nw;
            int64_t a = p + 1;
            while (a < e && is_space((unsigned char)b[a])) a++; /* line[1:].split()[0] */
            int64_t z = a;
            while (z < e && !is_space((unsigned char)b[z])) z++;
            memcpy(f->names + nw, b + a, (size_t)(z - a));
            nw += z - a;
            f->names[nw++] = 0;
        } else if (k >= 0) { /* text before the first header is ignored */
            int64_t a = p, z = e;
            while (a < z && is_space((unsigned char)b[a])) a++;
            while (z > a && is_space((unsigned char)b[z - 1])) z--;
            if (z > a) { memmove(b + w, b + a, (size_t)(z - a)); w += z - a; }
        }
        p = e + 1;
    }
    f->n = k + 1;
    f->seq_off[f->n] = w;
    f->name_off[f->n] = nw;
    *out = f;
    return PHX_OK;
}

int32_t phx_fasta_count(const phx_fasta *f) { return f ? f->n : 0; }

int phx_fasta_record(const phx_fasta *f, int32_t i, const char **name, const char **seq, int64_t *len) {
    if (!f || i < 0 || i >= f->n) return PHX_E_ARG;
    if (name) *name = f->names + f->name_off[i];
    if (seq) *seq = f->buf + f->seq_off[i];
    if (len) *len = f->seq_off[i + 1] - f->seq_off[i];
    return PHX_OK;
}

int phx_fasta_arrays(const phx_fasta *f, const char **names, const char **seqs, int64_t *lens) {
    if (!f) return PHX_E_ARG;
    for (int32_t i = 0; i < f->n; i++) {
        if (names) names[i] = f->names + f->name_off[i];
        if (seqs) seqs[i] = f->buf + f->seq_off[i];
        if (lens) lens[i] = f->seq_off[i + 1] - f->seq_off[i];
    }
    return PHX_OK;
}

void phx_fasta_free(phx_fasta *f) {
    if (!f) return;
    free(f->buf); free(f->names); free(f->seq_off); free(f->name_off);
    free(f);
}

/* ---- tabular writer, locus.py:39-56 ---- */
static char *put_int(char *p, int32_t v) {
    char t[12];
    int n = 0;
    uint32_t u = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
    if (v < 0) *p++ = '-';
    do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = t[--n];
    return p;
}

int phx_format_tabular(int32_t n, const char *const *names, const phx_gene *genes, const int64_t *offsets, const int32_t *status, char **text, int64_t *text_len) {
    if (n < 0 || !text || !text_len || (n > 0 && (!names || !offsets || !status))) return PHX_E_ARG;
    *text = NULL; *text_len = 0;
    /* upper bound of the text: header lines + per gene two coordinates (11 each), strand, name, score (<= 24), separators */
    int64_t need = 1;
    for (int32_t i = 0; i < n; i++) {
        if (status[i] < 0) continue;
        const int64_t ln = (int64_t)strlen(names[i]);
        need += 6 + ln + 1 + 34 + (offsets[i + 1] - offsets[i]) * (11 + 1 + 11 + 1 + 1 + 1 + ln + 1 + 24 + 1);
    }
    char *b = (char *)malloc((size_t)need);
    if (!b) return PHX_E_NOMEM;
    char *p = b;
    for (int32_t i = 0; i < n; i++) {
        if (status[i] < 0) continue;
        const size_t ln = strlen(names[i]);
        memcpy(p, "#id:\t", 5); p += 5;
        memcpy(p, names[i], ln); p += ln;
        *p++ = '\n';
        memcpy(p, "#START\tSTOP\tFRAME\tCONTIG\tSCORE\n", 31); p += 31;
        for (int64_t k = offsets[i]; k < offsets[i + 1]; k++) {
            const phx_gene *g = &genes[k];
            if (g->frame == 4 || g->frame == -4) continue; /* a tRNA feature: Locus.tabular lists features(include=['CDS']), locus.py:42 */
            const int32_t a = g->strand < 0 ? g->right : g->left, z = g->strand < 0 ? g->left : g->right; /* locus.py:44-46 */
            p = put_int(p, a); *p++ = '\t';
            p = put_int(p, z); *p++ = '\t';
            *p++ = (char)(44 - g->strand); *p++ = '\t'; /* chr(44 - strand), locus.py:51 */
            memcpy(p, names[i], ln); p += ln;
            *p++ = '\t';
            p += snprintf(p, 25, "%E", g->score);
            *p++ = '\n';
        }
    }
    *p = 0;
    *text = b; *text_len = p - b;
    return PHX_OK;
}

void phx_free_text(char *text) { free(text); }
