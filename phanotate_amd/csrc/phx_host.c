/* phx_host.c — host-side I/O of the drop-in CLI, in C so that reading a multi-contig FASTA and writing the gene table
 * do not cost more than the GPU path (include/phx.h, "host utilities").  No device code here.
 *
 * What it stands in for in the reference: the FASTA reader of the external `genbank` package (phanotate_modules/file.py:1-5,
 * phanotate.py:32-35: every record's name = first token of the header, README.md:45) and the tabular writer
 * phanotate_modules/locus.py:39-56.
 */
#define _DEFAULT_SOURCE
#define _POSIX_C_SOURCE 200809L
#include <ctype.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>

#include "phx_dec.h" /* phx_snprintf_c: number text in the "C" locale */
#include <unistd.h>
#include <zlib.h>

#include <immintrin.h>

#include "../../include/phx.h"

struct phx_fasta {
    char *buf;        /* the sequences back to back */
    char *names;      /* NUL-terminated names back to back */
    int64_t *seq_off; /* n + 1 */
    int64_t *name_off;
    int32_t n;
};

static int is_space(int c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

/* worker threads of the host utilities: PHX_HOST_THREADS, else the online cores, at most 16 */
static int host_threads(void) {
    const char *e = getenv("PHX_HOST_THREADS");
    long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    return n < 1 ? 1 : (n > 16 ? 16 : (int)n);
}
/* run fn(arg + i * stride) for i < n on n threads (the calling thread takes the last one; a thread that cannot be created runs inline) */
static void run_threads(void *(*fn)(void *), void *arg, size_t stride, int n) {
    pthread_t th[16];
    int made[16];
    for (int i = 0; i + 1 < n; i++) made[i] = pthread_create(&th[i], NULL, fn, (char *)arg + (size_t)i * stride) == 0;
    fn((char *)arg + (size_t)(n - 1) * stride);
    for (int i = 0; i + 1 < n; i++) { if (made[i]) pthread_join(th[i], NULL); else fn((char *)arg + (size_t)i * stride); }
}

/* a buffer of hundreds of MB is touched once, front to back: with 4 KB pages most of that time is page faults */
static void big_pages(void *p, size_t n) {
#ifdef MADV_HUGEPAGE
    if (n >= (size_t)(32 << 20)) { const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095; if (a < (uintptr_t)p + n) (void)madvise((void *)a, ((uintptr_t)p + n - a) & ~(size_t)4095, MADV_HUGEPAGE); }
#else
    (void)p; (void)n;
#endif
}
typedef struct { int fd; char *buf; int64_t beg, end; int64_t err; } rd_job; /* err: 0 whole slice read, -1 I/O error, 1 + bytes read for a short slice */
static void *rd_work(void *arg) {
    rd_job *j = (rd_job *)arg;
    int64_t n = j->beg;
    while (n < j->end) {
        const int64_t want = j->end - n > (1 << 30) ? (1 << 30) : j->end - n;
        const ssize_t got = pread(j->fd, j->buf + n, (size_t)want, (off_t)n);
        if (got < 0) { j->err = -1; return NULL; }
        if (got == 0) { j->err = 1 + (n - j->beg); return NULL; }
        n += got;
    }
    return NULL;
}

/* whole file into one buffer: a plain file with pread() on worker threads, a gzip file (magic 1f 8b) through zlib */
static int slurp(const char *path, char **out, int64_t *len) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return PHX_E_IO;
    unsigned char magic[2] = {0, 0};
    struct stat st;
    /* the magic is looked at with pread() and only in a regular file: a pipe, /dev/stdin or <(zcat x.gz) must reach zlib with every
     * byte still in it (zlib reads plain text and gzip alike) */
    const int regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    if (regular && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
        const int64_t size = (int64_t)st.st_size;
        char *b = (char *)malloc((size_t)size + 16);
        if (!b) { close(fd); return PHX_E_NOMEM; }
        big_pages(b, (size_t)size);
        /* slices of the file read side by side (pread): one thread copies a 500 MB file out of the page cache in 0.1 s */
        rd_job job[16];
        const int T = size < (8 << 20) ? 1 : host_threads();
        for (int t = 0; t < T; t++) { job[t].fd = fd; job[t].buf = b; job[t].beg = size / T * t; job[t].end = t + 1 == T ? size : size / T * (t + 1); job[t].err = 0; }
        run_threads(rd_work, job, sizeof(rd_job), T);
        close(fd);
        int64_t n = size;
        for (int t = T - 1; t >= 0; t--) { if (job[t].err < 0) { free(b); return PHX_E_IO; } if (job[t].err > 0) n = job[t].beg + (job[t].err - 1); } /* a file that shrank: up to the first short slice */
        *out = b; *len = n;
        return PHX_OK;
    }
    gzFile g = gzdopen(fd, "rb"); /* takes the descriptor over (position 0: nothing has been read from it) */
    if (!g) { close(fd); return PHX_E_IO; }
    gzbuffer(g, 1 << 20);
    int64_t cap = 1 << 22, n = 0;
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

/* The file is cut into one segment per thread at line starts.  Pass 1 sizes every segment (records, name bytes, sequence bytes in
 * front of / behind its first header), a prefix sum places them, pass 2 copies names and stripped sequence lines to their places. */
typedef struct {
    const char *b;
    int64_t p0, p1;               /* segment [p0, p1): whole lines */
    int64_t nrec, name_bytes, pre, post;
    int64_t rec0, name0, seq0;    /* first record index, name offset, sequence write offset of this segment */
    int have_prev;                /* a record starts before this segment: its leading sequence lines belong to it */
    phx_fasta *f;
    char *dst;
    int pass;
} fa_seg;

static void *fa_work(void *arg) {
    fa_seg *g = (fa_seg *)arg;
    const char *b = g->b;
    int64_t nrec = 0, nb = 0, pre = 0, post = 0;
    int64_t k = g->rec0 - 1, nw = g->name0, w = g->seq0;
    phx_fasta *f = g->f;
    for (int64_t p = g->p0; p < g->p1;) {
        const char *nl = (const char *)memchr(b + p, '\n', (size_t)(g->p1 - p));
        const int64_t e = nl ? nl - b : g->p1;
        if (b[p] == '>') {
            int64_t a = p + 1;
            while (a < e && is_space((unsigned char)b[a])) a++; /* line[1:].split()[0] */
            int64_t z = a;
            while (z < e && !is_space((unsigned char)b[z])) z++;
            if (g->pass == 1) { nrec++; nb += z - a + 1; }
            else {
                k++;
                f->seq_off[k] = w;
                f->name_off[k] = nw;
                memcpy(f->names + nw, b + a, (size_t)(z - a));
                nw += z - a;
                f->names[nw++] = 0;
            }
        } else {
            int64_t a = p, z = e;
            while (a < z && is_space((unsigned char)b[a])) a++;
            while (z > a && is_space((unsigned char)b[z - 1])) z--;
            if (g->pass == 1) { if (nrec) post += z - a; else pre += z - a; }
            else if (k >= 0 && z > a) { memcpy(g->dst + w, b + a, (size_t)(z - a)); w += z - a; } /* text before the first header is ignored */
        }
        p = e + 1;
    }
    if (g->pass == 1) { g->nrec = nrec; g->name_bytes = nb; g->pre = pre; g->post = post; }
    return NULL;
}

int phx_fasta_read(const char *path, phx_fasta **out) {
    if (!path || !out) return PHX_E_ARG;
    *out = NULL;
    char *b = NULL;
    int64_t len = 0;
    int rc = slurp(path, &b, &len);
    if (rc) return rc;
    fa_seg seg[16];
    int T = len < (4 << 20) ? 1 : host_threads();
    int64_t cut = 0;
    int ns = 0;
    for (int i = 0; i < T && cut < len; i++) { /* segment ends: the first line start at or behind (i + 1) * len / T */
        int64_t e = i + 1 == T ? len : (len / T) * (i + 1);
        if (e < cut) e = cut;
        if (e < len) { const char *nl = (const char *)memchr(b + e, '\n', (size_t)(len - e)); e = nl ? nl - b + 1 : len; }
        memset(&seg[ns], 0, sizeof(seg[ns]));
        seg[ns].b = b; seg[ns].p0 = cut; seg[ns].p1 = e; seg[ns].pass = 1;
        cut = e;
        ns++;
    }
    if (ns) run_threads(fa_work, seg, sizeof(fa_seg), ns);
    int64_t nrec = 0, name_bytes = 0, seq_bytes = 0;
    for (int i = 0; i < ns; i++) {
        seg[i].rec0 = nrec; seg[i].name0 = name_bytes; seg[i].seq0 = seq_bytes; seg[i].have_prev = nrec > 0;
        seq_bytes += (nrec > 0 ? seg[i].pre : 0) + seg[i].post;
        nrec += seg[i].nrec; name_bytes += seg[i].name_bytes;
    }
    phx_fasta *f = (phx_fasta *)calloc(1, sizeof(*f));
    if (!f) { free(b); return PHX_E_NOMEM; }
    f->buf = (char *)malloc((size_t)seq_bytes + 16);
    if (f->buf) big_pages(f->buf, (size_t)seq_bytes);
    f->names = (char *)malloc((size_t)name_bytes + 1);
    f->seq_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrec + 1));
    f->name_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nrec + 1));
    if (!f->buf || !f->names || !f->seq_off || !f->name_off || nrec > 0x7fffffff) { free(b); phx_fasta_free(f); return PHX_E_NOMEM; }
    for (int i = 0; i < ns; i++) { seg[i].pass = 2; seg[i].f = f; seg[i].dst = f->buf; }
    if (ns) run_threads(fa_work, seg, sizeof(fa_seg), ns);
    free(b);
    f->n = (int32_t)nrec;
    f->seq_off[f->n] = seq_bytes;
    f->name_off[f->n] = name_bytes;
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

typedef struct {
    int32_t c0, c1;          /* contigs [c0, c1) */
    const char *const *names;
    const phx_gene *genes;
    const int64_t *offsets;
    const int32_t *status;
    char *dst;               /* this thread's region of the text buffer */
    int64_t len;             /* bytes written */
} fmt_job;

static void *fmt_work(void *arg) {
    fmt_job *j = (fmt_job *)arg;
    char *p = j->dst;
    for (int32_t i = j->c0; i < j->c1; i++) {
        if (j->status[i] < 0) continue;
        const char *nm = j->names[i];
        const size_t ln = strlen(nm);
        memcpy(p, "#id:\t", 5); p += 5;
        memcpy(p, nm, ln); p += ln;
        *p++ = '\n';
        memcpy(p, "#START\tSTOP\tFRAME\tCONTIG\tSCORE\n", 31); p += 31;
        for (int64_t k = j->offsets[i]; k < j->offsets[i + 1]; k++) {
            const phx_gene *g = &j->genes[k];
            if (g->frame == 4 || g->frame == -4) continue; /* a tRNA feature: Locus.tabular lists features(include=['CDS']), locus.py:42 */
            const int32_t a = g->strand < 0 ? g->right : g->left, z = g->strand < 0 ? g->left : g->right; /* locus.py:44-46 */
            p = put_int(p, a); *p++ = '\t';
            p = put_int(p, z); *p++ = '\t';
            *p++ = (char)(44 - g->strand); *p++ = '\t'; /* chr(44 - strand), locus.py:51 */
            memcpy(p, nm, ln); p += ln;
            *p++ = '\t';
            p += phx_snprintf_c(p, 25, "%E", g->score); /* '%E' % weight (locus.py:54) in the "C" locale */
            *p++ = '\n';
        }
    }
    j->len = p - j->dst;
    return NULL;
}

int phx_format_tabular(int32_t n, const char *const *names, const phx_gene *genes, const int64_t *offsets, const int32_t *status, char **text, int64_t *text_len) {
    if (n < 0 || !text || !text_len || (n > 0 && (!names || !offsets || !status))) return PHX_E_ARG;
    *text = NULL; *text_len = 0;
    /* upper bound of the text: header lines + per gene two coordinates (11 each), strand, name, score (<= 24), separators */
    int64_t *bound = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    if (!bound) return PHX_E_NOMEM;
    int64_t need = 1;
    for (int32_t i = 0; i < n; i++) {
        bound[i] = need;
        if (status[i] < 0) continue;
        const int64_t ln = (int64_t)strlen(names[i]);
        need += 6 + ln + 1 + 34 + (offsets[i + 1] - offsets[i]) * (11 + 1 + 11 + 1 + 1 + 1 + ln + 1 + 24 + 1);
    }
    bound[n] = need;
    char *b = (char *)malloc((size_t)need);
    if (!b) { free(bound); return PHX_E_NOMEM; }
    /* contiguous contig ranges of about equal size, one per thread; every thread writes at its range's bound, then the pieces move together */
    fmt_job job[16];
    const int T = need < (1 << 20) ? 1 : host_threads();
    int nj = 0;
    int32_t c = 0;
    for (int t = 0; t < T && c < n; t++) {
        const int64_t target = t + 1 == T ? need : bound[0] + (need - bound[0]) / T * (t + 1);
        int32_t e = c;
        while (e < n && (bound[e + 1] <= target || e == c)) e++;
        if (t + 1 == T) e = n;
        job[nj].c0 = c; job[nj].c1 = e; job[nj].names = names; job[nj].genes = genes; job[nj].offsets = offsets; job[nj].status = status;
        job[nj].dst = b + (bound[c] - 1); job[nj].len = 0;
        nj++;
        c = e;
    }
    if (nj) run_threads(fmt_work, job, sizeof(fmt_job), nj);
    char *p = b;
    for (int t = 0; t < nj; t++) { if (job[t].dst != p) memmove(p, job[t].dst, (size_t)job[t].len); p += job[t].len; }
    free(bound);
    *p = 0;
    *text = b; *text_len = p - b;
    return PHX_OK;
}

void phx_free_text(char *text) { free(text); }

/* ---- bases packed for the link and for the kernels (phx_upload): residue-split bit planes ----
 * Position p = 3 k + r of a contig: stream r, index k.  A record holds 32 indices of the three streams as nine 32-bit words
 * [stream][b0, b1, amb] (36 bytes per 96 letters): b0, b1 = base code a0 c1 t2 g3 — for an ambiguity code the base the reference
 * counts it as (s, b, v -> g; the others -> a, functions.py:159-163) —, amb = not one of acgt; a letter outside acgtnryswkmbvdh
 * (KeyError in rev_comp, functions.py:20-24) is (0, 1, 1), a position outside the contig (0, 0, 1); any case (functions.py:144).
 * This is the form k_features works in (phx_feat_core.h): 3 bits per base cross PCIe, and the staging pass that has to touch every
 * letter anyway does the stride-3 split (pext) the kernels would otherwise pay for. */
static const uint8_t kCode3[32] = {6, 0, 7, 1, 4, 6, 6, 3, 4, 6, 6, 4, 6, 4, 4, 6, /* ` a b c d e f g h i j k l m n o */
                                   6, 6, 4, 7, 2, 6, 7, 4, 6, 4, 6, 6, 6, 6, 6, 6}; /* p q r s t u v w x y z { | } ~ DEL */
static inline uint8_t code3_of(uint8_t ch) {
    const uint8_t x = (uint8_t)(ch | 0x20u);
    return (x & 0xe0u) == 0x60u ? kCode3[x & 31u] : (uint8_t)6; /* ('@' and '[' .. '_' land on entries that are not letters) */
}
static inline void rec_outside(uint32_t *rec) { for (int i = 0; i < 9; i++) rec[i] = i % 3 == 2 ? ~0u : 0u; }
/* one record from up to 96 letters (positions behind n: outside) */
static void pack_rec_scalar(const uint8_t *in, size_t n, uint32_t *rec) {
    rec_outside(rec);
    if (n > 96) n = 96;
    for (size_t p = 0; p < n; p++) {
        const uint32_t cd = code3_of(in[p]), r = (uint32_t)(p % 3), k = (uint32_t)(p / 3);
        uint32_t *q = rec + r * 3;
        q[0] |= (cd & 1u) << k; q[1] |= ((cd >> 1) & 1u) << k;
        if (!(cd & 4u)) q[2] &= ~(1u << k);
    }
}
#if defined(__x86_64__)
__attribute__((target("avx2,bmi2"))) static size_t pack_avx2(const uint8_t *in, size_t n, uint32_t *out) { /* whole records only; returns the records written */
    const __m256i lut_lo = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i *)kCode3));
    const __m256i lut_hi = _mm256_broadcastsi128_si256(_mm_loadu_si128((const __m128i *)(kCode3 + 16)));
    const __m256i c20 = _mm256_set1_epi8(0x20), c1f = _mm256_set1_epi8(0x1f), ce0 = _mm256_set1_epi8((char)0xe0), c60 = _mm256_set1_epi8(0x60),
                  c10 = _mm256_set1_epi8(0x10), cbad = _mm256_set1_epi8(6);
    static const uint64_t M[3] = {0x9249249249249249ull, 0x2492492492492492ull, 0x4924924924924924ull}; /* letters 0..63 of stream r */
    static const uint32_t N[3] = {0x24924924u, 0x49249249u, 0x92492492u};                                 /* letters 64..95 */
    static const int CNT[3] = {22, 21, 21};
    size_t nr = 0;
    for (size_t i = 0; i + 96 <= n; i += 96, nr++) {
        __m256i cd[3];
        for (int j = 0; j < 3; j++) {
            const __m256i x = _mm256_or_si256(_mm256_loadu_si256((const __m256i *)(in + i + 32 * j)), c20);
            const __m256i v = _mm256_and_si256(x, c1f);
            const __m256i lo = _mm256_shuffle_epi8(lut_lo, v), hi = _mm256_shuffle_epi8(lut_hi, v);
            __m256i nb = _mm256_blendv_epi8(lo, hi, _mm256_cmpeq_epi8(_mm256_and_si256(v, c10), c10));
            cd[j] = _mm256_blendv_epi8(cbad, nb, _mm256_cmpeq_epi8(_mm256_and_si256(x, ce0), c60));
        }
        uint32_t *rec = out + nr * 9;
        for (int pl = 0; pl < 3; pl++) { /* bit `pl` of every code to bit 7 of its byte, then one mask bit per letter */
            const uint64_t m0 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(cd[0], 7 - pl));
            const uint64_t m1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(cd[1], 7 - pl));
            const uint32_t m2 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(cd[2], 7 - pl));
            const uint64_t lo64 = m0 | (m1 << 32);
            for (int r = 0; r < 3; r++) rec[r * 3 + pl] = (uint32_t)_pext_u64(lo64, M[r]) | ((uint32_t)_pext_u64(m2, N[r]) << CNT[r]);
        }
    }
    return nr;
}
#endif
/* n letters, the first of them at a position of its contig that is a multiple of 96 -> nrec records at `out`: ceil(n / 96) from the
 * letters (positions behind n: outside), the rest all-outside (a contig's unused records and the spacer behind it) */
void phx_pack_planes(const char *in, int64_t n, uint32_t *out, int64_t nrec) {
    size_t done = 0;
    if (n < 0) n = 0;
#if defined(__x86_64__)
    static int have = -1;
    if (have < 0) have = (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2")) ? 1 : 0;
    if (have && n >= 96) done = pack_avx2((const uint8_t *)in, (size_t)n, out);
#endif
    for (; (int64_t)done < nrec && (int64_t)done * 96 < n; done++) pack_rec_scalar((const uint8_t *)in + done * 96, (size_t)n - done * 96, out + done * 9);
    for (; (int64_t)done < nrec; done++) rec_outside(out + done * 9);
}
