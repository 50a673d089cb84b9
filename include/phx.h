/* phx.h — C-ABI of libphx, the MI355X-native drop-in for PHANOTATE's per-contig hot path.
 *
 * What it replaces in the reference (all paths under /root/reference):
 *   phanotate.py:45      orfs  = functions.get_orfs(locus)        functions.py:143-303
 *   phanotate.py:49      graph = functions.get_graph(orfs)        functions.py:307-454
 *   phanotate.py:56-64   fz.empty_graph / fz.add_edge / fz.get_path   (external `fastpathz`)
 *   phanotate.py:65-76   path -> (start, stop, strand, score) features   locus.py:29-37
 * The reference has no FFI of its own for this path (it is pure Python plus the fastpathz
 * CPython module); these entry points are what a ctypes binding in phanotate.py would call —
 * see INTEGRATION.md for the stub.
 *
 * Conventions: extern "C", plain pointers and sizes, no exceptions, no global state.  Every
 * function returns 0 (PHX_OK) or a negative PHX_E_*.  Per-contig problems (the inputs on which the
 * reference raises a Python exception) are reported in phx_result.status and never fail the batch.
 * There is NO CPU execution path in this library: without a usable HIP device phx_create fails with
 * PHX_E_NODEVICE.
 *
 * Threading: one phx_ctx per (host thread, device).  Calls on one ctx must not overlap.
 */
#ifndef PHX_H
#define PHX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHX_VERSION 410 /* 0.2.0: phx_create_ex, phx_set_trnas, phx_tap_dist, host I/O; phx_globals and PHX_N_STAGES grew; 0.2.1: phx_run_async, phx_wait;
                         * 0.3.0: phx_certified, phx_globals.certified, PHX_S_BADTRNA, more PHX_CREATE_* flags;
                         * 0.4.0: phx_params.start_w_text (the struct grew), phx_params_from_flags, the exact re-solve moved below the ABI
                         *        (phx_download* deliver reference-exact genes; phx_certified reports 2), phx_dump_text, PHX_CREATE_NO_EXACT;
                         *        (0.4.0 also REMOVED phx_rbs_table — an ABI break for a C caller that bound it: the table was a test hook of the Python Decimal
                         *        replay, which moved to tests/decimal_replay.py and computes it itself);
                         * 0.4.1: phx_run_async puts the certificate kernels behind the run unless exactness is off; env PHX_FRONT_SPINS (tests) */
#define PHX_MAX_CODONS 16

/* library-level errors */
#define PHX_OK 0
#define PHX_E_ARG (-1)      /* bad argument */
#define PHX_E_NODEVICE (-10) /* no HIP device / device index out of range */
#define PHX_E_HIP (-11)      /* a HIP runtime call failed; see phx_last_error() */
#define PHX_E_NOMEM (-12)
#define PHX_E_STATE (-13) /* call sequence error (e.g. run before upload) */
#define PHX_E_PARAM (-14) /* codon tables must be 3 letters of acgt; 1..16 starts/stops; minlen >= 6 */
#define PHX_E_IO (-15)    /* a file could not be opened or read */

/* per-contig status (phx_result.status); the reference's behaviour in brackets */
#define PHX_S_OK 0
#define PHX_S_BADLETTER (-2) /* letter outside acgtnryswkmbvdh [KeyError, functions.py:20-24] */
#define PHX_S_TOOSHORT (-3)  /* L < 6 [UnboundLocalError/KeyError in GCframe.get, gc_frame_plot.py:53-69] */
#define PHX_S_BADTRNA (-4)   /* phx_set_trnas: a hit of this contig has an end outside 1..L (no node can be placed there) */
#define PHX_S_PARALLEL (-6)  /* a bridge edge duplicates a connect edge [ValueError, graphs.py:74] */
#define PHX_S_OVERFLOW (-7)  /* path sums exceed the widest integer kernel (1088 bit: an ORF weight beyond ~1e300) AND the contig was not solved on the host: phx_download* solve such a contig in the reference's own unbounded arithmetic (phx_certified then reports 2) unless the context was created with PHX_CREATE_NO_EXACT / NO_CERTIFY */
#define PHX_S_LONGORF (-8)   /* (no longer produced: until 0.3.0 an ORF of more than 65535 codons — 196 kb without an in-frame stop, a scaffold's N run —
                              * overflowed the 16-bit class counters; such ORFs are now counted in 32 bits, functions.py:286-298 has no limit) */
#define PHX_S_NEGCYCLE (-9)  /* relaxation did not converge in V rounds */
#define PHX_S_NOPATH 1       /* warning: target unreachable from source; 0 genes reported */

typedef struct phx_ctx phx_ctx;

/* Flags of the reference CLI that reach the path (file_handling.py:51-66, phanotate.py:42-44). */
typedef struct phx_params {
    int32_t minlen;                        /* -l/--minlen, default 90 */
    int32_t n_start;                       /* -s/--start_codons */
    char start[PHX_MAX_CODONS][4];         /* lower-case, NUL-terminated */
    double start_w[PHX_MAX_CODONS];        /* weight / max(weight), file_handling.py:58-62 */
    int32_t n_stop;                        /* -e/--stop_codons */
    char stop[PHX_MAX_CODONS][4];
    /* The weights as the user wrote them on -s (before the division by their maximum), e.g. "0.85", "0.10", "0.05": the reference
     * holds Decimal(text) / max (file_handling.py:58-62), 28 digits that a double cannot carry, and the exactness guarantee of
     * phx_download* / phx_certified is stated against THAT value.  An entry left empty means "the shortest decimal that reads back as
     * start_w[i]" (what Decimal(repr(x)) would be).  phx_default_params and phx_params_from_flags fill both. */
    char start_w_text[PHX_MAX_CODONS][32];
} phx_params;

/* One called gene = one ORF edge on the shortest path (phanotate.py:71-76, locus.py:29-37).
 * left/right are 1-based inclusive coordinates, left < right on both strands; the tabular writer
 * (locus.py:39-56) prints left,right for '+' and right,left for '-'. */
typedef struct phx_gene {
    int32_t left;
    int32_t right;
    int32_t strand; /* +1 / -1 */
    int32_t frame;  /* reference Node.frame of the left node: +-1..3 for a CDS, +-4 for a tRNA (left.gene == 'tRNA', phanotate.py:75;
                     * Locus.tabular prints CDS features only, locus.py:42) */
    double score;   /* ORF edge weight, what the reference prints with '%E' */
} phx_gene;

typedef struct phx_result {
    int32_t status;  /* PHX_S_* */
    int32_t n_genes;
    phx_gene *genes; /* library-owned; in path order (ascending right coordinate), as phanotate.py:71-76 adds them */
} phx_result;

/* ---- stage-tap records (parity tests; layout is part of the ABI) ---- */
typedef struct phx_orf {
    int32_t start; /* Orf.start (1-based; fwd: first base of the start codon, rev: leftmost base of its reverse complement) */
    int32_t stop;  /* Orf.stop  (dict key: fwd first base of the stop codon, rev leftmost base of the rc stop codon / 1,2,3) */
    int32_t frame; /* +-1..3 */
    int32_t length;
    int32_t rbs;      /* score_rbs bin 0..27 */
    int32_t startidx; /* index into params.start of Orf.start_codon(), or -1 */
    int32_t group;    /* stop-group rank in reference insertion order */
    int32_t hist[9];  /* GC-frame class counts, (max_idx-1)*3 + (min_idx-1) */
    double pstop;
    double weight_rbs;
    double S; /* sum over sense codons of pos_max[max_idx]*pos_min[min_idx] */
    double weight;
} phx_orf;

typedef struct phx_node {
    int32_t pos;
    int8_t type;  /* 0 start, 1 stop, 2 source, 3 target */
    int8_t frame; /* +-1..3 CDS, +-4 tRNA (Node.gene == 'tRNA', functions.py:500-505), 0 for source/target */
    int16_t pad;
    int32_t other; /* Orfs.other_end[pos] (tRNA nodes: other_end['t' + str(pos)]) as seen by get_graph (functions.py:363-371); -1 for source/target */
    int32_t refidx; /* rank of this node in the reference's Graph.iternodes() order */
    double o;      /* the o1/o2 term of functions.py:373-384 for this position */
} phx_node;

typedef struct phx_edge { /* in-edge list order of the device graph: grouped by dst */
    int32_t src, dst;     /* device node ids (position-sorted; source = V-2, target = V-1) */
    double w;             /* Decimal weight of the reference, in fp64 (the solver adds trunc(w * 1000), edges.py:22: the device keeps
                           * that integer, the tap recomputes w and checks the two against each other) */
    int32_t inexact;      /* 1: the reference's integer W* = trunc(Decimal(w) * 1000) is not known to equal the solver's W = trunc(w * 1000).
                           * Before the certificate has been asked for (phx_certified / phx_download* / phx_tap_globals): the verdict of the fp64
                           * pipeline; after it: of the double-double evaluation (csrc/phx_refine.inc), which clears most flags and leaves, for the rest, */
    int32_t pad;
    double d1, d2, err;   /* W* in [W + D - eps, W + D + eps], D = d1 + d2 (both integer-valued), eps = 0 if err == 0 else floor(err) + 1 (err = inf: unknown) */
} phx_edge;

typedef struct phx_globals {
    int64_t L;
    double pstop;               /* functions.py:178, also pgap (functions.py:309) */
    double background_rbs[28];  /* functions.py:180-181, normalised */
    double training_rbs[28];    /* functions.py:254-255, normalised */
    double pos_max[4], pos_min[4]; /* functions.py:281-284 */
    int32_t n_orf, n_group, n_node, n_edge, n_bridge;
    int32_t n_limbs;     /* 64-bit limbs of the integer kernel that solved this batch */
    int32_t sssp_sweeps; /* outer sweeps of the device relaxation */
    int32_t sssp_iters;  /* relaxation rounds summed over all windows and sweeps */
    int32_t status;
    int32_t sssp_kernel; /* which kernel solved it: 0 global memory, 1 workgroup per contig, 2 wavefront per contig, 3 the same in its roomy configuration */
    int32_t sssp_handed_back; /* != 0: the wavefront kernel passed the contig on: 1 a node's 500 bp neighbourhood exceeds a window,
                               * 2 spill list full, 3 no convergence, 4 too many step-backs, 5 its planner made no progress for 20 ms
                               * (small batches run the two side by side; never seen outside test builds) */
    int32_t tie; /* equal-length alternatives to the shortest path (the reference's relaxation order decides, see phx_inorder.inc):
                  * 0 none, 1 they exist and the solver's path already was the reference's, 2 the path was replaced by the reference's */
    int32_t certified; /* phx_certified's verdict for this contig (1 / 0 / -1) */
    /* the integer counters behind the fp64 globals above (what a Decimal restatement of the reference starts from, see
     * tests/decimal_replay.py): RBS bin counts without the pseudo-count (functions.py:155-156,168-169,211), GC-frame training
     * counts (functions.py:261-279, index 1..3), g+c over the contig after the counting remap of functions.py:159-163 */
    uint32_t rbs_background_count[28], rbs_training_count[28];
    uint32_t gc_max_count[4], gc_min_count[4];
    int64_t gc_count;
} phx_globals;

/* ---- library ---- */
int phx_version(void);
int phx_device_count(void);
const char *phx_strerror(int code);
/* Text of the last failing HIP call on this ctx (or on the last failed phx_create when ctx==NULL). */
const char *phx_last_error(const phx_ctx *ctx);

/* Fills *p with the reference defaults: atg:0.85,gtg:0.10,ttg:0.05 / tag,tga,taa / minlen 90. */
void phx_default_params(phx_params *p);
/* The same from the reference's flags (file_handling.py:51-66): start_codons "atg:0.85,gtg:0.10,ttg:0.05" (codon:weight pairs; a
 * repeated codon keeps its first place and its last weight, like the dict the reference builds), stop_codons "tag,tga,taa", minlen.
 * NULL strings mean the defaults.  Codons are lower-cased; PHX_E_PARAM on anything that is not 3 letters of acgt / a decimal number. */
int phx_params_from_flags(const char *start_codons, const char *stop_codons, int32_t minlen, phx_params *p);

/* device: HIP ordinal.  stream: a hipStream_t the caller owns, or NULL to let the ctx create its own (non-blocking: it
 * does NOT synchronise with HIP's null stream).  All kernels and copies of this ctx are issued on it. */
int phx_create(const phx_params *params, int device, void *stream, phx_ctx **out);
/* The same with flags.  PHX_CREATE_USE_STREAM: `stream` is used exactly as given, and a NULL handle then means HIP's null
 * (legacy default) stream — which is what torch.cuda.current_stream().cuda_stream is unless the caller switched streams —
 * so that work of this ctx is ordered after whatever the caller enqueued there before (e.g. the kernels that produce a
 * buffer handed to phx_attach).  On the null stream the run is enqueued kernel by kernel (HIP cannot capture it into a graph). */
#define PHX_CREATE_USE_STREAM 1u
/* Development / test switches (results are the same with any of them): enqueue every run kernel by kernel instead of replaying a
 * captured HIP graph; size the buffers between kernels on every run (two host round trips) instead of only on the first; solve every
 * contig with the global-memory shortest-path kernel; keep contigs off the wavefront-per-contig kernel. */
#define PHX_CREATE_NO_GRAPH 2u
#define PHX_CREATE_SIZE_EVERY_RUN 4u
#define PHX_CREATE_SOLVER_GLOBAL 8u
#define PHX_CREATE_SOLVER_NO_WAVE 16u
/* NO_CERTIFY: phx_certified reports -1 (and the kernel's scratch, 20 + 32 bytes per node, is not allocated).  It also turns the
 * exactness machinery off as a whole — without a certificate nothing says which contigs would need the host re-solve —: the gene lists
 * of phx_download* are then the fp64-derived integers' shortest paths, WITHOUT the guarantee that they are the reference's.
 * CERT_TIGHT multiplies its error bounds by 2^36, so that ordinary inputs come out uncertified (tests of the host re-solve). */
#define PHX_CREATE_NO_CERTIFY 32u
#define PHX_CREATE_CERT_TIGHT 64u
#define PHX_CREATE_POISON 256u    /* every device buffer the context allocates is filled with the byte 0xA5 first: the library must not depend on fresh memory being zero */
#define PHX_CREATE_ONE_STREAM 512u /* no side streams: every kernel of a run on the context's one stream, in program order */
#define PHX_CREATE_CERT_WIDE 128u /* every contig through the certificate's general kernel (otherwise only contigs of more than 12288 nodes) */
#define PHX_CREATE_NO_DUO 4096u /* 128-bit contigs are solved by k_sssp_wave<2> (one wavefront per contig) instead of k_sssp_duo (a feeder and a solver wavefront per contig) */
#define PHX_CREATE_NO_SEG 8192u /* small batches solve every contig by one sweep (one wavefront pair), not by up to 16 segments side by side that k_seg_merge joins and proves (phx_sssp_seg.inc) */
#define PHX_CREATE_NO_FUSE 2048u /* batches of up to 4 contigs and 40 kb in all run their front end (ORF count ... edge fill) as the staged kernels of large batches, not as the one fused launch (k_front) */
#define PHX_CREATE_NO_EXACT 1024u /* phx_download* hand out the device's gene lists as they are: no certificate is asked for and no contig is solved again on the host */
int phx_create_ex(const phx_params *params, int device, void *stream, uint32_t flags, phx_ctx **out);
void phx_destroy(phx_ctx *ctx);

/* ---- the whole path, batched (replaces phanotate.py:40-76 for n contigs) ---- */
/* seq[i]: ASCII, any case, len[i] characters, not NUL-terminated; borrowed for the call. */
int phx_annotate(phx_ctx *ctx, int32_t n, const char *const *seq, const int64_t *len, phx_result *out /* [n] */);
void phx_free_results(phx_result *res, int32_t n);

/* The same in three steps, so a caller can keep inputs resident in HBM and time phx_run alone. */
/* phx_upload: the letters are packed on the host — worker threads, into pinned memory — into the form the kernels read (residue-split
 * bit planes, 3 bits per base: DESIGN.md §3) and copied piece by piece; it returns when the caller's strings are free again, the copies
 * are ordered before the run on the context's stream. */
int phx_upload(phx_ctx *ctx, int32_t n, const char *const *seq, const int64_t *len);
/* Alternative to phx_upload: the concatenated ASCII is already in device memory (offsets on host, n+1 entries); a kernel at the head
 * of every run packs it (the caller's buffer is only read, and must stay valid until the runs on it have ended). */
int phx_attach(phx_ctx *ctx, int32_t n, const void *d_ascii, const int64_t *offsets);
/* tRNA masking (functions.add_trnas, functions.py:457-509): the hits of an external tRNA finder for the contigs of the batch just
 * uploaded / attached, as the reference holds them in `trnas`: hits of contig i are (start[k], stop[k]) for k in
 * [offsets[i], offsets[i+1]); start < stop for a hit on the forward strand, start > stop (the pair reversed) for a complement hit;
 * 1-based, inside the contig (a contig with a hit outside 1..L gets status PHX_S_BADTRNA; the batch goes on).  The device then adds the tRNA nodes (frame +-4), their edges of weight -20 and the connect rules
 * of functions.py:388-399.  Call after phx_upload / phx_attach and before phx_run; a new upload forgets the hits.  Not calling
 * it (or offsets == NULL) is "no tRNA finder installed" (functions.py:493-495).  Running the finders is the caller's business
 * (phanotate_amd/trna.py does what functions.py:457-491 does). */
int phx_set_trnas(phx_ctx *ctx, const int64_t *offsets /* [n+1] */, const int32_t *start, const int32_t *stop);
int phx_run(phx_ctx *ctx);                        /* every kernel of the path; blocks until results are in HBM */
/* The same in two halves, for keeping two batches in flight on one GPU (two contexts on their own streams: while the latency-bound
 * shortest-path kernel of one batch runs, the other context's upload and throughput kernels fill the device; on the benchmark
 * batch two contexts alternating take 1.8 ms per batch instead of 2.3).  phx_run_async enqueues the run and returns; phx_wait
 * blocks until the results are in HBM and does what a run that did not fit needs (buffers grown, run repeated).  The first run
 * of a context is synchronous (it sizes the buffers between kernels).  Every other entry point on a context with a run in flight
 * waits for it first, so the pair is an optimisation, never a requirement.  phx_wait without a run: PHX_OK if results are there.
 * Unless exactness is off (PHX_CREATE_NO_EXACT / NO_CERTIFY, phx_set_exact(ctx, 0)) phx_run_async also enqueues the certificate kernels
 * (k_refine, k_certify) behind the run on the context's stream: the download that follows finds the certificate done instead of waiting
 * for it, and with two batches in flight it runs beside the other context's kernels (a stream of batches host to host: 2.6 -> 2.0 ms
 * per 1000 x 50 kb).  phx_run never does: the certificate stays on demand there (phx_certified / phx_download*). */
int phx_run_async(phx_ctx *ctx);
int phx_wait(phx_ctx *ctx);
/* Both download calls deliver gene lists that are the reference's: they ask for the certificate (phx_certified) and a contig the
 * device could not certify is solved again on the host, inside this call, on the reference's own Decimal-derived integers
 * (csrc/phx_exact.inc; worker threads, one contig each) — expected for none of a batch's contigs, see phx_certified.  A context
 * created with PHX_CREATE_NO_CERTIFY or PHX_CREATE_NO_EXACT skips both and hands out the device's lists. */
int phx_download(phx_ctx *ctx, phx_result *out);  /* D2H of the gene lists, [n] */
/* The same into caller-owned flat arrays (what a language binding wants: no per-contig allocation): genes of contig i are
 * genes[offsets[i] .. offsets[i+1]), in path order; status[i] as phx_result.status.  offsets has n+1 entries.  With
 * genes == NULL only offsets, status and total are filled (size query); cap = number of phx_gene records genes can take. */
int phx_download_flat(phx_ctx *ctx, phx_gene *genes, int64_t cap, int64_t *offsets /* [n+1] */, int32_t *status /* [n] */, int64_t *total);

/* ---- one process, several GPUs (contigs never interact, phanotate.py:40,56) ----
 * A pool = one host thread and two contexts per listed device (an ordinal may repeat).  phx_pool_annotate cuts the n contigs into
 * consecutive batches of at most batch_bases bases (<= 0: 4e8; with several devices at least two batches per device), sends batch k
 * to device k mod n_dev with two batches in flight per device (phx_run_async), and fills out[0..n) in input order exactly as
 * phx_annotate would (the reference's genes, see phx_download).  No process group, no collective.  trna_*: as phx_set_trnas for the
 * whole input (offsets has n + 1 entries), or NULL.  On an error every result is freed and the first error code is returned.
 * flags: PHX_CREATE_* without USE_STREAM. */
typedef struct phx_pool phx_pool;
int phx_pool_create(const phx_params *params, int32_t n_dev, const int32_t *devices, uint32_t flags, phx_pool **out);
void phx_pool_destroy(phx_pool *pool);
const char *phx_pool_last_error(const phx_pool *pool);
int phx_pool_annotate(phx_pool *pool, int32_t n, const char *const *seq, const int64_t *len, int64_t batch_bases,
                      const int64_t *trna_offsets, const int32_t *trna_start, const int32_t *trna_stop, phx_result *out /* [n] */);

/* Is every gene list what the REFERENCE'S integers give?  The reference solves on trunc(Decimal(w) * 1000) with 28 digits
 * (edges.py:17-23); the device derives its integers in fp64 and, where fp64 cannot decide the truncation, in double-double, and
 * flags the edges whose integer it still cannot prove equal to the reference's (the error bound of the Decimal chain
 * itself).  After the solve it proves, per contig and in exact integer arithmetic, that no difference inside those bounds can change
 * the path (an optimality certificate, csrc/phx_certify.inc).  cert[i] =
 *   1: proven on the device (also for contigs with an error status or without a path);
 *   2: not proven on the device, so the contig was solved again on the host on the reference's own integers (the Decimal chain
 *      replayed by csrc/phx_dec.c + phx_exact.inc) and phx_download* deliver THAT result;
 *   0: not proven and not solved again (context created with PHX_CREATE_NO_EXACT, or the replay met an operation it does not restate:
 *      never on the reference's inputs) — the genes are the exact solution for the device's integers;
 *  -1: the context was created with PHX_CREATE_NO_CERTIFY.
 * The proof is computed when it is first asked for after a run (here, by phx_download*, or by phx_tap_globals), from the state the
 * run left on the device — ~0.2 ms for a thousand 50 kb contigs; phx_run itself does not pay for it. */
int phx_certified(phx_ctx *ctx, int8_t *cert /* [n] */);
/* on = 0: phx_download* / phx_certified stop short of the host re-solve (as PHX_CREATE_NO_EXACT) and hand out the device's lists
 * for every contig, also for those already solved again; on = 1 (the default) switches it back.  For tests and measurements. */
int phx_set_exact(phx_ctx *ctx, int on);

/* ---- stage taps on the batch last processed by phx_run (parity tests) ---- */
int phx_tap_globals(phx_ctx *ctx, int32_t contig, phx_globals *out);
/* per 0-based position, each array L bytes (any may be NULL):
 *   cls  bits0-2 codon class at p (0 none,1 fwd start,2 rev start,3 fwd stop,4 rev stop; elif order of functions.py:198-215)
 *   gcc  low nibble fwd (max_idx-1)*3+(min_idx-1) of gc_pos_freq[p+1], high nibble the reversed triple
 *   binF/binR  score_rbs of dna[p-20:p+1] / rev_comp(dna[p:p+21]) */
int phx_tap_positions(phx_ctx *ctx, int32_t contig, uint8_t *cls, uint8_t *gcc, uint8_t *binF, uint8_t *binR);
int phx_tap_orfs(phx_ctx *ctx, int32_t contig, phx_orf *out /* [n_orf], reference iter_orfs order */);
int phx_tap_nodes(phx_ctx *ctx, int32_t contig, phx_node *out /* [n_node], device order */);
int phx_tap_edges(phx_ctx *ctx, int32_t contig, phx_edge *out /* [n_edge] */);
/* path as device node ids, source first; dist_limbs receives n_limbs 64-bit words (two's complement) */
int phx_tap_path(phx_ctx *ctx, int32_t contig, int32_t *path, int32_t cap, int32_t *n_path, uint64_t *dist_limbs, int32_t cap_limbs);
/* exact distance of every node from the source: n_node x n_limbs 64-bit words (two's complement, device node order);
 * an unreached node has a top word >= 2^61 */
int phx_tap_dist(phx_ctx *ctx, int32_t contig, uint64_t *dist_limbs, int64_t cap_words);

/* -d/--dump of the reference (phanotate.py:58,61) for one contig of the batch last run: one line per edge of its graph,
 *     repr(source) TAB repr(target) TAB str(weight * 1000)                                   (edges.py:17-23, nodes.py:14-21)
 * in Graph.iteredges order, the weights as the reference's 28-digit Decimals (the chain replayed by csrc/phx_dec.c on the integers
 * the device delivers, csrc/phx_exact.inc) — byte for byte what an upstream install prints.  *text is malloc'ed (NUL-terminated),
 * release with phx_free_text.  Host-side formatting, ~0.3 s for a 170 kb genome. */
int phx_dump_text(phx_ctx *ctx, int32_t contig, char **text, int64_t *text_len);

/* ---- the solver alone (the fastpathz boundary, phanotate.py:56-64) ----
 * Edges (src[i] -> dst[i]) over nodes 0..V-1 with integer weights given as n_limbs little-endian
 * 64-bit words each (two's complement).  Writes the node ids of the shortest path source..target to
 * path_out (cap entries) and its length to *n_path (0 if unreachable).  V < 2^29 (PHX_E_ARG beyond). */
int phx_solve(phx_ctx *ctx, int32_t V, int32_t E, const int32_t *src, const int32_t *dst, const uint64_t *w_limbs,
              int32_t n_limbs, int32_t source, int32_t target, int32_t *path_out, int32_t cap, int32_t *n_path,
              uint64_t *dist_limbs);

/* ---- measurement ---- */
#define PHX_N_STAGES 14
/* When on, every kernel launch of phx_run is bracketed by hipEvents on the ctx stream. */
int phx_set_profiling(phx_ctx *ctx, int on);
/* The same for a subset of the stages (bit k = stage k; 0 switches profiling off): two events per run instead of two per stage. */
int phx_set_profiling_stages(phx_ctx *ctx, uint32_t stage_mask);
/* ms[k] = accumulated GPU time of stage k since the last reset; names via phx_stage_name(k).  (Stage 10 was "edge_weights" and keeps its
   index and now times k_refine + k_certify: the overlap weights are evaluated inside the edge fill.  Stage 13, "wave_plan", runs on a side
   stream BESIDE stage 8, "edges_fill": it is in the table, not in the sum of the main stream's stages.) */
int phx_get_stage_ms(phx_ctx *ctx, float *ms /* [PHX_N_STAGES] */, int32_t *launches /* [PHX_N_STAGES] */, int reset);
const char *phx_stage_name(int k);
/* Contigs, over the life of the context, whose shortest-path wavefront was launched beside the planner of its windows (batches of up to
 * 3/4 of the device's SIMDs: DESIGN.md §4), saw no progress from it for ~20 ms and handed the contig to the workgroup kernel instead
 * (phx_globals.sssp_handed_back == 5; the results are the same).  That only happens when the planner's wavefronts were not resident
 * beside the solver's — other contexts or processes holding the SIMDs —, and after the first such run the context launches the solver
 * behind its planner.  0 on an undisturbed GPU. */
int64_t phx_plan_timeouts(phx_ctx *ctx);
/* Runs of this context whose front end was the single fused launch of small batches (k_front: steady-state runs of up to 4 contigs and 40 kb in all).
 * Negative (-(runs) - 1): that kernel once waited ~4 ms at a grid barrier because its workgroups were not all resident (the GPU was
 * shared), the run was repeated with the staged kernels, and the context has used those since. */
int64_t phx_front_runs(phx_ctx *ctx);
/* Runs of this context whose 128-bit contigs were solved in segments (batches of up to 32 contigs with a contig of 10 kb or more: a contig's
 * shortest path by up to 32 wavefront pairs side by side in frames of their own, joined by a constant each and PROVEN by one pass over the
 * edges — k_seg_join / k_seg_close, phx_sssp_seg.inc).  A contig whose segments cannot be joined or proven (about 2 % of random contigs:
 * every path downstream runs over an ORF edge from in front of a segment's margin) is solved by the one-sweep kernels in the same run —
 * phx_seg_fallbacks counts them — and later runs of that batch take the one-sweep kernels.  Either way the delivered distances, parents
 * and genes are the one-sweep solver's bit for bit.  PHX_CREATE_NO_SEG / env PHX_NO_SEG=1: never. */
int64_t phx_seg_runs(phx_ctx *ctx);
int64_t phx_seg_fallbacks(phx_ctx *ctx);
/* development: the per-segment records of contig i in the run last made (8 ints each: windows | done flag, solver status, first node, end node, the solver's time in
 * 10 ns ticks, phases, packs taken, step-backs); returns the number of records (0: that run did not use segments) */
int phx_seg_stats(phx_ctx *ctx, int32_t i, int32_t *out, int32_t cap_records);
/* sizes of the batch last run: positions, ORFs, nodes, edges (for the algorithmic-byte formula) */
int phx_batch_sizes(phx_ctx *ctx, int64_t *L, int64_t *n_orf, int64_t *n_node, int64_t *n_edge);

/* ---- host utilities (no device needed) ---- */
/* Deterministic synthetic phage-like contig: exactly L lower-case acgt chars. */
int phx_synth_contig(uint64_t seed, int64_t L, char *out);
/* The reference's number type on the host (csrc/phx_dec.c: Python's decimal.Decimal at prec 28, ROUND_HALF_EVEN), for the tests:
 * op = "add" "sub" "mul" "div" "pow" "ln" "exp" on the decimal texts a (and b) at `prec` digits, "str" (a as Decimal.__str__ prints
 * it), "float" (Decimal(float(a))), "repr" (repr(float(a))), "trunc1000" (int(a * 1000) as 18 hex words), "dd" (a as a double-double).
 * The result text goes to out; returns its length or a negative error. */
int phx_dec_eval(const char *op, const char *a, const char *b, int prec, char *out, int cap);
/* The double-double arithmetic of k_refine (csrc/phx_dd.h; it compiles for the host too), for the tests: op = "add" "sub" "mul" "div"
 * "exp" "log" on (ah + al) and (bh + bl), "repr" = the decimal value Decimal(repr(ah)) holds; phx_dd_shortest: the digits and decimal
 * exponent of repr(x) for 1e-10 <= x <= 1e10 (returns the number of digits, 0 outside that range). */
int phx_dd_eval(const char *op, double ah, double al, double bh, double bl, double *rh, double *rl);
int phx_dd_shortest(double x, uint64_t *digits, int32_t *exp10);

/* ---- host I/O of the CLI (no device needed; phx_host.c) ----
 * Files and texts beyond a few MB are parsed / formatted by worker threads (one per online core, at most 16; the environment
 * variable PHX_HOST_THREADS overrides the count): the results do not depend on it. */
/* FASTA, plain or gzip, read whole: what phanotate.py:32-35 gets from the external `genbank` package.  A record's name is the
 * first token of its header line (README.md:45); sequence lines are stripped of surrounding white space and joined, case kept
 * (the kernels lower-case); text before the first header is ignored. */
typedef struct phx_fasta phx_fasta;
int phx_fasta_read(const char *path, phx_fasta **out);
int32_t phx_fasta_count(const phx_fasta *f);
/* name: NUL-terminated; seq: *len characters, NOT terminated; both owned by f */
int phx_fasta_record(const phx_fasta *f, int32_t i, const char **name, const char **seq, int64_t *len);
/* all records at once, in the form phx_upload / phx_format_tabular take (any array may be NULL) */
int phx_fasta_arrays(const phx_fasta *f, const char **names, const char **seqs, int64_t *lens);
void phx_fasta_free(phx_fasta *f);
/* The reference's default output (Locus.tabular, locus.py:39-56) for n contigs from the flat arrays of phx_download_flat:
 * "#id:\t<name>", the column header, then START STOP FRAME CONTIG SCORE per gene ('%E' score; START > STOP on the reverse
 * strand).  Contigs with a negative status are skipped.  *text is malloc'ed (NUL-terminated), release with phx_free_text. */
int phx_format_tabular(int32_t n, const char *const *names, const phx_gene *genes, const int64_t *offsets, const int32_t *status, char **text, int64_t *text_len);
void phx_free_text(char *text);

#ifdef __cplusplus
}
#endif
#endif /* PHX_H */
