// phx_internal.h — device/host shared layouts of libphx (not part of the public ABI).
//
// Data layout in HBM (one batch = n contigs, concatenated):
//   per position:  the bases as residue-split bit planes (recs: 36-byte records of 96 positions, phx_features.inc); everything else
//       that is a predicate of a position or codon lives in bitmaps (bits, nbits: DESIGN.md §3)
//   per ORF   (index = meta.orf_off + k):  DOrf head (16 B), DOrfStat (32 B), weight f64, start-node id i32 — four arrays, each
//       written whole by one kernel (k_orf<true>, k_orf_stats, k_score, k_node_build)
//   per group (index = meta.grp_off + g, device order = (strand, frame, codon); DGrp.evkey = reference insertion order):  DGrp (32 B)
//   per node  (index = meta.node_off + v, v sorted by position; source = V-2, target = V-1):
//       DNode {pos i32, info i32, link u32, other i32}, no f64, in_off u32 (+1), dist NL x u64, parent i32
//   per edge  (index = meta.edge_off + e, grouped by destination node):  esrc u32 (source node | inexact << 31), ew i64 (the integer
//       weight trunc(w * 1000), encoded: see ew_encode in phx_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/phx.h"

#define PHX_TILE 1536       // a contig's bitmaps cover whole tiles of this many positions = 8 bitmap words (64 codons) per frame
#define PHX_CTG_THREADS 256 // per-contig workgroup kernels
#define PHX_N_CODON_BITMAPS 12 // 0-3 fwd start, rev start, fwd stop, rev stop; 4-9 GC frame plot: the comparisons a > b, b > c, a > c of a codon's three window counts and c > b, b > a, c > a for the reversed triple (max_idx / min_idx follow, see gc_planes); 10-11 'atg' on the forward / reverse strand
#define PHX_PLANE_GCF 4  // first of the three forward comparison planes
#define PHX_PLANE_GCR 7  // ... of the reversed triple
#define PHX_PLANE_ATG 10 // 'atg' forward, then reverse
#define PHX_PRE_G 2 // bitmap words per prefix-popcount record (k_bit_prefix; a power of two <= 8: nw is a multiple of 8)
#define PHX_BITMAP_WORDS_PER_NW (PHX_N_CODON_BITMAPS * 3) // codon bitmaps x 3 frames (the bases themselves: DBatch.recs)

// codon classes, in the elif order of functions.py:198-215
#define CLS_NONE 0
#define CLS_FS 1 // codon in start_codons
#define CLS_RS 2 // rev_comp(codon) in start_codons
#define CLS_FT 3 // codon in stop_codons
#define CLS_RT 4 // rev_comp(codon) in stop_codons
// cls byte (DParams.cls_tab, the cls tap; no per-position array on the device): bits0-2 class, bits3-6 start-codon index (FS: of codon, RS: of rc codon), bit7 rc(codon) in start_codons

// A gap edge whose weight is an entry of the contig's gap table (score_gap depends on the length and the strand pair only,
// functions.py:36-46): bit 29, the table index ("code") in bits 19-28, the source node in bits 0-18 — such an edge has NO entry in
// DBatch.ew (round 6: 69 % of the edges; k_edges<true> wrote 12 bytes for each).  Every other edge: source node in bits 0-28.
#define ESRC_F_GAP 0x20000000u
#define ESRC_GAP_NODES (1 << 19) // contigs with more nodes keep every edge in the explicit form
#define ESRC_IS_GAP(x) (((x) >> 29) & 1u)
#define ESRC_GAP_CODE(x) (((x) >> 19) & 1023u)
#define ESRC_GAP_WORD(src, code) ((uint32_t)(src) | ESRC_F_GAP | ((uint32_t)(code) << 19))
static inline __host__ __device__ uint32_t esrc_node(uint32_t x) { return x & ((x & ESRC_F_GAP) ? 0x7ffffu : 0x1fffffffu); }
#define ESRC_NODE(x) esrc_node(x)
#define GT_SAME 500 // gap table (DBatch.gtab, GT_N entries per contig): 0..499 same strand, length -2..497 (beyond 300: (1-pgap)^100 + length);
#define GT_DIFF 303 //   500..802 across strands (+20), length -2..300
#define GT_N 808
#define ESRC_INEXACT(x) ((x) >> 31)
#define ESRC_OFFPATH(x) (((x) >> 30) & 1u) // not for k_sssp_wave's common path: |W| >= 2^51, or the edge leaves the source node
#define ESRC_F_INEXACT 0x80000000u
#define ESRC_F_OFFPATH 0x40000000u

// node link word: bits 30-31 kind, bits 0-29 index
#define LINK_NONE 0u
#define LINK_START (1u << 30) // payload = ORF index (contig-relative)
#define LINK_STOP (2u << 30)  // payload = group index (contig-relative)
#define LINK_TRNA (3u << 30)  // payload = index of the tRNA node (contig-relative, DBatch.tnode)
#define LINK_IDX(x) ((x)&0x3fffffffu)
#define LINK_KIND(x) ((x)&0xc0000000u)

// ninfo: bits0-1 type (0 start,1 stop,2 source,3 target), bits 8-15 frame as int8
#define NINFO(type, frame) ((int32_t)(((uint32_t)(type)&3u) | (((uint32_t)(uint8_t)(int8_t)(frame)) << 8)))
#define NTYPE(i) ((i)&3)
#define NFRAME(i) ((int)(int8_t)(((uint32_t)(i) >> 8) & 0xff))

struct DParams { // device copy of phx_params + derived tables
    int32_t minlen;
    int32_t n_start;
    double start_w[PHX_MAX_CODONS];
    uint8_t cls_tab[72];  // codon code (c0 | c1<<2 | c2<<4) -> cls byte; entry 64 = no codon (0)
    uint8_t atg_tab[72];  // bit0: codon == 'atg', bit1: codon == 'cat', bits 2..5: start-codon index (as cls_tab bits 3..6); entry 64 = 0
    double sw_hi[PHX_MAX_CODONS], sw_lo[PHX_MAX_CODONS]; // the reference's start weights, Decimal(text) / max rounded to 28 digits (file_handling.py:58-62), as double-doubles (k_refine)
};

struct DOrf { // what the scan knows about an ORF: one 16-byte store by k_orf<true>
    int32_t start, stop; // reference Orf.start / Orf.stop (1-based)
    int8_t frame;        // +-1..3
    uint8_t pad[2];
    uint8_t flags;       // bit1: pseudo-start (no start codon)
    int32_t grp;         // contig-relative group index
};
struct DOrfStat { // its statistics and the per-position records of its start codon: two 16-byte stores by k_orf_stats
    uint16_t hist[9];    // GC frame class histogram over the sense codons
    uint8_t rbs;         // score_rbs bin
    int8_t startidx;     // index into params.start or -1 (pseudo-start)
    uint8_t flags;       // bit0: Orf.start_codon() == 'atg'; bit1: pseudo-start
    uint8_t pad[3];
    double pstop;        // Orf.p_stop, orfs.py:162-173
};

struct DGrp {
    int32_t stop;      // dict key
    int32_t orf_begin; // contig-relative index of the first ORF; ORFs are contiguous, nearest start first
    int32_t n;
    int32_t node; // device node id of the stop node
    int32_t frame;
    int32_t evkey; // position of the discovery event (0-based) or L + k for the k-th end fragment: reference insertion order
    int32_t pick;  // the longest ORF of the group whose start codon is 'atg' (index within the group; -1: none): the one the GC frame plot trains on, functions.py:261-279
    int32_t pad0;
};
static_assert(sizeof(DGrp) == 32, "group record");

struct DBridge {
    int32_t last, base;
};

// tRNA masking (functions.py:497-509): the nodes and edges add_trnas inserts, prepared on the host by phx_set_trnas
struct DTNode {
    int32_t pos;   // Node.position
    int32_t info;  // NINFO(type, +-4)
    int32_t other; // other_end['t' + str(pos)] as get_graph sees it (last writer wins)
    int32_t rank;  // insertion rank among the tRNA nodes (they follow every CDS node, functions.py:357)
};
struct DTEdge {
    int32_t src, dst; // tRNA node indices (contig-relative); weight -20 (functions.py:509)
};

struct DMeta { // one per contig
    int64_t off; // position offset into per-position arrays
    int32_t L;
    int32_t status;
    uint32_t gc;       // g+c after the counting remap of functions.py:159-163
    uint32_t bg[28];   // background RBS bin counts (no pseudo-count)
    uint32_t tr[28];   // training RBS bin counts
    uint32_t pmax[4];  // GC-frame training counts, index 1..3
    uint32_t pmin[4];
    int32_t n_orf, n_grp;
    int64_t orf_off, grp_off;
    int32_t n_node; // including source and target
    int32_t n_edge;
    int64_t node_off, edge_off;
    int32_t n_bridge;
    int32_t maxexp; // max binary exponent of |trunc(w*1000)| over the ORF edges
    int64_t bridge_off;  // this contig's slice of DBatch.bridge (layout, set on the host)
    int32_t bridge_cap;  // = L/500 + 2: a bridge needs more than 500 uncovered bases
    int32_t n_win;       // windows k_wave_plan laid out for k_sssp_wave (its slice starts at win_off, see k_layout1)
    int32_t n_genes;
    int32_t n_path;
    int64_t gene_off;
    int32_t sweeps;
    int32_t nw;            // bitmap words per (class, frame) = 8 * number of feature tiles
    int64_t bits_off;      // offset (in 64-bit words) of this contig's 12 bitmaps
    int64_t item_off;      // offset of this contig's 6*nw scan items
    int64_t cb_off;        // offset (64-bit words) of this contig's close-node bitmaps
    int32_t ncw;           // words per close-node bitmap = n_node/64 + 1
    int32_t sssp_fb;       // kernel (0 / 1) that takes the contig over if the wavefront kernel (mode 2) hands it back
    int64_t nbits_off;     // offset (64-bit words) of this contig's 9*nw node/coverage bitmap words; nbase offset = nbits_off/3
    int32_t n_orf_main, n_grp_main; // ORFs / groups emitted by the main loop (the end fragments follow)
    int32_t sssp_nl;   // 64-bit limbs this contig's path sums need (2, 4, 8 or 17)
    int32_t sssp_iters;
    int32_t sssp_why;  // why k_sssp_wave handed the contig back: 1 window limits, 2 spill list, 3 no convergence, 4 too many step-backs, 5 no progress of the planner it follows (0: it did not)
    int32_t sssp_mode; // 0 = global-memory kernel, 1 = workgroup-per-contig LDS kernel, 2 = wavefront-per-contig kernel, 3 = that kernel's roomy configuration,
                       // 4 = no device kernel: the path sums exceed 1088 bits, the host solves the contig (phx_exact.inc)
    int32_t n_open;    // entries of olist that are open nodes (incl. the target); the close nodes follow
    int32_t dense;     // some node has more than ~62 close / open nodes within the next 500 bp: k_sssp_wave's windows cannot take it (k_edges<false>)
    double wsum;       // sum of |w*1000| over the ORF edges (fp64, order-dependent rounding: used as a bound only)
    int64_t win_off;   // first window record of this contig in DBatch.win (capacity n_node/16 + 7)
    int32_t n_tnode, n_tedge; // tRNA nodes / edges of this contig (phx_set_trnas; layout, set on the host)
    int64_t tn_off, te_off;
    int32_t tie;       // k_inorder: 0 the shortest path is unique, 1 equal-length alternatives exist and the solver's path is the in-order one,
                       // 2 the path was replaced by the in-order one, -1 not resolved (cannot happen)
    int32_t cert;      // k_certify: 1 the path is proven to be the one the reference's Decimal-derived integers give, 0 not proven (phx_certify.inc)
    int32_t plan_prog; // DBatch.plan_stream: windows whose records k_wave_plan<2,0> has published (| WV_PLAN_DONE when it has finished; -1: it
                       // gave the contig up) — k_sssp_wave<2,0> runs beside the planner and consumes the windows as they appear
    int32_t seg_fail;  // DBatch.seg: the segments of this contig could not be joined or proven (k_seg_join / k_seg_close): one sweep solves it in the same run (phxk_seg_fallback)
    int64_t rec_off;   // first record of this contig in DBatch.recs (2 nw + 1 records: the last one is all "outside")
};

// What the host needs of a contig after every run (the full DMeta record, 0.5 KB, comes over only when a tap asks for it)
struct DRes {
    int32_t status, n_genes;
    int64_t gene_off;
    int32_t cert, tie; // DMeta.cert, DMeta.tie
};

// k_refine -> k_certify, per edge that is still flagged "inexact": the reference's integer W* lies in [W + D - eps, W + D + eps], D = d1 + d2
// (two integer-valued doubles), eps = 0 if err == 0, else floor(err) + 1; err = infinity: nothing is known (the contig is not certified)
struct DERef {
    double d1, d2, err;
};

struct DGene {
    int32_t left, right, strand, frame;
    double score;
};

// One node of the ORF graph (position-sorted; forward-strand slot before reverse-strand slot at equal positions).
struct DNode {
    int32_t pos;    // 1-based position (source 0, target L+1)
    int32_t info;   // NINFO(type, frame)
    uint32_t link;  // LINK_START | ORF index  or  LINK_STOP | group index
    int32_t other;  // Orfs.other_end[pos] as get_graph sees it (k_node_attr)
};

// Everything the kernels need, passed by value.
// One window of k_sssp_wave's sweep (written by k_wave_plan): nodes [v0, v1) iterate, [v0, va) become final.
struct DWin {
    int32_t v0, va, v1;
    uint32_t e0;     // first in-edge of v0 (contig-relative)
    uint32_t ne;     // in-edges of [v0, v1)
    uint32_t pa, pb; // phase A (close nodes) / B (open nodes): lanes in use | largest per-lane in-edge count << 8 | most helper lanes of a node << 16
    uint32_t sp;     // spill entries of the window
};
#define WIN_ROLES 128 // lane records per window: 64 of phase A, 64 of phase B (uint2 each, see wv_role_pack)

// Batch totals and decisions computed on the device (k_layout1 / k_layout2), so that a run needs no host round trip
// when the buffers of the context are already large enough; the host reads them back with the final results.
struct DTotals {
    int64_t orf, grp, node, cb, edge; // totals over the batch
    int32_t nlmax;       // widest integer class (64-bit limbs) any contig needs
    int32_t class_mask;  // bit 4*k + mode: some contig wants limb class k (2,4,8,17 limbs) solved by kernel `mode`;
                         // bit 16 + k: class k has contigs that were routed to the workgroup kernel from the start (dense)
    int32_t vmax;        // largest node count
    int32_t overflow;    // bit 0: ORF/node buffers, bit 1: edge/distance buffers too small -> the later kernels do nothing;
                         // bit 2: k_inorder's scratch (DBatch.tie) too small: tie_need bytes are wanted
    int64_t lds_need[4]; // per limb class: dynamic LDS k_sssp_lds needs (max over the contigs it may get)
    int64_t tie_need;    // bytes of scratch the contigs with equal-length alternative paths asked for (k_inorder)
    int32_t plan_timeouts; // contigs whose k_sssp_wave gave up waiting for the planner it follows (DMeta.sssp_why 5) in this run: the host
    int32_t front_abort;   //   then stops launching the solver beside its planner on this context (phx_plan_timeouts)
    uint32_t gsync;        // k_front (small batches: the front end in one launch): arrivals at its grid barriers; front_abort: a workgroup
                           //   waited too long for the others (not all resident): the host runs the batch again with the staged kernels
    int32_t seg_abort;     // segments, and a contig neither they nor the one sweep behind them could take (more nodes than k_seg_close's table; windows the tight
                           //   planner cannot lay out): the host runs the batch again without segments
    int32_t seg_fallbacks; // contigs whose segments could not be joined or proven in this run (solved by one sweep behind them)
    int32_t pad_t2;
};
struct DCaps {
    int64_t orf, grp, node, cb, edge; // elements the buffers of the context hold
    int64_t win;                      // window records (and WIN_ROLES lane records each)
    int32_t limbs;                    // 64-bit words per node in `dist`
    int32_t flags;                    // bit 0: force the global-memory solver, bit 1: keep contigs off the wavefront kernel, bit 2: the host sizes the
                                      //   buffers after each layout kernel (first run of a context): totals beyond the capacities are no overflow
};

struct DBatch {
    int32_t n_contig;
    int64_t mean_len;   // mean contig length of the batch (launch geometry of the per-contig kernels)
    DMeta *meta;
    DTotals *tot;
    DRes *res;          // per contig: status, gene count, first gene record (k_results, the last kernel of a run)
    int32_t *sord;      // batches beyond one contig per SIMD: contig of workgroup i of k_sssp_wave, most in-edges first (k_sssp_order); else null
    int64_t *lpart;     // k_layout*_a -> _b: per workgroup of 256 contigs the four totals (batches beyond 1024 contigs)
    DCaps caps;
    const DParams *params;
    // per position
    uint32_t *recs;       // the bases as residue-split bit planes: record 0 is a pad, contig i owns the records DMeta.rec_off .. rec_off + 2 nw, one
                          //   more pad at the end (phx_features.inc; written by phx_upload's copies or by k_pack_planes)
    const uint32_t *voff; // per contig (+ 1): its first record - 1 = its first "virtual word" (k_features' index space)
    const uint32_t *wfirst; // per 62 virtual words: the contig of the first of them (k_features)
    uint32_t vtotal;      // virtual words of the batch = records without the two pads
    int32_t defcod;       // the codon tables are the reference's defaults (file_handling.py:51-53): k_features uses their formulas
    uint64_t *nbits;    // per contig 9*nw words: node bitmap (forward slot), node bitmap (reverse slot), coverage bitmap; zeroed every run
    uint32_t *nbase;    // per contig 3*nw words: node rank at the start of every 64-position word
    uint64_t *cbits;    // per contig 2*ncw words over node ids: close nodes of the forward strand (forward stops), of the reverse strand (reverse starts); written whole by k_node_order
    uint64_t *bits;     // per contig: [12 codon bitmaps][frame 0..2][nw] (bit k of frame f <-> position f+3k)
    int32_t *iprev;     // per (strand, frame, word): the last item in front of it whose word holds a stop codon, -1: none (k_orf<false> -> k_orf<true>)
    uint2 *item;        // per (strand, frame, word): exclusive ORF / group offsets of the stop events in that word
    const DTNode *tnode; // tRNA nodes / edges of the batch (null: none)
    const DTEdge *tedge;
    int32_t *tnid;      // per tRNA node: its device node id (k_node_build -> k_edges)
    uint32_t *cpre, *bpre; // prefix popcounts of the GC-frame class bitmaps (per PHX_PRE_G words) / of the bases (a, c, t, g per record) (k_bit_prefix, phx_orf.inc)
    uint64_t *tbits;    // per contig 12*nw words at 4/3 * nbits_off: node bitmaps of the tRNA nodes, planes (strand, type) = forward start, forward stop, reverse start, reverse stop; zeroed every run
    DBridge *bridge;    // per contig bridge_cap entries: the uncovered runs of functions.py:334 (k_node_rank -> k_edges)
    // per ORF / group
    DOrf *orf;
    DOrfStat *ostat;    // k_orf_stats -> k_score, k_node_attr
    double *oweight;    // Orf.weight: k_score -> genes (and the tap variant of k_edges)
    long long *owi;     // the same as the solver's integer (ew_encode) and
    uint8_t *oflag;     //   its flags: bit 0 |W| >= 2^51, bit 1 inexact (k_score -> k_edges: bits 30 and 31 of esrc)
    int32_t *onode;     // device node id of the ORF's start node: k_node_build -> k_edges
    DGrp *grp;
    // per node
    DNode *node;        // position, type/frame, link to its ORF / group, other_end: one 16-byte record per node
    int32_t *parent;
    uint32_t *in_off;
    double *no;
    int32_t *npos;      // node positions alone (k_node_attr -> k_edges: its gap-edge loops need nothing else of a candidate, 4 instead of 16 bytes per look)
    DWin *win;          // k_wave_plan -> k_sssp_wave: window records,
    uint2 *wrole;       //   WIN_ROLES lane records per window
    int32_t *olist;     // per contig V-1 node ids: open nodes, the target, close nodes (k_node_order -> k_edges)
    int4 *erank;        // per open CDS node: the id ranges of its two neighbour scans — gap edges [x, y], overlap candidates [z, w] — as the counting pass ranked them
                        //    (k_edges<false> -> k_edges<true>: four rank queries of three gathers each that the filling pass need not repeat)
    uint64_t *ehit;     // per node: verdicts of its first 64 overlap-edge candidates (k_edges<false> -> k_edges<true>)
    uint32_t *mreach;   // per node, 4 x node capacity: the lowest node an overlap (backward) edge out of this close node ends in (0xffffffff: none).
                        //   k_node_attr presets, k_edges<false> fills three partial arrays (see there), k_edges_scan leaves their minimum in the
                        //   first: k_wave_plan watches a close node only where that reaches into the final part, k_sssp_wave steps back to it
    uint64_t *dist;
    int32_t dist_stride; // 64-bit words reserved per node in `dist` (max limbs of the batch)
    // per edge
    uint32_t *esrc;     // source node | off-path << 30 | inexact << 31 (ESRC_NODE / ESRC_OFFPATH / ESRC_INEXACT)
    long long *ew;      // integer weight, encoded (ew_encode / ew_decode); nothing for an edge in the ESRC_F_GAP form until k_edges_expand fills it in
    int32_t orf_rows;   // 1: the rows of the close CDS nodes (the ORF edges, functions.py:311-318) are written by k_edges_orf, a thread per ORF, on a side stream beside
                        //    k_edges<true>, which then stops at the open nodes
    int32_t gap_code;   // 1: k_edges<true> writes gap edges in the coded form (the batch's 128-bit contigs go to k_sssp_duo, which reads the gap table; a batch for
                        //    k_sssp_wave<2> — beyond one contig per SIMD pair — keeps plain rows: completing DBatch.ew for every contig cost more than the fill gained)
    long long *gtab;    // per contig GT_N entries: the encoded integer weight of a gap edge by table index (k_edges_scan: ONE workgroup per contig evaluates the 803 powers —
                        //    until round 6 each of the four workgroups per contig of k_edges<true> did: 15 % of that kernel's VALU instructions); entry GT_N - 1: the
                        //    bits of (1 - pgap)^100, GT_N - 2: some entry carries a flag
    uint16_t *gtabf;    // per contig GT_N entries: the high part of a gap edge's source word, >> 19: flags (bits 11-12 <- 30-31) | coded (bit 10 <- ESRC_F_GAP) | code (bits 0-9)
    uint32_t *esrcf;    // tap variant of k_edges<true> only: plain source nodes and
    double *ewf;        //   fp64 weights, into scratch of their own
    const uint64_t *ewl; // optional integer weights (phx_solve), n_limbs words per edge
    const uint32_t *ekey; // optional rank of every edge in the caller's edge order (phx_solve); else the reference's node insertion order is used
    uint8_t *tie;        // scratch of k_inorder (bump-allocated through DTotals.tie_need)
    int64_t tie_cap;
    int32_t *cint;       // scratch of k_certify: 5 words per node (tree edge, two (link, kappa) pairs),
    uint64_t *csig;      //   2 x dist_stride limbs per node (sigma, double-buffered)
    DERef *eref;         // per edge (sparse: written for the edges that stay flagged): bounds on W* - W from k_refine
    double cert_scale;   // factor on k_certify's error bounds (1; PHX_CREATE_CERT_TIGHT: 2^36, so that the tests see uncertified contigs)
    int32_t defer_overlap; // 1: k_edges<true> queues the overlap edges of a workgroup and evaluates their weights after the neighbour scan (needs node ids < 2^21)
    // output
    int32_t *path;
    DGene *genes;
    uint32_t *gene_total;
    int32_t duo;         // 128-bit contigs of the wavefront solver: k_sssp_duo (two wavefronts per contig) instead of k_sssp_wave<2> (PHX_CREATE_NO_DUO / PHX_NO_DUO: 0)
    int32_t plan_stream; // small batches (a lone contig's planner is longer than the edge fill it hides behind): 0, or the limb count (2, 4, 8) of the
                         // class whose k_sssp_wave is launched without waiting for its k_wave_plan and follows DMeta.plan_prog
    // segments (phx_sssp_seg.inc): small batches — a contig's shortest path by up to SEG_KMAX wavefront pairs side by side, joined and proven by k_seg_merge
    int32_t seg;          // 1: on for this run
    int32_t seg_margin_bp; // sequence a segment sweeps in front of the nodes it commits
    int32_t seg_stream;    // 1: the segments' solvers are launched beside their planner wavefronts and follow DBatch.segw[.][0] as k_sssp_wave follows DMeta.plan_prog
                           //    (batches of up to 4 contigs behind k_front: nothing else would hide the planner there)
    int32_t segw_ints;     // ints of segw that k_reset clears at the head of a run
    int32_t seg_nofb;      // 1: the one sweep for flagged contigs is not launched in this run (an earlier run of the same batch flagged none): a flag now repeats the run
    DWin *swin;           // window / lane records of the segments: SEG_KMAX x the capacity of `win` / `wrole`
    uint2 *swrole;
    uint64_t *sdist;      // distances in the segments' own frames: SEG_KMAX slices of sdist_nodes nodes, 2 words each
    int64_t sdist_nodes;
    int32_t *segw;        // per (contig, segment): windows laid out (-1: none), status of its solver (0: done)
    int32_t front_spins; // k_front: polls of a grid barrier before a workgroup raises front_abort (FRONT_SPINS; env PHX_FRONT_SPINS, -1: the first wait — the tests' way to the staged re-run)
    int32_t gpack;      // batches beyond 4096 contigs: gene records go to a fixed place per contig (grp_off + tn_off; no shared counter) and k_gene_pack moves them together into genes_c
    DGene *genes_c;
};

#ifdef __cplusplus
extern "C" {
#endif
void phx_pack_planes(const char *in, int64_t n, uint32_t *out, int64_t nrec); // phx_host.c: letters -> records (DBatch.recs), `in` starts at a record boundary
// kernel launchers (phx_kernels.hip); all asynchronous on `stream`
void phxk_features(const DBatch *b, uint32_t v_begin, uint32_t v_end, void *stream); // the records [v_begin + 1, v_end + 1)
void phxk_features_tap(const DBatch *b, int contig, uint32_t v_begin, uint32_t v_end, uint32_t *tapbuf, void *stream); // RBS bins of every window of one contig as bit planes: [record of the contig][strand][stream][bit]
void phxk_pack_planes(const DBatch *b, const void *letters, void *stream);            // phx_attach: the caller's letters -> records
void phxk_orf_count(const DBatch *b, void *stream);
void phxk_orf_emit(const DBatch *b, void *stream);
void phxk_bit_prefix(const DBatch *b, void *stream);
void phxk_orf_stats(const DBatch *b, void *stream);
void phxk_score(const DBatch *b, void *stream);
void phxk_nodes(const DBatch *b, void *stream);
void phxk_node_attr(const DBatch *b, void *stream);
void phxk_edges_count(const DBatch *b, void *stream);
void phxk_layout1(const DBatch *b, void *stream); // after orf_count: ORF / group / node offsets, totals
void phxk_layout2(const DBatch *b, void *stream); // after edges_count: edge offsets, integer class and solver per contig, totals
void phxk_edges_fill(const DBatch *b, void *stream);
void phxk_edges_orf(const DBatch *b, void *stream); // DBatch.orf_rows: the ORF edges' rows, a thread per ORF (beside phxk_edges_fill)
void phxk_edges_expand(const DBatch *b, int n_limbs, int mode, void *stream); // DBatch.ew of the gap edges from the gap table: the contigs of limb class n_limbs that kernel `mode` solves (n_limbs 0: every contig)
void phxk_edges_tap(const DBatch *b, void *stream); // k_edges<true> once more, writing fp64 weights and plain sources to DBatch.ewf / esrcf (taps)
void phxk_sssp_order(const DBatch *b, void *stream);
size_t phxk_sssp_lds_bytes(int V, int n_limbs);
void phxk_wave_plan(const DBatch *b, int wide_too, void *stream);
int phxk_sssp_wave_ok(int n_limbs); // limb classes the wavefront-per-contig kernel is built for
void phxk_sssp(const DBatch *b, int n_limbs, int mode, size_t lds_bytes, void *stream);
void phxk_inorder(const DBatch *b, int nl_mask, void *stream);
void phxk_refine(const DBatch *b, void *stream);  // before k_certify: the flagged edges once more in double-double (flags cleared / DBatch.eref)
void phxk_certify(const DBatch *b, int nl_mask, int vmax, void *stream); // after k_inorder: DMeta.cert
void phxk_gene_pack(const DBatch *b, void *stream);
void phxk_reset(const DBatch *b, const void *meta0, unsigned long long nbits_words, unsigned long long tbits_words, void *stream); // the head of a run: bitmaps, totals and per-contig records back to their start values
int phxk_front_blocks_y(const DBatch *b); // workgroups per contig of k_front, 0: the batch is not one for it
void phxk_front(const DBatch *b, void *stream); // small batches: ORF count ... edge fill in one launch (phx_front.inc)
void phxk_seg_merge(const DBatch *b, int vmax, void *stream); // DBatch.seg: after the segment solvers (phxk_sssp mode 2, 128 bits): join + proof + parents
int phxk_seg_kmax(void);
void phxk_seg_fallback(const DBatch *b, void *stream); // ... and one sweep (k_wave_plan<2,0>, k_sssp_duo<0>) for the contigs k_seg_join / k_seg_close flagged (DMeta.seg_fail)
void phxk_results(const DBatch *b, void *stream); // after every solver kernel of the run: parents as the reference's in-place Bellman-Ford leaves them
#ifdef __cplusplus
}
#endif
