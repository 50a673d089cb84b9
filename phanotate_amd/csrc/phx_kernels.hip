// phx_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for PHANOTATE's per-contig hot path; one translation
// unit, the stages live in the .inc files below.
//
// Stage            kernels                                  file               reference code restated
// --------------   --------------------------------------   ----------------   ------------------------------------------
// features         k_features                               phx_features.inc   functions.py:158-171 (per-base loop),
//                                                                              score_rbs 48-138, gc_frame_plot.py:29-74
// ORF scan         k_orf<EMIT>                              phx_orf.inc        functions.py:184-251, orfs.py:17-32
// ORF statistics   k_orf_stats                              phx_orf.inc        orfs.py:162-173, functions.py:261-279,286-298
// ORF weight       k_score                                  phx_orf.inc        functions.py:254-257,281-284,300-301, orfs.py:122-127
// nodes            k_node_cov/_rank/_build/_attr            phx_graph.inc      functions.py:311-318 (nodes), 320-333 (coverage),
//                                                                              363-384 (other_end / o1,o2)
// edges            k_edges<FILL>, k_edges_scan              phx_graph.inc      functions.py:334-354, 360-452
// layout           k_layout1, k_layout2, k_sssp_order,      phx_layout.inc     (offsets, integer class and solver per contig; launch order of the
//                  k_gene_pack, k_results                                      solver and gene slots for large batches; result records)
// shortest path    k_wave_plan + k_sssp_wave<2> (wavefront  phx_sssp_wave.inc  fastpathz (phanotate.py:56-64), exact NL x 64-bit
//                  / contig)
//                  k_sssp_lds<NL> (workgroup / contig),     phx_sssp.inc       integers; path -> genes phanotate.py:65-76,
//                  k_sssp<NL> + k_path<NL> (global memory)                     locus.py:29-37
//                  k_inorder (ties between equal-length paths)  phx_inorder.inc    relaxation order of the reference's solver
//                  k_refine (flagged edges in double-double) phx_refine.inc    functions.py:26-46,286-301: bounds on the reference's integers
//                  k_certify (fp64 vs Decimal weights)      phx_certify.inc    edges.py:17-23: the integers the reference solves on
//
// No MFMA anywhere: the path has no dense contraction.  All integer outputs are
// bit-exact with the reference; fp64 is used where the reference uses Decimal (edge weights only).
#include <hip/hip_runtime.h>
#include <type_traits>

#include "phx_internal.h"

#define NT PHX_CTG_THREADS

// ------------------------------------------------------------------------------------------------
// wave / block primitives (wave64)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v = v > t ? v : t;
    }
    return v;
}
// Exclusive block sum-scan over N threads (N multiple of 64, <= 1024). lds: N/64+1 words.
template <int N>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *lds, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int i = 0; i < N / 64; i++) { uint32_t t = lds[i]; lds[i] = s; s += t; }
        lds[N / 64] = s;
    }
    __syncthreads();
    uint32_t r = inc - v + lds[w];
    *total = lds[N / 64];
    __syncthreads();
    return r;
}
// Exclusive block max-scan (identity 0).
template <int N>
__device__ __forceinline__ uint32_t block_excl_max(uint32_t v, uint32_t *lds, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_max(v);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int i = 0; i < N / 64; i++) { uint32_t t = lds[i]; lds[i] = s; s = s > t ? s : t; }
        lds[N / 64] = s;
    }
    __syncthreads();
    uint32_t prev = __shfl_up(inc, 1);
    if (lane == 0) prev = 0;
    uint32_t r = prev > lds[w] ? prev : lds[w];
    *total = lds[N / 64];
    __syncthreads();
    return r;
}

// value of the lane `n` places to the left inside the 16-lane DPP row (own value for the first n lanes)
template <int N>
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x110 + N, 0xf, 0xf, false); }
template <int N>
__device__ __forceinline__ uint64_t dpp_row_shr64(uint64_t x) { return ((uint64_t)dpp_row_shr<N>((uint32_t)(x >> 32)) << 32) | dpp_row_shr<N>((uint32_t)x); }
// sum over a 16-lane DPP row; the total lands in lane 15 of the row (sub = lane & 15)
__device__ __forceinline__ uint64_t row16_sum_u64(uint64_t x, int sub) {
    uint64_t y;
    y = dpp_row_shr64<1>(x); if (sub >= 1) x += y;
    y = dpp_row_shr64<2>(x); if (sub >= 2) x += y;
    y = dpp_row_shr64<4>(x); if (sub >= 4) x += y;
    y = dpp_row_shr64<8>(x); if (sub >= 8) x += y;
    return x;
}

// ------------------------------------------------------------------------------------------------
// Edge records as the kernels keep them (DBatch.esrc / DBatch.ew, 4 + 8 bytes per edge).
//   ew: the integer the solver works on, W = trunc(w * 1000) (edges.py:22), so that no consumer converts fp64 again: a value with
//       |W| < 2^62 as it is (bits 63 and 62 equal); a wider one as the bit pattern of the double p = trunc(w * 1000) — whose bit 62 is
//       always set, |p| >= 2^62 — with bit 62 forced to differ from bit 63, which marks it (ew_decode, phx_sssp.inc, restores it).
//   esrc: source node in bits 0..29; bit 30 = "off the solver's common path": |W| >= 2^51 (its integer is not the low half of a narrow
//       128-bit value) or the edge leaves the source node (whose distance has its own ring slot), so that k_sssp_wave tests one
//       bit per in-edge; bit 31 = "inexact": the reference's integer trunc(Decimal(w) * 1000) may differ from W
//       (k_certify's eps_e > 0, phx_certify.inc) — decided here, where the fp64 weight still exists.
// The fp64 weights themselves are not kept: the taps recompute them (k_edges<true, true>).
#define EW_WIDE(x) (((((unsigned long long)(x)) >> 63) ^ (((unsigned long long)(x)) >> 62)) & 1ull)
__device__ __forceinline__ long long ew_encode(double w) {
    const double t = trunc(w * 1000.0);
    if (fabs(t) < 2251799813685248.0) return __double_as_longlong(t + 6755399441055744.0) - 0x4338000000000000ll; // |t| < 2^51: the integer sits in the low mantissa bits of t + 1.5 * 2^52
    if (fabs(t) < 4611686018427387904.0) return (long long)t;
    const long long bits = __double_as_longlong(t);
    return bits < 0 ? (bits & ~(1ll << 62)) : bits;
}
__device__ __forceinline__ bool ew_narrow51(long long w) { return (unsigned long long)(w + (1ll << 51)) < (1ull << 52); }
// cert_eps(w) == 0 (phx_certify.inc): p = w * 1000 is farther from the next integer than an upper bound of its error bound
// |p| (|exponent| + 8) 2^-46 x scale — |exponent| + 8 <= 60 for 1/2 <= |p| < 2^52, and a |p| below 1/2 truncates to 0 on both sides
// whatever its exponent —, so the reference's truncation cannot differ.  c = scale * 2^-46.
__device__ __forceinline__ bool cert_eps_is_zero_fast(double w, double c) {
#ifdef EW_NOCHECK
    return true;
#endif
    const double p = w * 1000.0, a = fabs(p);
    const double f = a - fabs(trunc(p));
    const bool far = (f < 1.0 - f ? f : 1.0 - f) > a * (60.0 * c);
    return w == -20.0 || (a < 4503599627370496.0 && far);
}
#define CERT_C(b) ((b).cert_scale * 1.4210854715202004e-14) // 2^-46
// one edge weight in registers: what goes to DBatch.ew (the encoded integer; the fp64 bits in the tap variant) and the inexact flag
struct EWt { unsigned long long bits; uint32_t fl; int code; }; // code >= 0: an entry of the contig's gap table (the edge goes out in the ESRC_F_GAP form)
template <bool TAPW>
__device__ __forceinline__ EWt make_ew(double w, double c) {
    EWt r;
    r.code = -1;
    if (TAPW) { r.bits = (unsigned long long)__double_as_longlong(w); r.fl = 0u; }
    else {
        r.bits = (unsigned long long)ew_encode(w);
        r.fl = (cert_eps_is_zero_fast(w, c) ? 0u : ESRC_F_INEXACT) | (ew_narrow51((long long)r.bits) ? 0u : ESRC_F_OFFPATH);
    }
    return r;
}

// the encoded integer weight of edge e (contig-relative) whose source word is sw: a coded gap edge (ESRC_F_GAP) has it in the contig's gap table
__device__ __forceinline__ long long edge_wenc(uint32_t sw, const long long *ew, uint32_t e, const long long *gt) { return ESRC_IS_GAP(sw) ? gt[ESRC_GAP_CODE(sw)] : ew[e]; }
__device__ __forceinline__ const long long *gtab_of(const DBatch &b, const DMeta *meta) { return b.gtab ? b.gtab + (size_t)(meta - b.meta) * GT_N : nullptr; } // (phx_solve: no table, no coded edge)

// functions.py:174-178: both strands are counted, so Pa == Pt and Pg == Pc.
__device__ __forceinline__ double contig_pstop(uint32_t gc, int L) {
    double fa = (double)((uint32_t)L - gc), fg = (double)gc;
    double d = (double)((int64_t)L * 2);
    double Pa = fa / d, Pt = fa / d, Pg = fg / d;
    return Pt * Pa * Pa + Pt * Pg * Pa + Pt * Pa * Pg;
}
// gc_frame_plot.py:7-28
__device__ __forceinline__ int max_idx(int a, int b, int c) { return a > b ? (a > c ? 1 : 3) : (b > c ? 2 : 3); }
__device__ __forceinline__ int min_idx(int a, int b, int c) { return a > b ? (b > c ? 3 : 2) : (a > c ? 3 : 1); }

#include "phx_features.inc"
#include "phx_orf.inc"
#include "phx_graph.inc"
#include "phx_sssp.inc"
#include "phx_layout.inc"
#include "phx_front.inc"
#include "phx_sssp_seg.inc"
#include "phx_sssp_wave.inc"
#include "phx_sssp_duo.inc"
#include "phx_inorder.inc"
#include "phx_refine.inc"
#include "phx_certify.inc"

// ------------------------------------------------------------------------------------------------
// launchers
template <int NL>
static void launch_certify(const DBatch *b, int vcap, hipStream_t s) {
    const size_t lds = (size_t)vcap * 12 + 16;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_certify<NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_certify<NL>, dim3(b->n_contig), dim3(CERT_T), lds, s, *b, vcap);
    hipLaunchKernelGGL(k_certify_wide<NL>, dim3(b->n_contig), dim3(CERT_T), 0, s, *b);
}
extern "C" {
void phxk_features(const DBatch *b, uint32_t v_begin, uint32_t v_end, void *stream) {
    if (v_end <= v_begin) return;
    const unsigned g = (v_end - 1) / FEAT_SPAN - v_begin / FEAT_SPAN + 1;
    if (b->defcod) hipLaunchKernelGGL((k_features<false, true>), dim3(g), dim3(64), 0, (hipStream_t)stream, *b, v_begin, v_end, (uint32_t *)nullptr, -1);
    else hipLaunchKernelGGL((k_features<false, false>), dim3(g), dim3(64), 0, (hipStream_t)stream, *b, v_begin, v_end, (uint32_t *)nullptr, -1);
}
// (host copies of voff are the caller's: it passes the contig's range)
void phxk_features_tap(const DBatch *b, int contig, uint32_t v_begin, uint32_t v_end, uint32_t *tapbuf, void *stream) {
    if (v_end <= v_begin) return;
    const unsigned g = (v_end - 1) / FEAT_SPAN - v_begin / FEAT_SPAN + 1;
    hipLaunchKernelGGL((k_features<true, false>), dim3(g), dim3(64), 0, (hipStream_t)stream, *b, v_begin, v_end, tapbuf, contig);
}
void phxk_pack_planes(const DBatch *b, const void *letters, void *stream) {
    if (b->n_contig > 0) hipLaunchKernelGGL(k_pack_planes, dim3(b->n_contig, 8), dim3(64), 0, (hipStream_t)stream, *b, (const uint8_t *)letters);
}
// Workgroups per contig of the per-contig kernels: `full` for the benchmark's 50 kb contigs, fewer for batches of short contigs
// (a 2 kb contig has ~100 nodes: four workgroups of 256 threads would leave three idle), by the batch's mean contig length.
#ifndef YS_EMIT
#define YS_EMIT 6
#endif
#ifndef YS_STATS
#define YS_STATS 8
#endif
static unsigned ysplit(const DBatch *b, unsigned full) {
    const unsigned y = (unsigned)((b->mean_len + 8191) / 8192);
    // a few long contigs (T4 alone: 21 pieces of 8 kb) leave the chip empty: four times as many workgroups per contig while they all fit at once
    const unsigned cap = (unsigned)b->n_contig * 4u * full <= 256u ? 4u * full : full;
    return y < 1u ? 1u : (y > cap ? cap : y);
}
void phxk_orf_count(const DBatch *b, void *stream) {
    if (b->mean_len < 16384) hipLaunchKernelGGL((k_orf<false, 256>), dim3(b->n_contig), dim3(256), 0, (hipStream_t)stream, *b);
    else hipLaunchKernelGGL((k_orf<false, ORF_COUNT_T>), dim3(b->n_contig), dim3(ORF_COUNT_T), 0, (hipStream_t)stream, *b);
}
void phxk_orf_emit(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_orf<true>, dim3(b->n_contig, ysplit(b, YS_EMIT)), dim3(NT), 0, (hipStream_t)stream, *b); }
void phxk_bit_prefix(const DBatch *b, void *stream) {
    hipLaunchKernelGGL(k_bit_prefix, dim3(b->n_contig, 7), dim3(64), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_gap_table, dim3(b->n_contig), dim3(NT), 0, (hipStream_t)stream, *b); // (needs the g+c counts of k_features only: here it hides beside the ORF scan)
}
void phxk_orf_stats(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_orf_stats, dim3(b->n_contig, ysplit(b, YS_STATS)), dim3(NT), 0, (hipStream_t)stream, *b); }
void phxk_score(const DBatch *b, void *stream) {
    if (b->n_contig <= 16 && b->mean_len >= 32768) { hipLaunchKernelGGL(k_score_big, dim3(b->n_contig), dim3(1024), 0, (hipStream_t)stream, *b); return; }
    hipLaunchKernelGGL(k_score, dim3(b->n_contig), dim3(b->mean_len >= 8192 ? NT : 64), 0, (hipStream_t)stream, *b);
}
// node stage, part 1: needs the ORF / group records of k_orf<true> only (not their statistics), so the launcher runs it
// beside k_orf_stats / k_score
void phxk_nodes(const DBatch *b, void *stream) {
#ifndef NODES_STAGED
    // one workgroup per contig runs the four bodies in a row — unless the batch is a few LONG contigs (T4: one 256-thread workgroup then takes
    // 111 us and k_node_attr waits for it; the staged kernels spread coverage and records over four workgroups per contig: 60 us)
    if (!(b->n_contig <= 16 && b->mean_len >= 32768)) {
        hipLaunchKernelGGL(k_nodes_fused, dim3(b->n_contig), dim3(NT), 0, (hipStream_t)stream, *b);
        return;
    }
#endif
    hipLaunchKernelGGL(k_node_cov, dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_node_rank, dim3(b->n_contig), dim3(NT), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_node_build, dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_node_order, dim3(b->n_contig), dim3(NT), 0, (hipStream_t)stream, *b);
}
// part 2: other_end and the p_stop attribute of every node (reads DOrf.pstop of k_orf_stats)
void phxk_node_attr(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_node_attr, dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b); }
void phxk_edges_count(const DBatch *b, void *stream) {
    hipLaunchKernelGGL(k_edges<false>, dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b);
    if (b->n_contig <= 16 && b->mean_len >= 32768) hipLaunchKernelGGL(k_edges_scan_big, dim3(b->n_contig), dim3(1024), 0, (hipStream_t)stream, *b);
    else hipLaunchKernelGGL(k_edges_scan, dim3(b->n_contig), dim3(ES_T), 0, (hipStream_t)stream, *b);
}
void phxk_edges_fill(const DBatch *b, void *stream) {
    if (b->gap_code) hipLaunchKernelGGL((k_edges<true, false, true>), dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b); // gap edges in the coded form
    else hipLaunchKernelGGL(k_edges<true>, dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b);
}
void phxk_edges_orf(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_edges_orf, dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b); }
void phxk_edges_expand(const DBatch *b, int nl, int mode, void *stream) { hipLaunchKernelGGL(k_edges_expand, dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b, nl, mode); }
void phxk_edges_tap(const DBatch *b, void *stream) { hipLaunchKernelGGL((k_edges<true, true>), dim3(b->n_contig, ysplit(b, 4)), dim3(NT), 0, (hipStream_t)stream, *b); }
// phx_solve: relaxation, path walk (no genes: DBatch.genes is null), in-order parents
void phxk_sssp_only(const DBatch *b, int nl, void *stream) {
    phxk_sssp(b, nl, 0, 0, stream);
    phxk_inorder(b, nl == 2 ? 1 : nl == 4 ? 2 : nl == 8 ? 4 : 8, stream);
}

// nl_mask: bit k set = some contig of the batch has 2 / 4 / 8 / 17 limbs (k = 0..3)
void phxk_inorder(const DBatch *b, int nl_mask, void *stream) {
    dim3 g(b->n_contig);
    hipStream_t s = (hipStream_t)stream;
    // the benchmark's kind of contig: 256 threads; batches of short contigs (a workgroup is one contig with a hundred nodes): one wavefront
    // and a quarter of the LDS for the path walk, so that four times as many contigs are in flight
    const bool small = b->mean_len < 8192;
    const bool big = (b->n_contig <= 64 && b->mean_len >= 65536) || (b->n_contig <= 8 && b->mean_len >= 16384); // a handful of long contigs (or a few of any length: Lambda): the chip is empty, a contig's one workgroup may as well be a large one — and its path walk goes in strides
    if (big && (nl_mask & 1)) { hipLaunchKernelGGL((k_inorder<2, IO_T_BIG>), g, dim3(IO_T_BIG), 0, s, *b); nl_mask &= ~1; }
    if (small) {
        if (nl_mask & 1) hipLaunchKernelGGL((k_inorder<2, 64>), g, dim3(64), 0, s, *b);
        if (nl_mask & 2) hipLaunchKernelGGL((k_inorder<4, 64>), g, dim3(64), 0, s, *b);
        if (nl_mask & 4) hipLaunchKernelGGL((k_inorder<8, 64>), g, dim3(64), 0, s, *b);
        if (nl_mask & 8) hipLaunchKernelGGL((k_inorder<17, 64>), g, dim3(64), 0, s, *b);
    } else {
        if (nl_mask & 1) hipLaunchKernelGGL((k_inorder<2, IO_T_FULL>), g, dim3(IO_T_FULL), 0, s, *b);
        if (nl_mask & 2) hipLaunchKernelGGL((k_inorder<4, IO_T_FULL>), g, dim3(IO_T_FULL), 0, s, *b);
        if (nl_mask & 4) hipLaunchKernelGGL((k_inorder<8, IO_T_FULL>), g, dim3(IO_T_FULL), 0, s, *b);
        if (nl_mask & 8) hipLaunchKernelGGL((k_inorder<17, IO_T_FULL>), g, dim3(IO_T_FULL), 0, s, *b);
    }
}

// vmax: the largest node count the batch is expected to hold (the last run's): sizes the LDS tables of k_certify (12 bytes per node, at
// most 144 KB); a contig with more nodes than that goes through k_certify_wide, launched right behind (it returns at once otherwise)
void phxk_certify(const DBatch *b, int nl_mask, int vmax, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    int vcap = ((vmax > 1024 ? vmax : 1024) + 255) & ~255;
    if (vcap > 12288) vcap = 12288;
    if (vmax < 0) vcap = 0; // (test switch: everything to k_certify_wide)
    if (nl_mask & 1) launch_certify<2>(b, vcap, s);
    if (nl_mask & 2) launch_certify<4>(b, vcap, s);
    if (nl_mask & 4) launch_certify<8>(b, vcap, s);
    if (nl_mask & 8) launch_certify<17>(b, vcap, s);
}

void phxk_refine(const DBatch *b, void *stream) {
    // (one workgroup per contig; a few long contigs — T4 alone: 29 000 edges, 15 rounds of its scan — get up to 16: 97 -> see DESIGN.md §7)
    const unsigned y = b->n_contig <= 16 ? ysplit(b, 4) : 1u;
    hipLaunchKernelGGL(k_refine, dim3(b->n_contig, y), dim3(RF_T), 0, (hipStream_t)stream, *b);
}

void phxk_gene_pack(const DBatch *b, void *stream) {
    const unsigned g = (unsigned)((b->n_contig + LMB_T - 1) / LMB_T);
    hipLaunchKernelGGL(k_gene_pack_a, dim3(g), dim3(LMB_T), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_gene_pack_b, dim3(g, 8), dim3(LMB_T), 0, (hipStream_t)stream, *b);
}
void phxk_reset(const DBatch *b, const void *meta0, unsigned long long nbits_words, unsigned long long tbits_words, void *stream) {
    unsigned long long work = nbits_words / 2 + tbits_words / 2 + (unsigned long long)b->n_contig * (sizeof(DMeta) / 8);
    unsigned g = (unsigned)((work + 255) / 256);
    g = g < 1u ? 1u : (g > 4096u ? 4096u : g);
    hipLaunchKernelGGL(k_reset, dim3(g), dim3(256), 0, (hipStream_t)stream, *b, (const DMeta *)meta0, nbits_words, tbits_words);
}
int phxk_front_blocks_y(const DBatch *b) {
    if (b->n_contig < 1 || b->n_contig > FRONT_MAX_CONTIGS || b->mean_len * b->n_contig > (40 << 10)) return 0; // (since the staged kernels of a few long contigs run with four times as many workgroups per contig — ysplit — they win from ~32 kb on: 8 / 16 / 24 kb fused 0.273 / 0.304 / 0.307 ms against 0.291 / 0.314 / 0.314 staged, 32 kb 0.313 / 0.312, Lambda 0.369 / 0.360, 2 x 50 kb 0.395 / 0.376)
    unsigned y = ysplit(b, 8);
    while (y > 1 && (unsigned)b->n_contig * y > FRONT_MAX_BLOCKS) y--;
    return (unsigned)b->n_contig * y <= FRONT_MAX_BLOCKS ? (int)y : 0;
}
void phxk_front(const DBatch *b, void *stream) {
    const int y = phxk_front_blocks_y(b);
    if (y > 0) hipLaunchKernelGGL(k_front, dim3(b->n_contig, y), dim3(NT), 0, (hipStream_t)stream, *b);
}
void phxk_seg_merge(const DBatch *b, int vmax, void *stream) {
    // workgroups per contig of the join: one per 16 nodes of the largest contig the last run saw (a workgroup walks its share in rounds of four
    // dependent loads: T4 on 64 workgroups took 5.4 rounds, 22 us), at most 4096 in all
    const int n = b->n_contig > 0 ? b->n_contig : 1;
    int y = vmax > 0 ? (vmax + SEGJ_T / SEGJ_LPN - 1) / (SEGJ_T / SEGJ_LPN) : 64;
    const int cap = 4096 / n;
    y = y > cap ? cap : y;
    y = y < 4 ? 4 : y;
    hipLaunchKernelGGL(k_seg_join, dim3(b->n_contig, y), dim3(SEGJ_T), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_seg_close, dim3(b->n_contig), dim3(SEGM_T), 0, (hipStream_t)stream, *b);
}
int phxk_seg_kmax(void) { return SEG_KMAX; }
void phxk_seg_fallback(const DBatch *b, void *stream) {
    hipLaunchKernelGGL((k_wave_plan<2, 0>), dim3(b->n_contig), dim3(64), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL((k_sssp_duo<0, false>), dim3(b->n_contig), dim3(128), 0, (hipStream_t)stream, *b);
}
void phxk_results(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_results, dim3((unsigned)((b->n_contig + LMB_T - 1) / LMB_T)), dim3(LMB_T), 0, (hipStream_t)stream, *b); }
size_t phxk_sssp_lds_bytes(int V, int nl) { return sssp_lds_bytes(V, nl); }
// one workgroup for up to 1024 contigs; larger batches in two passes of a workgroup per 256 contigs
void phxk_layout1(const DBatch *b, void *stream) {
    if (b->n_contig <= LAYOUT_T) { hipLaunchKernelGGL(k_layout1, dim3(1), dim3(LAYOUT_T), 0, (hipStream_t)stream, *b); return; }
    const unsigned g = (unsigned)((b->n_contig + LMB_T - 1) / LMB_T);
    hipLaunchKernelGGL(k_layout1_a, dim3(g), dim3(LMB_T), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_layout1_b, dim3(g), dim3(LMB_T), 0, (hipStream_t)stream, *b);
}
void phxk_layout2(const DBatch *b, void *stream) {
    if (b->n_contig <= LAYOUT_T) { hipLaunchKernelGGL(k_layout2, dim3(1), dim3(LAYOUT_T), 0, (hipStream_t)stream, *b); return; }
    const unsigned g = (unsigned)((b->n_contig + LMB_T - 1) / LMB_T);
    hipLaunchKernelGGL(k_layout2_a, dim3(g), dim3(LMB_T), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_layout2_b, dim3(g), dim3(LMB_T), 0, (hipStream_t)stream, *b);
}

void phxk_sssp_order(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_sssp_order, dim3(1), dim3(LMB_T), 0, (hipStream_t)stream, *b); }

int phxk_sssp_wave_ok(int nl) { return nl == 2 || nl == 4 || nl == 8; }
// windows and lane assignments of k_sssp_wave (needs the node records and in-edge offsets, not the edges); wide_too: the batch
// (may) hold 256-bit contigs for the wavefront kernel, whose lanes keep fewer in-edges
void phxk_wave_plan(const DBatch *b, int wide_too, void *stream) {
    if (b->seg) hipLaunchKernelGGL((k_wave_plan<2, 0, true>), dim3(b->n_contig * b->seg), dim3(64), 0, (hipStream_t)stream, *b); // one wavefront per segment
    else hipLaunchKernelGGL((k_wave_plan<2, 0>), dim3(b->n_contig), dim3(64), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL((k_wave_plan<2, 1>), dim3(b->n_contig), dim3(64), 0, (hipStream_t)stream, *b); // the contigs the tight configuration could not take
    if (wide_too & 1) hipLaunchKernelGGL((k_wave_plan<4, 1>), dim3(b->n_contig), dim3(64), 0, (hipStream_t)stream, *b);
    if (wide_too & 2) hipLaunchKernelGGL((k_wave_plan<8, 1>), dim3(b->n_contig), dim3(64), 0, (hipStream_t)stream, *b);
}

// mode 0: global-memory kernel (+ k_path); mode 1: workgroup-per-contig LDS kernel with `lds_bytes` of dynamic LDS;
// mode 2: wavefront-per-contig kernel (contigs it hands back carry their fallback mode in sssp_mode afterwards); mode 3: the same
// kernel in its roomy configuration (128-bit contigs that k_wave_plan could not lay out in the tight one)
void phxk_sssp(const DBatch *b, int nl, int mode, size_t lds_bytes, void *stream) {
    dim3 g(b->n_contig), t(NT);
    hipStream_t s = (hipStream_t)stream;
    // k_sssp_duo reads the weights of the coded gap edges from the contig's gap table; every other solver kernel reads plain (source, weight) rows:
    // k_sssp_lds and k_sssp complete DBatch.ew of the contigs they take themselves (expand_contig), k_sssp_wave gets it completed by a launch in front of it
    if ((mode == 2 || mode == 3) && !(mode == 2 && nl == 2 && b->duo) && b->gtab && b->gap_code) phxk_edges_expand(b, nl, mode, stream);
    if (mode == 2 || mode == 3) {
        if (nl == 2 && mode == 2 && b->duo) { // two wavefronts per contig: the feeder prepares the windows, the solver runs the phases (phx_sssp_duo.inc).
            // (The roomy configuration — the few contigs whose windows need more spill entries than the tight one holds — stays with k_sssp_wave<2, 1>.)
            dim3 t2(128);
            if (b->seg && b->seg_stream) hipLaunchKernelGGL((k_sssp_duo<0, true, true>), dim3(b->n_contig * b->seg), t2, 0, s, *b); // ... beside their planner wavefronts
            else if (b->seg) hipLaunchKernelGGL((k_sssp_duo<0, false, true>), dim3(b->n_contig * b->seg), t2, 0, s, *b); // a wavefront pair per segment (phx_sssp_seg.inc)
            else if (b->plan_stream == 2) hipLaunchKernelGGL((k_sssp_duo<0, true>), g, t2, 0, s, *b);
            else hipLaunchKernelGGL((k_sssp_duo<0, false>), g, t2, 0, s, *b);
        } else if (nl == 2 && mode == 2) {
            const size_t lb = wv_lds_bytes<2, 0>();
            (void)hipFuncSetAttribute((const void *)k_sssp_wave<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            if (b->plan_stream == 2) { // small batches: beside k_wave_plan<2,0>, following its counter
                (void)hipFuncSetAttribute((const void *)k_sssp_wave<2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL((k_sssp_wave<2, 0, true>), g, dim3(64), lb, s, *b);
            } else
            hipLaunchKernelGGL((k_sssp_wave<2, 0>), g, dim3(64), lb, s, *b);
        } else if (nl == 2) {
            const size_t lb = wv_lds_bytes<2, 1>();
            (void)hipFuncSetAttribute((const void *)k_sssp_wave<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            hipLaunchKernelGGL((k_sssp_wave<2, 1>), g, dim3(64), lb, s, *b);
        } else if (nl == 4) {
            const size_t lb = wv_lds_bytes<4, 1>();
            (void)hipFuncSetAttribute((const void *)k_sssp_wave<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            if (b->plan_stream == 4) {
                (void)hipFuncSetAttribute((const void *)k_sssp_wave<4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL((k_sssp_wave<4, 1, true>), g, dim3(64), lb, s, *b);
            } else
            hipLaunchKernelGGL((k_sssp_wave<4, 1>), g, dim3(64), lb, s, *b);
        } else {
            const size_t lb = wv_lds_bytes<8, 1>();
            (void)hipFuncSetAttribute((const void *)k_sssp_wave<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
            if (b->plan_stream == 8) {
                (void)hipFuncSetAttribute((const void *)k_sssp_wave<8, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
                hipLaunchKernelGGL((k_sssp_wave<8, 1, true>), g, dim3(64), lb, s, *b);
            } else
            hipLaunchKernelGGL((k_sssp_wave<8, 1>), g, dim3(64), lb, s, *b);
        }
        return;
    }
    if (mode == 0) {
        dim3 gp((b->n_contig + 63) / 64), tp(64);
        switch (nl) {
        case 2: hipLaunchKernelGGL(k_sssp<2>, g, t, 0, s, *b); hipLaunchKernelGGL(k_path<2>, gp, tp, 0, s, *b); break;
        case 4: hipLaunchKernelGGL(k_sssp<4>, g, t, 0, s, *b); hipLaunchKernelGGL(k_path<4>, gp, tp, 0, s, *b); break;
        case 8: hipLaunchKernelGGL(k_sssp<8>, g, t, 0, s, *b); hipLaunchKernelGGL(k_path<8>, gp, tp, 0, s, *b); break;
        default: hipLaunchKernelGGL(k_sssp<17>, g, t, 0, s, *b); hipLaunchKernelGGL(k_path<17>, gp, tp, 0, s, *b); break;
        }
        return;
    }
    switch (nl) {
    case 2:
        (void)hipFuncSetAttribute((const void *)k_sssp_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_sssp_lds<2>, g, dim3(SW_THREADS), lds_bytes, s, *b, mode, (int)lds_bytes); break;
    case 4:
        (void)hipFuncSetAttribute((const void *)k_sssp_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_sssp_lds<4>, g, dim3(SW_THREADS), lds_bytes, s, *b, mode, (int)lds_bytes); break;
    case 8:
        (void)hipFuncSetAttribute((const void *)k_sssp_lds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_sssp_lds<8>, g, dim3(SW_THREADS), lds_bytes, s, *b, mode, (int)lds_bytes); break;
    default:
        (void)hipFuncSetAttribute((const void *)k_sssp_lds<17>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(k_sssp_lds<17>, g, dim3(SW_THREADS), lds_bytes, s, *b, mode, (int)lds_bytes); break;
    }
}
}

