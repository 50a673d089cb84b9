// phx_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels for PHANOTATE's per-contig hot path.
//
// Stage           kernel            granularity                       reference code restated
// -------------   ---------------   -------------------------------   ---------------------------------------
// features        k_features        1 workgroup / 2048-position tile   functions.py:158-171 (per-base loop),
//                                                                      score_rbs 48-138, gc_frame_plot.py:29-74
// ORF scan        k_orf<EMIT>       1 workgroup / contig               functions.py:184-251, orfs.py:17-32
// ORF stats       k_orf_stats       thread / ORF, thread / group       orfs.py:162-173, functions.py:261-279,286-298
// ORF score       k_score           thread / ORF                       functions.py:254-257,281-284,300-301, orfs.py:122-127
// nodes           k_nodes           1 workgroup / contig               functions.py:311-318 (nodes), 320-333 (coverage),
//                                                                      363-384 (other_end / o1,o2)
// edges           k_edges<FILL>     thread / destination node          functions.py:334-354, 360-452
// shortest path   k_sssp<NL>        1 workgroup / contig               fastpathz (phanotate.py:56-64), exact NL x 64-bit ints
// genes           k_path<NL>        thread / contig                    phanotate.py:65-76, locus.py:29-37
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

// ------------------------------------------------------------------------------------------------
// k_features: ASCII -> per-position feature bytes.  Base code: a=0 c=1 t=2 g=3 (complement = x^2,
// GC = x&1); bit2 = "cannot match a motif/codon" (ambiguity code or outside the contig), bit3 = outside.
#define FW (PHX_TILE + 2 * PHX_HALO)

__device__ __forceinline__ uint32_t base_code(uint32_t ch, bool &bad) {
    if (ch >= 'A' && ch <= 'Z') ch |= 0x20u; // .lower(), functions.py:144
    switch (ch) {
    case 'a': return 0; case 'c': return 1; case 't': return 2; case 'g': return 3;
    case 's': case 'b': case 'v': return 3u | 4u; // counted as g, functions.py:160-161
    case 'n': case 'r': case 'y': case 'w': case 'k': case 'm': case 'd': case 'h': return 0u | 4u; // counted as a
    default: bad = true; return 0u | 4u; // KeyError in rev_comp, functions.py:20-24
    }
}
__device__ __forceinline__ int off_class(int o) { return o <= 4 ? 0 : (o <= 10 ? 1 : (o <= 12 ? 2 : 3)); }

// Packed per-class scores of the k-mers that start at s[0] (k = 3..min(6,v)), given the 6 symbols.
// Every motif of score_rbs starts with ag, ga or gg, so the table is stored compactly: for every usable prefix
// length v = 6,5,4,3 a block [pair][remaining symbols] (768 + 192 + 48 + 12 words) and one trailing zero word for
// "no motif can match".  Codes: a0 c1 t2 g3, symbol j at bits 2j.  Branch-free: one LDS read.
#define RBS_TAB_WORDS 1021
__device__ __forceinline__ uint32_t kmer_lookup(const uint32_t *tab, uint32_t code, int v) {
    // pair index + 1 for (s0 | s1<<2): ag = 0|3<<2 = 12 -> 1, ga = 3|0 = 3 -> 2, gg = 15 -> 3, else 0
    const uint32_t pi = (((1u << 24) | (2u << 6) | (3u << 30)) >> (2 * (code & 15u))) & 3u;
    const int vv = v > 6 ? 6 : (v < 2 ? 2 : v);
    const uint32_t sh = 2u * (uint32_t)(vv - 2);                 // remaining symbols -> 8,6,4,2 bits
    const uint32_t size = 1u << sh;                              // 256,64,16,4
    const uint32_t base = 3u * (340u - ((4u << sh) - 4u) / 3u); // 0,768,960,1008 for v = 6,5,4,3  (340 = 256+64+16+4)
    const uint32_t idx = base + (pi - 1u) * size + ((code >> 4) & (size - 1u));
    return tab[(pi != 0u && vv >= 3) ? idx : (uint32_t)(RBS_TAB_WORDS - 1)];
}

#define FPT (PHX_TILE / PHX_FEAT_THREADS) // tile positions per thread in the output stage (consecutive)
#define FIT 9                              // window indices per thread in the scan stages (consecutive, multiple of 3)
#define FWX 1665                           // staged window: 64 halo + 1536 + 65 halo = 9 * 185 indices
#define FACT (FWX / FIT)                   // threads that own window indices
static_assert(FWX % FIT == 0 && FIT % 3 == 0 && FWX >= PHX_TILE + 2 * PHX_HALO && FACT <= PHX_FEAT_THREADS, "feature window");
#define SCPAD 8 // s_code / s_W carry 8 "outside" entries on either side so that the rolling scans need no bounds checks

// per-position outputs of one thread (FPT consecutive positions).  FULL: the whole tile (and the 2 bases after
// it) lies inside the contig, so no position needs a bounds test.
template <bool FULL>
__device__ __forceinline__ void feature_outputs(int tid, int p0, int L, const DParams *P, const uint8_t *sc, const uint8_t *sW, const uint32_t *s_AF,
                                                const uint32_t *s_AR, uint32_t *s_hist, uint8_t *o_cls, uint8_t *o_gcc, uint16_t *o_rbs,
                                                uint32_t &nz0) {
    const int j0 = tid * FPT;
    const int x0 = PHX_HALO + j0;
    uint32_t af[FPT + 12], ar[FPT + 12]; // af[m] = AF[x0 - 15 + m], ar[m] = AR[x0 + 3 + m]
#pragma unroll
    for (int m = 0; m < FPT + 12; m++) { af[m] = s_AF[x0 - 15 + m]; ar[m] = s_AR[x0 + 3 + m]; }
    uint32_t cd[FPT + 2], wv[FPT + 2];
#pragma unroll
    for (int m = 0; m < FPT + 2; m++) { cd[m] = sc[SCPAD + x0 + m]; wv[m] = sW[SCPAD + x0 + m]; }
#pragma unroll
    for (int j = 0; j < FPT; j++) {
        const int p = p0 + j0 + j;
        const bool in = FULL || p < L, codon = FULL || p <= L - 3;
        const uint32_t c0 = cd[j], c1 = cd[j + 1], c2 = cd[j + 2];
        // codon class: entry 64 of the tables is "no codon" (ambiguous base, or past the end)
        const uint32_t ci = (codon && !((c0 | c1 | c2) & 4u)) ? ((c0 & 3u) | ((c1 & 3u) << 2) | ((c2 & 3u) << 4)) : 64u;
        const uint32_t cls = P->cls_tab[ci], atg = P->atg_tab[ci];
        const int w0 = (int)wv[j], w1 = (int)wv[j + 1], w2 = (int)wv[j + 2];
        // 2-bit fields: max_idx-1, min_idx-1 of the forward triple, then of the reversed triple
        const uint32_t gcc = (uint32_t)(max_idx(w0, w1, w2) - 1) | ((uint32_t)(min_idx(w0, w1, w2) - 1) << 2) |
                             ((uint32_t)(max_idx(w2, w1, w0) - 1) << 4) | ((uint32_t)(min_idx(w2, w1, w0) - 1) << 6);
        // score_rbs bins: forward window dna[p-20:p+1] (needs p >= 20), reverse window rev_comp(dna[p:p+21]);
        // offsets 3-4 use class byte 0, 5-10 byte 1, 11-12 byte 2, 13-15 byte 3
        uint32_t bf = 0, br = 0;
#pragma unroll
        for (int o = 3; o <= 15; o++) {
            const int sh = 8 * (o <= 4 ? 0 : (o <= 10 ? 1 : (o <= 12 ? 2 : 3)));
            const uint32_t sf = (af[15 + j - o] >> sh) & 0xffu, sr = (ar[j + o - 3] >> sh) & 0xffu;
            bf = sf > bf ? sf : bf;
            br = sr > br ? sr : br;
        }
        const bool fwin = in && p >= 20; // background: full-length forward window i = p-20 (functions.py:168)
        bf = fwin ? bf : 0u;
        br = in ? br : 0u;
        nz0 += (fwin && bf == 0 ? 1u : 0u) + (in && br == 0 ? 1u : 0u); // bin 0 is by far the most frequent: counted per thread
        if (bf) atomicAdd(&s_hist[bf], 1u);
        if (br) atomicAdd(&s_hist[br], 1u); // background: reverse-complemented window i = p (functions.py:169)
        o_cls[j0 + j] = (uint8_t)(in ? cls : 0u);
        o_gcc[j0 + j] = (uint8_t)gcc;
        o_rbs[j0 + j] = (uint16_t)(bf | (br << 5) | ((atg & 1u) << 10) | ((atg & 2u) << 10));
    }
}

__global__ __launch_bounds__(PHX_FEAT_THREADS) void k_features(DBatch b, const DTile *__restrict__ tiles, int n_tiles) {
    __shared__ uint8_t s_code[FWX + 2 * SCPAD];
    __shared__ uint16_t s_pref[FWX];
    __shared__ uint8_t s_W[FWX + 2 * SCPAD];
    __shared__ uint32_t s_AF[FWX], s_AR[FWX];
    __shared__ uint32_t s_tab[RBS_TAB_WORDS];
    __shared__ uint8_t s_lut[256];
    __shared__ __align__(16) uint8_t o_cls[PHX_TILE], o_gcc[PHX_TILE];
    __shared__ __align__(16) uint16_t o_rbs[PHX_TILE];
    __shared__ uint32_t s_hist[28];
    __shared__ uint32_t s_scan[PHX_FEAT_THREADS / 64 + 1];
    __shared__ uint32_t s_gc, s_bad;

    const int tid = threadIdx.x;
    // once per workgroup: the motif table and the ASCII -> base-code table (bit4 = letter outside the alphabet);
    // the workgroup then walks over tiles (grid-stride)
    for (int i = tid; i < 768; i += PHX_FEAT_THREADS) s_tab[i] = b.rbs_t6[i];
    if (tid < 192) s_tab[768 + tid] = b.rbs_t5[tid];
    if (tid < 48) s_tab[960 + tid] = b.rbs_t4[tid];
    if (tid < 12) s_tab[1008 + tid] = b.rbs_t3[tid];
    if (tid == 0) s_tab[RBS_TAB_WORDS - 1] = 0;
    {
        bool bad = false;
        const uint32_t c = base_code((uint32_t)tid, bad);
        s_lut[tid] = (uint8_t)(c | (bad ? 16u : 0u));
    }
    const DParams *P = b.params;
    // the ASCII of the next tile is fetched into registers while the current tile is processed
    constexpr int NLD = (FWX + 2 * SCPAD + PHX_FEAT_THREADS - 1) / PHX_FEAT_THREADS;
    uint8_t r_asc[NLD];
    auto fetch = [&](int t) {
        const DTile tl = tiles[t];
        const DMeta *m = &b.meta[tl.contig];
        const int Ln = m->L;
        const uint8_t *__restrict__ asc = b.ascii + m->off;
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int idx = tid + j * PHX_FEAT_THREADS - SCPAD;
            const int p = tl.p0 - PHX_HALO + idx;
            const bool in = idx >= 0 && idx < FWX && p >= 0 && p < Ln;
            r_asc[j] = in ? asc[p] : (uint8_t)0; // 0 is not a nucleotide letter: maps to "outside" below
        }
    };
    if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (int ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
        const DTile tile = tiles[ti];
        DMeta *meta = &b.meta[tile.contig];
        const int L = meta->L;
        const int64_t off = meta->off;
        const int p0 = tile.p0;
        __syncthreads(); // the previous tile's LDS is no longer read
        if (tid < 28) s_hist[tid] = 0;
        if (tid == 0) { s_gc = 0; s_bad = 0; }

        // 1. stage the tile (+halo) as base codes; positions outside the contig (and the pads) read as "outside"
        uint32_t badacc = 0, mygc = 0;
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int x = tid + j * PHX_FEAT_THREADS;
            const int idx = x - SCPAD;
            const int p = p0 - PHX_HALO + idx;
            const bool in = idx >= 0 && idx < FWX && p >= 0 && p < L;
            const uint32_t c = in ? s_lut[r_asc[j]] : 12u;
            badacc |= c;
            mygc += (idx >= PHX_HALO && idx < PHX_HALO + PHX_TILE) ? (c & 1u) : 0u;
            if (x < FWX + 2 * SCPAD) s_code[x] = (uint8_t)(c & 15u);
        }
        if (ti + (int)gridDim.x < n_tiles) fetch(ti + gridDim.x);
        __syncthreads();
        const uint8_t *sc = s_code; // sc[SCPAD + idx]

        {
            // 2a. per-residue GC counts of this thread's FIT consecutive window indices (packed 3 x 10 bit)
            uint32_t loc = 0;
            if (tid < FACT) {
                const int i0 = tid * FIT;
#pragma unroll
                for (int k = 0; k < FIT; k++) loc += (uint32_t)(sc[SCPAD + i0 + k] & 1u) << (10 * (k % 3));
            }
            uint32_t tot;
            const uint32_t ex = block_excl_scan<PHX_FEAT_THREADS>(loc, s_scan, &tot);
            if (tid < FACT) {
                const int i0 = tid * FIT;
                // 2b. exclusive per-residue GC prefix over the window
                uint32_t run[3] = {ex & 1023u, (ex >> 10) & 1023u, (ex >> 20) & 1023u};
#pragma unroll
                for (int k = 0; k < FIT; k++) {
                    s_pref[i0 + k] = (uint16_t)run[k % 3];
                    run[k % 3] += sc[SCPAD + i0 + k] & 1u;
                }
            }
        }
        __syncthreads();
        if (tid < FACT) {
            // 3a. k-mer class scores with rolling 6-mer codes over the thread's consecutive indices:
            //     leftward 6-mer s[k] = dna[y-k] (forward windows) scanning up, rightward complemented 6-mer
            //     s[k] = comp(dna[y+k]) (reverse windows) scanning down
            const int i0 = tid * FIT;
            uint32_t code = 0; int v = 0;
#pragma unroll
            for (int k = 5; k >= 1; k--) {
                const uint32_t c = sc[SCPAD + i0 - k];
                code = ((code << 2) | (c & 3u)) & 4095u;
                v = (c & 4u) ? 0 : v + 1;
            }
#pragma unroll
            for (int k = 0; k < FIT; k++) {
                const uint32_t c = sc[SCPAD + i0 + k];
                code = ((code << 2) | (c & 3u)) & 4095u;
                v = (c & 4u) ? 0 : v + 1;
                s_AF[i0 + k] = kmer_lookup(s_tab, code, v);
            }
            code = 0; v = 0;
#pragma unroll
            for (int k = 5; k >= 1; k--) {
                const uint32_t c = sc[SCPAD + i0 + FIT - 1 + k];
                code = ((code << 2) | ((c & 3u) ^ 2u)) & 4095u;
                v = (c & 4u) ? 0 : v + 1;
            }
#pragma unroll
            for (int k = FIT - 1; k >= 0; k--) {
                const uint32_t c = sc[SCPAD + i0 + k];
                code = ((code << 2) | ((c & 3u) ^ 2u)) & 4095u;
                v = (c & 4u) ? 0 : v + 1;
                s_AR[i0 + k] = kmer_lookup(s_tab, code, v);
            }
        }
        // 3b. W(q) = GC count over q+3m, m in [-19,20] (gc_frame_plot.py:44-59), for the tile and 8 positions beyond
        for (int idx = PHX_HALO + tid; idx < PHX_HALO + PHX_TILE + 8; idx += PHX_FEAT_THREADS) {
            const int hi = idx + 60 < FWX ? idx + 60 : FWX - 1;
            s_W[SCPAD + idx] = (uint8_t)(s_pref[hi] + (sc[SCPAD + hi] & 1u) - s_pref[idx - 57]);
        }
        __syncthreads();

        // 4. per-position outputs; every thread owns FPT consecutive positions, the k-mer scores it needs sit in registers
        uint32_t nz0 = 0;
        if (p0 + PHX_TILE + 2 <= L) feature_outputs<true>(tid, p0, L, P, sc, s_W, s_AF, s_AR, s_hist, o_cls, o_gcc, o_rbs, nz0);
        else feature_outputs<false>(tid, p0, L, P, sc, s_W, s_AF, s_AR, s_hist, o_cls, o_gcc, o_rbs, nz0);
        // the last 20 forward background windows are right-truncated: dna[i:i+21] with i > L-21, i.e.
        // s[k] = dna[L-1-k] for k < len = L-i (functions.py:168 with python slice clipping)
        if (L - 1 >= p0 && L - 1 < p0 + PHX_TILE) {
            const int nt = L < 20 ? L : 20;
            if (tid < nt) {
                const int len = tid + 1;
                const int ie = PHX_HALO + (L - 1 - p0);
                uint32_t best = 0;
                for (int o = 3; o <= 15; o++) {
                    int vmax = len - o;
                    if (vmax < 3) break;
                    if (vmax > 6) vmax = 6;
                    uint32_t code = 0; int v = 0; bool ok = true;
                    for (int k = 0; k < vmax; k++) {
                        const uint32_t c = sc[SCPAD + ie - o - k];
                        ok = ok && !(c & 4u);
                        if (ok) v++;
                        code |= (c & 3u) << (2 * k);
                    }
                    const uint32_t scv = (kmer_lookup(s_tab, code, v) >> (8 * off_class(o))) & 0xffu;
                    best = scv > best ? scv : best;
                }
                atomicAdd(&s_hist[best], 1u);
            }
        }
        if (mygc) atomicAdd(&s_gc, mygc);
        if (nz0) atomicAdd(&s_hist[0], nz0);
        if (badacc & 16u) s_bad = 1;
        __syncthreads();
        // 5. write-out.  Full tiles with 16-byte aligned rows go out as uint4 (coalesced 1 KiB per wave instruction).
        const bool vec = p0 + PHX_TILE <= L && ((off + p0) & 15) == 0;
        if (vec) {
            uint4 *gc4 = (uint4 *)(b.cls + off + p0), *gr4 = (uint4 *)(b.rbs + off + p0);
            for (int i = tid; i < PHX_TILE / 16; i += PHX_FEAT_THREADS) gc4[i] = ((const uint4 *)o_cls)[i];
            for (int i = tid; i < PHX_TILE / 8; i += PHX_FEAT_THREADS) gr4[i] = ((const uint4 *)o_rbs)[i];
        } else {
            for (int j = tid; j < PHX_TILE && p0 + j < L; j += PHX_FEAT_THREADS) {
                b.cls[off + p0 + j] = o_cls[j];
                b.rbs[off + p0 + j] = o_rbs[j];
            }
        }
        // Bit-sliced outputs by wavefront ballot, one 64-bit word per 64 lanes:
        //  * codon bitmaps (lane <-> codon k of frame f, position f+3k): start/stop classes and the 9 GC-frame classes of the
        //    forward and of the reversed triple.  p0 is a multiple of 1536 = 3*512, so codon k of this tile is bit (k & 63)
        //    of word p0/192 + k/64;
        //  * base bitmaps (lane <-> position): unambiguous a, c, t, g.
        {
            const int lane = tid & 63, wv = tid >> 6;
            uint64_t *bits = b.bits + meta->bits_off;
            const int nw = meta->nw;
            const int wbase = p0 / 192;
            for (int pair = wv; pair < 24; pair += PHX_FEAT_THREADS / 64) {
                const int f = pair >> 3, wi = pair & 7;
                const int j = f + 3 * (64 * wi + lane);
                const uint32_t c = o_cls[j] & 7u, g = o_gcc[j];
                uint64_t m[PHX_N_CODON_BITMAPS];
                m[0] = __ballot(c == CLS_FS); m[1] = __ballot(c == CLS_RS); m[2] = __ballot(c == CLS_FT); m[3] = __ballot(c == CLS_RT);
                // GC-frame classes as marginals (2-bit fields of g): max_idx == 1, == 2, min_idx == 1, == 2 of the forward and
                // of the reversed triple (== 3 is the complement)
                m[4] = __ballot((g & 3u) == 0u); m[5] = __ballot((g & 3u) == 1u); m[6] = __ballot((g & 12u) == 0u); m[7] = __ballot((g & 12u) == 4u);
                m[8] = __ballot((g & 48u) == 0u); m[9] = __ballot((g & 48u) == 16u); m[10] = __ballot((g & 192u) == 0u); m[11] = __ballot((g & 192u) == 64u);
                uint64_t mine = 0; // lane id keeps bitmap id's word: one store instruction for all 22 words
#pragma unroll
                for (int id = 0; id < PHX_N_CODON_BITMAPS; id++) mine = lane == id ? m[id] : mine;
                if (lane < PHX_N_CODON_BITMAPS) bits[(size_t)(lane * 3 + f) * nw + wbase + wi] = mine;
            }
            uint64_t *bb = bits + (size_t)PHX_N_CODON_BITMAPS * 3 * nw; // base bitmaps: [a,c,t,g][3*nw words over positions]
            const int pbase = p0 / 64;
            for (int w = wv; w < PHX_TILE / 64; w += PHX_FEAT_THREADS / 64) {
                const uint32_t c = sc[SCPAD + PHX_HALO + 64 * w + lane]; // invalid (bit 2) for ambiguity codes and outside the contig
                const uint64_t ma = __ballot(c == 0u), mc = __ballot(c == 1u), mt = __ballot(c == 2u), mg = __ballot(c == 3u);
                if (lane < 4) bb[(size_t)lane * 3 * nw + pbase + w] = lane == 0 ? ma : (lane == 1 ? mc : (lane == 2 ? mt : mg));
            }
        }
        if (tid < 28 && s_hist[tid]) atomicAdd(&meta->bg[tid], s_hist[tid]);
        if (tid == 0) {
            if (s_gc) atomicAdd(&meta->gc, s_gc);
            if (s_bad) atomicMin(&meta->status, PHX_S_BADLETTER);
        }
    } // tiles
}

// ------------------------------------------------------------------------------------------------
// ORF scan on the codon bitmaps.  A "stop event" (forward stop codon, or the closing reverse-complement stop
// codon) owns its stop-group (orfs.py:17-32).  Work item = one 64-codon word of one (strand, frame):
// k_orf<false> counts ORFs / groups per item and turns the counts into exclusive offsets with one block
// scan; k_orf<true> re-walks the items and writes the records.  The device order of groups is therefore
// (strand, frame, codon); the reference's insertion order is kept as DGrp.evkey.
struct OrfOut {
    DOrf *orf;
    DGrp *grp;
    unsigned long long *nbF, *nbR; // node-existence bitmaps over positions: forward-strand / reverse-strand node at that key
};
__device__ __forceinline__ void mark_node(unsigned long long *nb, int q) { atomicOr(&nb[q >> 6], 1ull << (q & 63)); }
struct FrameBits {
    const uint64_t *FS, *RS, *FT, *RT;
    int f;    // 0-based frame = position of codon 0
    int ncod; // complete codons in this frame
};
// highest set bit with index < k, or -1
__device__ __forceinline__ int prev_bit(const uint64_t *__restrict__ B, int k) {
    if (k <= 0) return -1;
    int w = (k - 1) >> 6;
    uint64_t m = B[w] & (~0ull >> (63 - ((k - 1) & 63)));
    while (true) {
        if (m) return (w << 6) + 63 - __clzll((long long)m);
        if (--w < 0) return -1;
        m = B[w];
    }
}
__device__ __forceinline__ uint64_t range_mask(int w, int lo, int hi) { // bits of word w that lie in [lo, hi]
    uint64_t m = ~0ull;
    if ((lo >> 6) == w) m &= ~0ull << (lo & 63);
    if ((hi >> 6) == w) m &= ~0ull >> (63 - (hi & 63));
    return m;
}

// Forward group closed by the stop codon (or, for the end fragment, the last codon of the frame) k:
// functions.py:202-214 / 229-239.  Starts are emitted nearest first == reversed(starts[frame]).
template <bool EMIT>
__device__ int fwd_group(const FrameBits &F, const uint8_t *__restrict__ cls, const uint16_t *__restrict__ rbs, int k, int dmin,
                         OrfOut o, int obase, int gidx, int evkey) {
    const int kp = prev_bit(F.FT, k);
    const int lo = kp + 1, hi = k - dmin; // start codons ks with stop+2-start+1 >= minlen
    const int p = F.f + 3 * k;
    int n = 0;
    if (hi >= lo) {
        for (int w = hi >> 6; w >= (lo >> 6); w--) {
            uint64_t m = F.FS[w] & range_mask(w, lo, hi);
            if (!EMIT) n += __popcll(m);
            else
                while (m) {
                    const int j = 63 - __clzll((long long)m);
                    m &= ~(1ull << j);
                    const int q = F.f + 3 * ((w << 6) + j);
                    DOrf *r = &o.orf[obase + n];
                    r->start = q + 1; r->stop = p + 1; r->frame = (int8_t)(F.f + 1);
                    r->rbs = (uint8_t)(q >= 20 ? (rbs[q] & 31u) : 0u); // dna[start-21:start] is empty for start < 21 (functions.py:208)
                    r->startidx = (int8_t)((cls[q] >> 3) & 15);
                    r->flags = (uint8_t)((rbs[q] >> 10) & 1u);
                    r->grp = gidx; r->node = -1;
                    mark_node(o.nbF, q);
                    n++;
                }
        }
    }
    // no stop to the left: the frame opens with a pseudo-start unless its first codon is a start (functions.py:186-191)
    if (kp < 0 && !(F.FS[0] & 1ull) && hi >= 0 && k > 0) {
        if (EMIT) {
            const int q0 = F.f;
            DOrf *r = &o.orf[obase + n];
            r->start = q0 + 1; r->stop = p + 1; r->frame = (int8_t)(F.f + 1);
            r->rbs = 0; // start <= 3 < 21
            r->startidx = -1;
            r->flags = (uint8_t)((rbs[q0] >> 10) & 1u);
            r->grp = gidx; r->node = -1;
            mark_node(o.nbF, q0);
        }
        n++;
    }
    if (EMIT && n) {
        DGrp *g = &o.grp[gidx];
        g->stop = p + 1; g->orf_begin = obase; g->n = n; g->node = -1; g->frame = F.f + 1; g->evkey = evkey;
        mark_node(o.nbF, p);
    }
    return n;
}

// Reverse group: ORFs between the previous rc-stop of the frame (or the frame's first codon: stops =
// {-1:1,-2:2,-3:3}, functions.py:184) and the pending rc-starts, emitted when the closing rc-stop k is
// met (functions.py:215-227), or after the loop for the open group (virt: functions.py:240-251, with a
// pseudo-start on the last codon unless rev_comp(codon) is a start codon).  Starts ascend.
template <bool EMIT>
__device__ int rev_group(const FrameBits &F, const uint8_t *__restrict__ cls, const uint16_t *__restrict__ rbs, int k, bool virt, int L,
                         int dmin, OrfOut o, int obase, int gidx, int evkey) {
    const int kp = prev_bit(F.RT, virt ? F.ncod : k);
    const int sk = kp >= 0 ? kp : 0;
    const int lo = sk + dmin, hi = virt ? F.ncod - 1 : k - 1;
    const int psk = F.f + 3 * sk;
    int n = 0;
    if (hi >= lo) {
        for (int w = lo >> 6; w <= (hi >> 6); w++) {
            uint64_t m = F.RS[w] & range_mask(w, lo, hi);
            if (!EMIT) n += __popcll(m);
            else
                while (m) {
                    const int j = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int s = F.f + 3 * ((w << 6) + j);
                    DOrf *r = &o.orf[obase + n];
                    r->start = s + 1; r->stop = psk + 1; r->frame = (int8_t)(-(F.f + 1));
                    const int jj = s + 3; // dna[start:start+21] with start = i+2 (functions.py:221)
                    r->rbs = (uint8_t)(jj < L ? ((rbs[jj] >> 5) & 31u) : 0u);
                    r->startidx = (int8_t)((cls[s] >> 3) & 15);
                    r->flags = (uint8_t)((rbs[s] >> 11) & 1u);
                    r->grp = gidx; r->node = -1;
                    mark_node(o.nbR, s);
                    n++;
                }
        }
    }
    if (virt && F.ncod - 1 >= lo) {
        const int s = F.f + 3 * (F.ncod - 1);
        if (!(cls[s] & 0x80u)) { // functions.py:241-242
            if (EMIT) {
                DOrf *r = &o.orf[obase + n];
                r->start = s + 1; r->stop = psk + 1; r->frame = (int8_t)(-(F.f + 1));
                const int jj = s + 3;
                r->rbs = (uint8_t)(jj < L ? ((rbs[jj] >> 5) & 31u) : 0u);
                r->startidx = -1;
                r->flags = (uint8_t)((rbs[s] >> 11) & 1u);
                r->grp = gidx; r->node = -1;
                mark_node(o.nbR, s);
            }
            n++;
        }
    }
    if (EMIT && n) {
        DGrp *g = &o.grp[gidx];
        g->stop = psk + 1; g->orf_begin = obase; g->n = n; g->node = -1; g->frame = -(F.f + 1); g->evkey = evkey;
        mark_node(o.nbR, psk);
    }
    return n;
}

__device__ __forceinline__ FrameBits frame_bits(const uint64_t *bits, int nw, int f, int L) {
    FrameBits F;
    F.FS = bits + (size_t)(0 * 3 + f) * nw; F.RS = bits + (size_t)(1 * 3 + f) * nw;
    F.FT = bits + (size_t)(2 * 3 + f) * nw; F.RT = bits + (size_t)(3 * 3 + f) * nw;
    F.f = f;
    F.ncod = L - f >= 3 ? (L - f) / 3 : 0;
    return F;
}

template <bool EMIT>
__global__ __launch_bounds__(NT) void k_orf(DBatch b) {
    if (EMIT && b.tot->overflow) return;
    __shared__ uint32_t s_scan[NT / 64 + 1];
    DMeta *meta = &b.meta[blockIdx.x];
    const int L = meta->L;
    if (meta->status < 0 || L < 6) {
        if (threadIdx.x == 0) {
            if (L < 6 && meta->status == 0) meta->status = PHX_S_TOOSHORT;
            meta->n_orf = 0; meta->n_grp = 0;
        }
        return;
    }
    const int64_t off = meta->off;
    const uint8_t *__restrict__ cls = b.cls + off;
    const uint16_t *__restrict__ rbs = b.rbs + off;
    const uint64_t *bits = b.bits + meta->bits_off;
    const int nw = meta->nw;
    uint2 *item = b.item + meta->item_off;
    const int minlen = b.params->minlen;
    const int dmin = (minlen - 1) / 3; // codons between start and stop so that the ORF length reaches minlen
    OrfOut o;
    o.orf = EMIT ? b.orf + meta->orf_off : nullptr;
    o.grp = EMIT ? b.grp + meta->grp_off : nullptr;
    o.nbF = (unsigned long long *)(b.nbits + meta->nbits_off);
    o.nbR = o.nbF + 3 * (size_t)nw;
    const int nitems = 6 * nw;
    // count pass: one workgroup per contig (it needs the block scan); emit pass: the items are spread over
    // gridDim.y workgroups, every thread takes one item at a time (offsets are already known)
    const int nthr = EMIT ? NT * (int)gridDim.y : NT;
    const int gtid = EMIT ? (int)blockIdx.y * NT + (int)threadIdx.x : (int)threadIdx.x;
    const int per = EMIT ? 1 : (nitems + NT - 1) / NT;
    const int ia = EMIT ? gtid : (int)threadIdx.x * per, ib = EMIT ? nitems : (ia + per < nitems ? ia + per : nitems);
    const int istep = EMIT ? nthr : 1;
    uint32_t so = 0, sg = 0;
    for (int it = ia; it < ib; it += istep) {
        const int sf = it / nw, w = it - sf * nw;
        const int s = sf / 3, f = sf - 3 * s;
        const FrameBits F = frame_bits(bits, nw, f, L);
        uint64_t m = (s == 0 ? F.FT : F.RT)[w];
        uint32_t no = 0, ng = 0;
        uint2 base = make_uint2(0, 0);
        if (EMIT) base = item[it];
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int k = (w << 6) + j;
            const int ev = f + 3 * k; // main-loop events are discovered in position order (functions.py:195)
            const int n = s == 0 ? fwd_group<EMIT>(F, cls, rbs, k, dmin, o, (int)(base.x + no), (int)(base.y + ng), ev)
                                 : rev_group<EMIT>(F, cls, rbs, k, false, L, dmin, o, (int)(base.x + no), (int)(base.y + ng), ev);
            no += (uint32_t)n;
            ng += n ? 1u : 0u;
        }
        if (!EMIT) item[it] = make_uint2(no, ng);
        so += no; sg += ng;
    }
    int run_orf, run_grp;
    if (!EMIT) {
        uint32_t tot, gtot;
        uint32_t exo = block_excl_scan<NT>(so, s_scan, &tot);
        uint32_t exg = block_excl_scan<NT>(sg, s_scan, &gtot);
        for (int it = ia; it < ib; it++) {
            const uint2 c = item[it];
            item[it] = make_uint2(exo, exg);
            exo += c.x; exg += c.y;
        }
        run_orf = (int)tot; run_grp = (int)gtot;
    } else {
        run_orf = meta->n_orf_main; run_grp = meta->n_grp_main;
    }
    // fragments at the right end, functions.py:229-251: frame 1 fwd, frame 1 rev, frame 2 fwd, ...
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        if (!EMIT) { meta->n_orf_main = run_orf; meta->n_grp_main = run_grp; }
        for (int f = 0; f < 3; f++) {
            const FrameBits F = frame_bits(bits, nw, f, L);
            if (F.ncod < 1) continue;
            const int ke = F.ncod - 1;
            if (!((F.FT[ke >> 6] >> (ke & 63)) & 1ull)) { // else the main loop closed the group and starts[frame] is empty
                const int n = fwd_group<EMIT>(F, cls, rbs, ke, dmin, o, run_orf, run_grp, L + 2 * f);
                run_orf += n; run_grp += n ? 1 : 0;
            }
            const int n = rev_group<EMIT>(F, cls, rbs, ke, true, L, dmin, o, run_orf, run_grp, L + 2 * f + 1);
            run_orf += n; run_grp += n ? 1 : 0;
        }
        if (!EMIT) { meta->n_orf = run_orf; meta->n_grp = run_grp; }
        else if (run_orf != meta->n_orf || run_grp != meta->n_grp) meta->status = PHX_E_STATE; // cannot happen
    }
}

// ------------------------------------------------------------------------------------------------
// ORF statistics on the bit-sliced features, one thread per ORF: the 3x3 GC-frame class histogram over the sense
// codons (functions.py:286-298) is a popcount over a codon range of 9 class bitmaps, p_stop (orfs.py:162-173) a
// popcount over a position range of the a/t/g(/c) base bitmaps.  Then one thread per stop-group for the GC frame
// plot training of functions.py:261-279 (same popcounts over a sub-range of the first 'atg' ORF).
__device__ __forceinline__ uint32_t popc_range(const uint64_t *__restrict__ B, int lo, int hi) { // set bits with index in [lo, hi]
    if (hi < lo) return 0u;
    uint32_t n = 0;
    for (int w = lo >> 6; w <= (hi >> 6); w++) n += (uint32_t)__popcll(B[w] & range_mask(w, lo, hi));
    return n;
}
// 3x3 histogram (max_idx-1)*3 + (min_idx-1) over the codons [lo, hi] of one frame from the four marginal bitmaps
// M[0] = max_idx==1, M[1] = max_idx==2, M[2] = min_idx==1, M[3] = min_idx==2 (stride = distance between bitmaps)
__device__ __forceinline__ void class_hist(const uint64_t *__restrict__ M, size_t stride, int lo, int hi, uint32_t h[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++) h[i] = 0;
    if (hi < lo) return;
    for (int w = lo >> 6; w <= (hi >> 6); w++) {
        const uint64_t rm = range_mask(w, lo, hi);
        const uint64_t x0 = M[w], x1 = M[stride + w], n0 = M[2 * stride + w], n1 = M[3 * stride + w];
        const uint64_t x[3] = {x0 & rm, x1 & rm, ~(x0 | x1) & rm};
        const uint64_t n[3] = {n0, n1, ~(n0 | n1)};
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int c = 0; c < 3; c++) h[a * 3 + c] += (uint32_t)__popcll(x[a] & n[c]);
    }
}

__global__ __launch_bounds__(NT) void k_orf_stats(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    DMeta *meta = &b.meta[blockIdx.x];
    if (meta->status < 0) return;
    DOrf *orf = b.orf + meta->orf_off;
    const DGrp *grp = b.grp + meta->grp_off;
    const int nw = meta->nw;
    const uint64_t *bits = b.bits + meta->bits_off;
    const uint64_t *bb = bits + (size_t)PHX_N_CODON_BITMAPS * 3 * nw; // [a,c,t,g][3*nw]
    const int gtid = (int)blockIdx.y * NT + (int)threadIdx.x, gstride = (int)gridDim.y * NT;
    bool ovf = false;
    for (int k = gtid; k < meta->n_orf; k += gstride) {
        DOrf *r = &orf[k];
        const int start = r->start, stop = r->stop;
        const bool fwd = r->frame > 0;
        const int f = (fwd ? r->frame : -r->frame) - 1;
        // seq = positions [lo1, hi1] (1-based): fwd start..stop+2, rev stop..start+2 (functions.py:207,220)
        const int lo1 = fwd ? start : stop, hi1 = (fwd ? stop : start) + 2;
        const int ncod = (hi1 - lo1 + 1) / 3;
        if (ncod > 65535) ovf = true;
        // sense codons: fwd codon indices [k0, k1) from start to just before the stop / last codon (functions.py:290);
        // rev (k0, k1] from just after the stop key up to the start codon (functions.py:295)
        const int k0 = (lo1 - 1 - f) / 3, k1 = k0 + ncod - 1;
        const int clo = fwd ? k0 : k0 + 1, chi = fwd ? k1 - 1 : k1;
        uint32_t h[9];
        class_hist(bits + (size_t)((fwd ? 4 : 8) * 3 + f) * nw, (size_t)3 * nw, clo, chi, h);
        for (int cl = 0; cl < 9; cl++) r->hist[cl] = (uint16_t)h[cl];
        uint32_t na = popc_range(bb + (size_t)0 * 3 * nw, lo1 - 1, hi1 - 1), nc = popc_range(bb + (size_t)1 * 3 * nw, lo1 - 1, hi1 - 1);
        uint32_t nt = popc_range(bb + (size_t)2 * 3 * nw, lo1 - 1, hi1 - 1), ng = popc_range(bb + (size_t)3 * 3 * nw, lo1 - 1, hi1 - 1);
        if (!fwd) { const uint32_t t = na; na = nt; nt = t; ng = nc; } // coding strand: a<->t, g<->c
        // Orf.p_stop, orfs.py:162-173
        const double n = (double)(3 * ncod);
        const double Pa = (double)na / n, Pt = (double)nt / n, Pg = (double)ng / n;
        r->pstop = Pt * Pa * Pa + Pt * Pg * Pa + Pt * Pa * Pg;
        atomicAdd(&meta->tr[r->rbs], 1u); // training_rbs, functions.py:211,224,239,251
    }
    // GC frame plot training: per group, the first ORF longest->shortest whose start codon is 'atg'
    for (int g = gtid; g < meta->n_grp; g += gstride) {
        const DGrp G = grp[g];
        int pick = -1; // emission order is nearest-first, iter_in is farthest-first (orfs.py:38-46): search from the back
        for (int k = G.n - 1; k >= 0; k--)
            if (orf[G.orf_begin + k].flags & 1) { pick = k; break; }
        if (pick < 0) continue;
        const DOrf *r = &orf[G.orf_begin + pick];
        const int start = r->start, stop = r->stop;
        const bool fwd = start < stop;
        const int f = (start - 1) % 3;
        int clo, chi; // codon index range (inclusive) of the bases visited by the training loop
        if (fwd) { // range(start+n, stop-36, 3), functions.py:270-271
            const int nn = (int)((double)(stop - start) / 8.0) * 3;
            clo = (start + nn - 1 - f) / 3;
            chi = (stop - 36 - 1 - 1 - f) / 3; // last base < stop-36 in this frame
            if (stop - 36 - 1 - 1 - f < 0) chi = -1;
        } else if (stop < start) { // range(start-n, stop+36, -3), functions.py:275-276
            const int nn = (int)((double)(start - stop) / 8.0) * 3;
            chi = (start - nn - 1 - f) / 3;
            clo = (stop + 36 - 1 - f) / 3 + 1; // first base > stop+36 in this frame
        } else continue;
        uint32_t mx[3] = {0, 0, 0}, mn[3] = {0, 0, 0};
        {
            uint32_t h[9];
            class_hist(bits + (size_t)((fwd ? 4 : 8) * 3 + f) * nw, (size_t)3 * nw, clo, chi, h);
            for (int cl = 0; cl < 9; cl++) { mx[cl / 3] += h[cl]; mn[cl % 3] += h[cl]; }
        }
        for (int i = 0; i < 3; i++) {
            if (mx[i]) atomicAdd(&meta->pmax[i + 1], mx[i]);
            if (mn[i]) atomicAdd(&meta->pmin[i + 1], mn[i]);
        }
    }
    if (ovf) atomicMin(&meta->status, PHX_S_OVERFLOW);
}

// ORF weight: functions.py:254-257 (RBS), 281-284 (normalise), 286-301 + orfs.py:122-127.
// hold = prod ((1-pstop)^pos_max[imax])^pos_min[imin] = (1-pstop)^S ; weight = -(1/hold)*w_start*w_rbs
__global__ __launch_bounds__(NT) void k_score(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    __shared__ int s_maxexp;
    __shared__ double s_wsum;
    DMeta *meta = &b.meta[blockIdx.x];
    if (meta->status < 0) return;
    if (threadIdx.x == 0) { s_maxexp = 0; s_wsum = 0.0; }
    __syncthreads();
    DOrf *orf = b.orf + meta->orf_off;
    const DParams *P = b.params;
    double bgs = 0, trs = 0;
    for (int i = 0; i < 28; i++) { bgs += 1.0 + (double)meta->bg[i]; trs += 1.0 + (double)meta->tr[i]; }
    double pmax[4], pmin[4];
    {
        double ymx = 1.0, ymn = 1.0;
        for (int i = 0; i < 4; i++) {
            pmax[i] = 1.0 + (double)(i ? meta->pmax[i] : 0u);
            pmin[i] = 1.0 + (double)(i ? meta->pmin[i] : 0u);
            ymx = pmax[i] > ymx ? pmax[i] : ymx;
            ymn = pmin[i] > ymn ? pmin[i] : ymn;
        }
        for (int i = 0; i < 4; i++) { pmax[i] /= ymx; pmin[i] /= ymn; }
    }
    int mymax = 0;
    double mysum = 0.0;
    for (int k = threadIdx.x; k < meta->n_orf; k += NT) {
        DOrf *r = &orf[k];
        double S = 0;
        for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) S += (double)r->hist[a * 3 + c] * (pmax[a + 1] * pmin[c + 1]);
        const double tr = (1.0 + (double)meta->tr[r->rbs]) / trs;
        const double bg = (1.0 + (double)meta->bg[r->rbs]) / bgs;
        const double w_rbs = tr / bg;
        double s = exp(-S * log1p(-r->pstop));
        if (r->startidx >= 0) s = s * P->start_w[r->startidx];
        s = s * w_rbs;
        r->weight = -s;
        int e;
        frexp(s * 1000.0, &e);
        if (!(s < 1.0e300)) e = 4096; // inf / nan: force the overflow status
        mymax = e > mymax ? e : mymax;
        if (s < 1.0e300) mysum += s * 1000.0;
    }
    if (mymax) { atomicMax(&s_maxexp, mymax); atomicAdd(&s_wsum, mysum); }
    __syncthreads();
    if (threadIdx.x == 0) { meta->maxexp = s_maxexp; meta->wsum = s_wsum; }
}

// ------------------------------------------------------------------------------------------------
// Nodes: one per ORF start and one per stop-group, sorted by position (then forward before reverse).
// k_orf<true> has set one bit per node in two position bitmaps (forward-strand slot, reverse-strand slot); a node's
// id is its rank = popcounts below it.  k_node_cov marks the bases covered by the longest ORF of every stop-group
// in a third bitmap; k_node_rank turns the per-word popcounts into bases and finds the >500 bp uncovered runs of
// functions.py:320-334; k_node_build lets every ORF / group write its own node; k_node_attr adds other_end and o1/o2.
struct LinkInfo { int time; int val; bool stop; int idx; int far; };
// time = when the reference wrote other_end[pos] for this slot (all ORFs of one group are added by one
// event, so the group's evkey orders the writers); val = what it wrote last; far = index of the group's
// farthest ORF when the slot is a stop node.
__device__ __forceinline__ LinkInfo link_info(uint32_t link, const DOrf *orf, const DGrp *grp) {
    LinkInfo r;
    r.idx = (int)LINK_IDX(link);
    if (LINK_KIND(link) == LINK_START) { r.stop = false; r.far = -1; r.time = grp[orf[r.idx].grp].evkey; r.val = orf[r.idx].stop; }
    else { r.stop = true; const DGrp g = grp[r.idx]; r.far = g.orf_begin + g.n - 1; r.time = g.evkey; r.val = orf[r.far].start; }
    return r;
}

// coverage by the longest ORF of every stop-group (functions.py:321-330), 16 lanes per group, one atomicOr per word
__global__ __launch_bounds__(NT) void k_node_cov(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    DMeta *meta = &b.meta[blockIdx.x];
    if (meta->status < 0) return;
    const int L = meta->L;
    const DOrf *orf = b.orf + meta->orf_off;
    const DGrp *grp = b.grp + meta->grp_off;
    unsigned long long *cov = (unsigned long long *)(b.nbits + meta->nbits_off) + 6 * (size_t)meta->nw;
    const int sub = threadIdx.x & 15;
    for (int g = (int)blockIdx.y * (NT / 16) + ((int)threadIdx.x >> 4); g < meta->n_grp; g += (int)gridDim.y * (NT / 16)) {
        const DGrp G = grp[g];
        const DOrf *r = &orf[G.orf_begin + G.n - 1];
        int mi = r->start < r->stop ? r->start : r->stop;
        int ma = r->start > r->stop ? r->start : r->stop;
        if (ma > L - 1) ma = L - 1;
        if (ma <= mi) continue;
        for (int w = (mi >> 6) + sub; w <= ((ma - 1) >> 6); w += 16) atomicOr(&cov[w], range_mask(w, mi, ma - 1)); // bases[n] = n for n in [mi, ma)
    }
}

// per-word node-rank bases, and the bridge events (a covered base more than 500 after the previous covered base)
__global__ __launch_bounds__(NT) void k_node_rank(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    __shared__ uint32_t s_scan[NT / 64 + 1];
    DMeta *meta = &b.meta[blockIdx.x];
    const int tid = threadIdx.x;
    if (meta->status < 0) {
        if (tid == 0) { meta->n_node = 0; meta->n_edge = 0; }
        return;
    }
    const int nwp = 3 * meta->nw;
    const uint64_t *nbF = b.nbits + meta->nbits_off, *nbR = nbF + nwp, *cov = nbR + nwp;
    uint32_t *nbase = b.nbase + meta->nbits_off / 3;
    if (tid == 0) meta->n_bridge = 0;
    __syncthreads();
    const int per = (nwp + NT - 1) / NT;
    const int a = tid * per, e = a + per < nwp ? a + per : nwp;
    uint32_t cnt = 0, lastc = 0; // nodes in my words; last covered base in my words (0 = none; base 0 is never covered)
    for (int w = a; w < e; w++) {
        cnt += (uint32_t)(__popcll(nbF[w]) + __popcll(nbR[w]));
        if (cov[w]) lastc = (uint32_t)(w * 64 + 63 - __clzll((long long)cov[w]));
    }
    uint32_t tot, mtot;
    uint32_t ex = block_excl_scan<NT>(cnt, s_scan, &tot);
    uint32_t pm = block_excl_max<NT>(lastc, s_scan, &mtot);
    for (int w = a; w < e; w++) {
        nbase[w] = ex;
        ex += (uint32_t)(__popcll(nbF[w]) + __popcll(nbR[w]));
        if (cov[w]) {
            const int first = w * 64 + __ffsll((long long)cov[w]) - 1;
            if (first - (int)pm > 500) { // functions.py:334 (gaps inside one 64-base word cannot exceed 500)
                const int k = atomicAdd(&meta->n_bridge, 1);
                if (k < PHX_MAX_BRIDGE) { meta->bridge[k].last = (int)pm; meta->bridge[k].base = first; }
            }
            pm = (uint32_t)(w * 64 + 63 - __clzll((long long)cov[w]));
        }
    }
    if (tid == 0) {
        const int run = (int)tot, L = meta->L;
        int32_t *npos = b.npos + meta->node_off, *ninfo = b.ninfo + meta->node_off, *nother = b.nother + meta->node_off;
        uint32_t *nlink = b.nlink + meta->node_off;
        double *no = b.no + meta->node_off;
        const double pgap = contig_pstop(meta->gc, L);
        if (run + 2 != meta->n_node) meta->status = PHX_E_STATE; // n_node was sized as n_orf + n_grp + 2
        else {
            npos[run] = 0; ninfo[run] = NINFO(2, 0); nlink[run] = 0; nother[run] = -1; no[run] = pgap;                           // source, functions.py:440
            npos[run + 1] = L + 1; ninfo[run + 1] = NINFO(3, 0); nlink[run + 1] = 0; nother[run + 1] = -1; no[run + 1] = pgap; // target
        }
    }
    __syncthreads();
    if (tid == 0 && meta->n_bridge > PHX_MAX_BRIDGE) meta->status = PHX_S_OVERFLOW;
}

// node id of the node at 0-based key position q in the forward (rev = false) or reverse slot
__device__ __forceinline__ int node_rank(const uint64_t *nbF, const uint64_t *nbR, const uint32_t *nbase, int q, bool rev) {
    const int w = q >> 6;
    const uint64_t below = (q & 63) ? (~0ull >> (64 - (q & 63))) : 0ull;
    int id = (int)nbase[w] + __popcll(nbF[w] & below) + __popcll(nbR[w] & below);
    if (rev) id += (int)((nbF[w] >> (q & 63)) & 1ull);
    return id;
}

// every ORF writes its start node, every stop-group its stop node (functions.py:311-318)
__global__ __launch_bounds__(NT) void k_node_build(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    DMeta *meta = &b.meta[blockIdx.x];
    if (meta->status < 0 || meta->n_node <= 2) return;
    const int nwp = 3 * meta->nw;
    const uint64_t *nbF = b.nbits + meta->nbits_off, *nbR = nbF + nwp;
    const uint32_t *nbase = b.nbase + meta->nbits_off / 3;
    DOrf *orf = b.orf + meta->orf_off;
    DGrp *grp = b.grp + meta->grp_off;
    int32_t *npos = b.npos + meta->node_off, *ninfo = b.ninfo + meta->node_off;
    uint32_t *nlink = b.nlink + meta->node_off;
    unsigned long long *cF = (unsigned long long *)(b.cbits + meta->cb_off), *cR = cF + meta->ncw;
    const int gtid = (int)blockIdx.y * NT + (int)threadIdx.x, gstride = (int)gridDim.y * NT;
    for (int k = gtid; k < meta->n_orf; k += gstride) {
        DOrf *r = &orf[k];
        const int id = node_rank(nbF, nbR, nbase, r->start - 1, r->frame < 0);
        npos[id] = r->start; ninfo[id] = NINFO(0, r->frame); nlink[id] = LINK_START | (uint32_t)k;
        r->node = id;
        if (r->frame < 0) atomicOr(&cR[id >> 6], 1ull << (id & 63)); // reverse start = close node
    }
    for (int g = gtid; g < meta->n_grp; g += gstride) {
        DGrp *G = &grp[g];
        const int id = node_rank(nbF, nbR, nbase, G->stop - 1, G->frame < 0);
        npos[id] = G->stop; ninfo[id] = NINFO(1, G->frame); nlink[id] = LINK_STOP | (uint32_t)g;
        G->node = id;
        if (G->frame > 0) atomicOr(&cF[id >> 6], 1ull << (id & 63)); // forward stop = close node
    }
}

// other_end[pos] (last writer wins, orfs.py:19-30) and the o1/o2 term (functions.py:373-384), thread per node
__global__ __launch_bounds__(NT) void k_node_attr(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    DMeta *meta = &b.meta[blockIdx.x];
    if (meta->status < 0 || meta->n_node <= 2) return;
    const int L = meta->L;
    const DOrf *orf = b.orf + meta->orf_off;
    const DGrp *grp = b.grp + meta->grp_off;
    const int32_t *npos = b.npos + meta->node_off, *ninfo = b.ninfo + meta->node_off;
    const uint32_t *nlink = b.nlink + meta->node_off;
    int32_t *nother = b.nother + meta->node_off;
    double *no = b.no + meta->node_off;
    const double pgap = contig_pstop(meta->gc, L);
    const int run = meta->n_node - 2;
    for (int v = (int)blockIdx.y * NT + (int)threadIdx.x; v < run; v += (int)gridDim.y * NT) {
        const int fr = NFRAME(ninfo[v]);
        const uint32_t lmine = nlink[v];
        // the node of the other strand at the same position, if any, is my neighbour in the id order (forward first)
        uint32_t lother = 0;
        if (fr > 0) { if (v + 1 < run && npos[v + 1] == npos[v]) lother = nlink[v + 1]; }
        else { if (v >= 1 && npos[v - 1] == npos[v]) lother = nlink[v - 1]; }
        LinkInfo a = link_info(lmine, orf, grp);
        int oe = a.val;
        double o = pgap;
        if (!lother) {
            if (a.stop) o = orf[a.far].pstop; // longest ORF of the group
        } else {
            LinkInfo c = link_info(lother, orf, grp);
            const bool other_wins = c.time > a.time;
            if (other_wins) oe = c.val;
            if (a.stop || c.stop) {
                const LinkInfo sl = a.stop ? a : c; // the slot that makes `l in my_orfs` true
                const LinkInfo st = a.stop ? c : a;
                const DGrp G = grp[sl.idx];
                int hit = -1;
                for (int k = 0; k < G.n; k++)
                    if (orf[G.orf_begin + k].start == oe) { hit = G.orf_begin + k; break; }
                if (hit >= 0) o = orf[hit].pstop;        // get_orf(other_end[l], l)
                else if (!st.stop) o = orf[st.idx].pstop; // get_orf(l, other_end[l])
            }
        }
        nother[v] = oe;
        no[v] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Edges, enumerated per destination node (CSR by destination).
// score_gap functions.py:36-46, score_overlap functions.py:26-34.
__device__ __forceinline__ double score_gap(int length, bool diff, double pgap) {
    const double g = 1.0 - pgap;
    if (length > 300) return pow(g, 100.0) + (double)length;
    double s = 1.0 / pow(g, (double)length / 3.0);
    if (diff) s += 20.0;
    return s;
}
__device__ __forceinline__ double score_overlap(int length, bool diff, double pstop) {
    double s = 1.0 / pow(1.0 - pstop, (double)length);
    if (diff) s += 20.0;
    return s;
}

struct EdgeSink {
    uint32_t *esrc;
    double *ew;
    int n;
    bool defer; // overlap weights are finished by k_edge_weights (node ids must fit 21 bits)
};
template <bool FILL>
__device__ __forceinline__ void emit_edge(EdgeSink &s, int src, double w) {
    if (FILL) { s.esrc[s.n] = (uint32_t)src; s.ew[s.n] = w; }
    s.n++;
}
// Overlap edge r -> l: 1/(1-pstop)^length (+20 across strands), functions.py:26-34.  pow() inside the divergent
// neighbour scan would run for the whole wavefront whenever one lane needs it, so the scan only records
// (source, length, direction, pstop) and k_edge_weights evaluates the power with every lane busy.
#define EDGE_PENDING 0x80000000u
template <bool FILL>
__device__ __forceinline__ void emit_overlap(EdgeSink &s, int src, int length, bool diff, double ps) {
    if (FILL) {
        if (s.defer) { s.esrc[s.n] = (uint32_t)src | ((uint32_t)length << 21) | (diff ? 0x40000000u : 0u) | EDGE_PENDING; s.ew[s.n] = ps; }
        else { s.esrc[s.n] = (uint32_t)src; s.ew[s.n] = score_overlap(length, diff, ps); }
    }
    s.n++;
}

// Thread per destination node.  Every connector edge ends in an open node (forward start, reverse stop, target) and
// starts in a close node (forward stop, reverse start) less than 500 bp away, so the candidates of a node are the set
// bits of two node-id bitmaps (close nodes of either strand, written by k_node_build) inside an id range that the
// position-rank structure gives in O(1):
//   * left neighbours (gap edges, functions.py:401-405,417-419,427-433): EVERY close node in the range is an edge
//     (one exception: forward start <- reverse start needs r-l > 2), so the count is a popcount;
//   * right neighbours (overlap edges, functions.py:406-416,423-426,434-438): the other_end tests decide per candidate.
template <bool FILL>
__global__ __launch_bounds__(NT) void k_edges(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    DMeta *meta = &b.meta[blockIdx.x];
    if (meta->status < 0 || meta->n_node <= 0) return;
    const int L = meta->L;
    const int V = meta->n_node, ncds = V - 2, SRC = V - 2, TGT = V - 1;
    const DOrf *orf = b.orf + meta->orf_off;
    const DGrp *grp = b.grp + meta->grp_off;
    const int32_t *npos = b.npos + meta->node_off, *ninfo = b.ninfo + meta->node_off, *nother = b.nother + meta->node_off;
    const uint32_t *nlink = b.nlink + meta->node_off;
    const double *no = b.no + meta->node_off;
    uint32_t *in_off = b.in_off + meta->node_off + blockIdx.x; // V+1 entries per contig
    const int nwp = 3 * meta->nw;
    const uint64_t *nbF = b.nbits + meta->nbits_off, *nbR = nbF + nwp;
    const uint32_t *nbase = b.nbase + meta->nbits_off / 3;
    const uint64_t *cF = b.cbits + meta->cb_off, *cR = cF + meta->ncw; // close nodes by node id: forward stops / reverse starts
    const double pgap = contig_pstop(meta->gc, L);
    const int nbr = meta->n_bridge;
    bool parallel = false;
    // score_gap depends only on (length, direction): tabulate 1/(1-pgap)^(length/3) for length -2..300 once per
    // workgroup instead of one pow per gap edge; length > 300 is (1-pgap)^100 + length (functions.py:40-41)
    __shared__ double s_gap[304];
    __shared__ double s_g100;
    if (FILL) {
        for (int i = threadIdx.x; i < 303; i += NT) s_gap[i] = 1.0 / pow(1.0 - pgap, (double)(i - 2) / 3.0);
        if (threadIdx.x == 0) s_g100 = pow(1.0 - pgap, 100.0);
        __syncthreads();
    }
    auto gap = [&](int length, bool diff) -> double {
        if (!FILL) return 0.0;
        if (length > 300) return s_g100 + (double)length;
        return diff ? s_gap[length + 2] + 20.0 : s_gap[length + 2];
    };
    // number of nodes whose 1-based position is < p1  (= id of the first node at position >= p1)
    auto rank_lt = [&](int p1) -> int {
        if (p1 <= 1) return 0;
        if (p1 > L) return ncds;
        return node_rank(nbF, nbR, nbase, p1 - 1, false);
    };
    for (int base = (int)blockIdx.y * NT; base < V; base += (int)gridDim.y * NT) {
        const int v = base + (int)threadIdx.x;
        EdgeSink sink;
        sink.n = 0;
        sink.defer = b.defer_overlap != 0;
        if (FILL && v < V) { sink.esrc = b.esrc + meta->edge_off + in_off[v]; sink.ew = b.ew + meta->edge_off + in_off[v]; }
        if (v < V && v != SRC) {
            if (v == TGT) {
                // functions.py:449-452: every close node with L - pos <= 2000
                const int a = rank_lt(L - 2000);
                for (int w = a >> 6; w <= (ncds - 1) >> 6 && ncds > 0; w++) {
                    uint64_t m = (cF[w] | cR[w]) & range_mask(w, a, ncds - 1);
                    if (!FILL) sink.n += __popcll(m);
                    else
                        while (m) { const int u = (w << 6) + __ffsll((long long)m) - 1; m &= m - 1; emit_edge<FILL>(sink, u, gap(L - npos[u], false)); }
                }
            } else {
                const int iv = ninfo[v];
                const int t = NTYPE(iv), f = NFRAME(iv), pos = npos[v];
                const bool open = (t == 0 && f > 0) || (t == 1 && f < 0);
                if (!open) {
                    // ORF edges, functions.py:311-318
                    if (t == 1) { // forward stop: one edge per start of the group
                        const DGrp G = grp[LINK_IDX(nlink[v])];
                        if (!FILL) sink.n += G.n;
                        else for (int k = 0; k < G.n; k++) emit_edge<FILL>(sink, orf[G.orf_begin + k].node, orf[G.orf_begin + k].weight);
                    } else { // reverse start: from the group's stop node
                        const DOrf *r = &orf[LINK_IDX(nlink[v])];
                        emit_edge<FILL>(sink, FILL ? grp[r->grp].node : 0, FILL ? r->weight : 0.0);
                    }
                } else {
                    if (pos <= 2000) emit_edge<FILL>(sink, SRC, gap(pos, false)); // functions.py:445-448
                    // ---- v as right node: gap edges from every close node l with 0 < r-l < 500 ----
                    {
                        const int a = rank_lt(pos - 499), e = rank_lt(pos) - 1; // ids [a, e]
                        // same strand as v: forward stop -> forward start, reverse start -> reverse stop; other strand: +20
                        const uint64_t *same = t == 0 ? cF : cR, *diff = t == 0 ? cR : cF;
                        for (int w = e >> 6; e >= a && w >= (a >> 6); w--) {
                            const uint64_t rm = range_mask(w, a, e);
                            uint64_t ms = same[w] & rm, md = diff[w] & rm;
                            if (!FILL) {
                                sink.n += __popcll(ms) + __popcll(md);
                                if (t == 0) // forward start <- reverse start needs r-l > 2 (functions.py:431): look at the nearest ones only
                                    while (md) { const int j = 63 - __clzll((long long)md); md &= ~(1ull << j); if (pos - npos[(w << 6) + j] > 2) break; sink.n--; }
                            } else {
                                uint64_t m = ms | md;
                                while (m) { // descending ids = nearest first
                                    const int j = 63 - __clzll((long long)m);
                                    m &= ~(1ull << j);
                                    const int u = (w << 6) + j;
                                    const int d = pos - npos[u];
                                    const bool isd = (md >> j) & 1ull;
                                    if (isd && t == 0 && d <= 2) continue;
                                    emit_edge<FILL>(sink, u, gap(d - 3, isd));
                                }
                            }
                        }
                    }
                    // ---- v as left node: overlap edges from close nodes r with 0 < r-l < 500 that pass the other_end tests ----
                    {
                        const int my_other = nother[v];
                        const double my_o = no[v];
                        const int a = rank_lt(pos + 1), e = rank_lt(pos + 500) - 1; // ids [a, e]
                        for (int w = a >> 6; e >= a && w <= (e >> 6); w++) {
                            uint64_t m = (cF[w] | cR[w]) & range_mask(w, a, e);
                            while (m) {
                                const int j = __ffsll((long long)m) - 1;
                                m &= m - 1;
                                const int u = (w << 6) + j;
                                const int r = npos[u], rf = NFRAME(ninfo[u]), r_other = nother[u];
                                const int d = r - pos;
                                bool hit, diff;
                                if (t == 1) { // v = reverse stop (l), lf < 0
                                    if (rf < 0) { hit = f != rf && r < my_other && r_other < pos; diff = false; }  // right is a reverse start
                                    else { hit = r_other + 3 < pos && r < my_other; diff = true; }                 // right is a forward stop
                                } else { // v = forward start (l), lf > 0
                                    if (rf > 0) { hit = f != rf && r < my_other && r_other < pos; diff = false; }  // right is a forward stop
                                    else { hit = r_other < pos && r < my_other; diff = true; }                     // right is a reverse start
                                }
                                if (hit) emit_overlap<FILL>(sink, u, d + 3, diff, (my_o + no[u]) / 2.0); // ave([o1,o2]), functions.py:385
                            }
                        }
                    }
                    // long non-coding bridges, functions.py:334-354 (v as right node)
                    for (int k = 0; k < nbr && k < PHX_MAX_BRIDGE; k++) {
                        const int last = meta->bridge[k].last, bs = meta->bridge[k].base;
                        if (!(bs - 1 <= pos && pos < bs + 500)) continue;
                        // left nodes with last-500 < l <= last+1
                        for (int u = rank_lt(last - 499); u < ncds && npos[u] <= last + 1; u++) {
                            const int lt = NTYPE(ninfo[u]), lf = NFRAME(ninfo[u]);
                            const int d = pos - npos[u];
                            bool hit = false, diff = false;
                            if (t == 0) { // v forward start
                                if (lt == 1 && lf > 0) hit = true;
                                else if (lt == 0 && lf < 0) { hit = true; diff = true; }
                            } else { // v reverse stop
                                if (lt == 0 && lf < 0) hit = true;
                                else if (lt == 1 && lf > 0) { hit = true; diff = true; }
                            }
                            if (hit) {
                                if (d < 500) parallel = true; // the connect loop adds the same edge again: ValueError graphs.py:74
                                emit_edge<FILL>(sink, u, gap(d - 3, diff));
                            }
                        }
                    }
                }
            }
        }
        if (!FILL && v < V) in_off[v] = (uint32_t)sink.n; // in-degree; k_edges_scan turns it into an offset
    }
    if (parallel) atomicMin(&meta->status, PHX_S_PARALLEL);
}

// dense pass over the batch's edge arrays: finish the overlap weights recorded by k_edges<true>
__global__ __launch_bounds__(256) void k_edge_weights(uint32_t *__restrict__ esrc, double *__restrict__ ew, const DTotals *__restrict__ tot) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tot->overflow || e >= tot->edge) return;
    const uint32_t sv = esrc[e];
    if (sv & EDGE_PENDING) {
        const int length = (int)((sv >> 21) & 511u);
        esrc[e] = sv & 0x1fffffu;
        ew[e] = score_overlap(length, (sv & 0x40000000u) != 0, ew[e]);
    }
}

// in-degrees -> exclusive offsets (CSR by destination), one workgroup per contig
__global__ __launch_bounds__(NT) void k_edges_scan(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    __shared__ uint32_t s_scan[NT / 64 + 1];
    DMeta *meta = &b.meta[blockIdx.x];
    if (meta->status < 0 || meta->n_node <= 0) {
        if (threadIdx.x == 0) meta->n_edge = 0;
        return;
    }
    const int V = meta->n_node;
    uint32_t *in_off = b.in_off + meta->node_off + blockIdx.x;
    const int per = (V + NT - 1) / NT;
    const int a = (int)threadIdx.x * per, e = a + per < V ? a + per : V;
    uint32_t sum = 0;
    for (int v = a; v < e; v++) sum += in_off[v];
    uint32_t tot;
    uint32_t ex = block_excl_scan<NT>(sum, s_scan, &tot);
    for (int v = a; v < e; v++) { const uint32_t d = in_off[v]; in_off[v] = ex; ex += d; }
    if (threadIdx.x == 0) { in_off[V] = tot; meta->n_edge = (int)tot; }
}

// ------------------------------------------------------------------------------------------------
// Exact integers for the path sums: NL little-endian 64-bit limbs, two's complement.
template <int NL>
struct WInt {
    uint64_t v[NL];
};
#define WINF_TOP 0x7fffffffffffffffull
template <int NL>
__device__ __forceinline__ WInt<NL> wi_inf() {
    WInt<NL> r;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) r.v[i] = 0;
    r.v[NL - 1] = WINF_TOP;
    return r;
}
template <int NL>
__device__ __forceinline__ bool wi_is_inf(const WInt<NL> &a) { return a.v[NL - 1] == WINF_TOP; }
// x is integer-valued (result of trunc()); |x| < 2^(64*NL-2) is guaranteed by the caller's choice of NL
template <int NL>
__device__ __forceinline__ WInt<NL> wi_from_double(double x) {
    WInt<NL> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = 0;
    const uint64_t bits = (uint64_t)__double_as_longlong(x);
    const int ef = (int)((bits >> 52) & 0x7ff);
    if (ef != 0) {
        uint64_t m = (bits & 0xfffffffffffffull) | (1ull << 52);
        const int e = ef - 1075; // value = m * 2^e
        if (e <= 0) {
            if (e > -64) r.v[0] = m >> (-e);
        } else {
            const int w = e >> 6, s = e & 63;
#pragma unroll
            for (int i = 0; i < NL; i++) {
                if (i == w) r.v[i] |= m << s;
                if (i == w + 1 && s) r.v[i] |= m >> (64 - s);
            }
        }
        if (bits >> 63) { // negate
            uint64_t c = 1;
#pragma unroll
            for (int i = 0; i < NL; i++) { uint64_t t = ~r.v[i] + c; c = (c && t == 0) ? 1 : 0; r.v[i] = t; }
        }
    }
    return r;
}
template <int NL>
__device__ __forceinline__ WInt<NL> wi_add(const WInt<NL> &a, const WInt<NL> &b) {
    WInt<NL> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint64_t s = a.v[i] + b.v[i];
        uint64_t c1 = s < a.v[i];
        uint64_t s2 = s + c;
        uint64_t c2 = s2 < s;
        r.v[i] = s2;
        c = c1 | c2;
    }
    return r;
}
template <int NL>
__device__ __forceinline__ bool wi_lt(const WInt<NL> &a, const WInt<NL> &b) { // signed a < b
    if (a.v[NL - 1] != b.v[NL - 1]) return (int64_t)a.v[NL - 1] < (int64_t)b.v[NL - 1];
#pragma unroll
    for (int i = NL - 2; i >= 0; i--)
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
    return false;
}
template <int NL>
__device__ __forceinline__ bool wi_eq(const WInt<NL> &a, const WInt<NL> &b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < NL; i++) e = e && a.v[i] == b.v[i];
    return e;
}
template <int NL>
__device__ __forceinline__ WInt<NL> wi_load(const uint64_t *p) {
    WInt<NL> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = p[i];
    return r;
}
template <int NL>
__device__ __forceinline__ void wi_store(uint64_t *p, const WInt<NL> &a) {
#pragma unroll
    for (int i = 0; i < NL; i++) p[i] = a.v[i];
}

// Shortest path source -> target with exact integer weights trunc(w*1000) (edges.py:22; fastpathz keeps
// the integer part).  The graph is not a DAG: nodes are relaxed in position order, window by
// window, each window iterated (Jacobi inside the window, so the result is schedule-independent) until it
// is stable, and the sweep over windows is repeated until a whole sweep changes no distance.  The fixed
// point of min-plus relaxation is the unique exact distance vector.  Ties between equal-length paths are
// broken canonically: parent = the tight in-edge with the lowest in-edge index (every node re-evaluates
// all its in-edges in the last, change-free sweep), see DESIGN.md.
#define PE_NONE 0xffffffffu

// path -> genes (phanotate.py:65-76, locus.py:29-37); pedge[v] = in-edge index (contig-relative) of v's parent
__device__ void emit_path_and_genes(const DBatch &b, DMeta *meta, const uint32_t *pedge, bool target_reached) {
    meta->n_genes = 0; meta->n_path = 0; meta->gene_off = 0;
    const int V = meta->n_node;
    const int SRC = V - 2, TGT = V - 1;
    const uint32_t *esrc = b.esrc + meta->edge_off;
    const int32_t *npos = b.npos + meta->node_off, *ninfo = b.ninfo + meta->node_off;
    const uint32_t *nlink = b.nlink + meta->node_off;
    const DOrf *orf = b.orf + meta->orf_off;
    const DGrp *grp = b.grp + meta->grp_off;
    int32_t *path = b.path + meta->node_off;
    if (!target_reached) { meta->status = PHX_S_NOPATH; return; }
    int n = 0;
    for (int v = TGT; v != SRC && n <= V; v = (int)esrc[pedge[v]]) n++;
    if (n > V) { meta->status = PHX_S_NEGCYCLE; return; }
    {
        int k = n;
        for (int v = TGT;; v = (int)esrc[pedge[v]]) { path[k--] = v; if (v == SRC || k < 0) break; }
    }
    meta->n_path = n + 1;
    const int npairs = n / 2; // shortest_path[1:] taken two at a time (file_handling.pairwise)
    const uint32_t g0 = atomicAdd(b.gene_total, (uint32_t)npairs);
    meta->gene_off = g0;
    meta->n_genes = npairs;
    for (int i = 0; i < npairs; i++) {
        const int a = path[2 * i + 1], bb = path[2 * i + 2];
        DGene g;
        g.left = npos[a];
        g.right = npos[bb] + 2; // locus.py:30
        g.frame = NFRAME(ninfo[a]);
        g.strand = g.frame < 0 ? -1 : 1;
        double w = 0.0; // Graph.weight, graphs.py:91-96
        const int ta = NTYPE(ninfo[a]);
        if (ta == 0 && g.frame > 0 && LINK_KIND(nlink[a]) == LINK_START) {
            const DOrf *r = &orf[LINK_IDX(nlink[a])];
            if (grp[r->grp].node == bb) w = r->weight;
        } else if (ta == 1 && g.frame < 0 && LINK_KIND(nlink[bb]) == LINK_START) {
            const DOrf *r = &orf[LINK_IDX(nlink[bb])];
            if (grp[r->grp].node == a) w = r->weight;
        }
        g.score = w;
        b.genes[g0 + i] = g;
    }
}

// ---- general kernel: distances in global memory (any V); also serves phx_solve ----
template <int NL>
__global__ __launch_bounds__(NT) void k_sssp(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    __shared__ int s_flag[2];
    DMeta *meta = &b.meta[blockIdx.x];
    const int V = meta->n_node;
    if (meta->status < 0 || V <= 2 || meta->sssp_nl != NL || meta->sssp_mode != 0) return;
    const int SRC = V - 2;
    const uint32_t *in_off = b.in_off + meta->node_off + blockIdx.x;
    const uint32_t *esrc = b.esrc + meta->edge_off;
    const double *ew = b.ew + meta->edge_off;
    const uint64_t *ewl = b.ewl ? b.ewl + (size_t)meta->edge_off * NL : nullptr; // phx_solve: integer weights given as limbs
    uint64_t *dist = b.dist + (size_t)meta->node_off * b.dist_stride;
    uint32_t *pedge = (uint32_t *)(b.parent + meta->node_off);
    const int tid = threadIdx.x;
    for (int v = tid; v < V; v += NT) {
        WInt<NL> d = wi_inf<NL>();
        if (v == SRC) {
#pragma unroll
            for (int i = 0; i < NL; i++) d.v[i] = 0;
        }
        wi_store<NL>(dist + (size_t)v * NL, d);
        pedge[v] = PE_NONE;
    }
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    __syncthreads();
    const int nchunk = (V + NT - 1) / NT;
    int sweeps = 0, it = 0;
    bool any = true, bad = false;
    while (any && !bad) {
        any = false;
        for (int c = 0; c < nchunk; c++) {
            const int v = c * NT + tid;
            int inner = 0;
            bool chg = true;
            while (chg) {
                bool improved = false, moved = false;
                WInt<NL> best;
                uint32_t be = PE_NONE;
                if (v < V && v != SRC) {
                    best = wi_load<NL>(dist + (size_t)v * NL);
                    be = pedge[v];
                    const uint32_t be0 = be;
                    const uint32_t e0 = in_off[v], e1 = in_off[v + 1];
                    for (uint32_t e = e0; e < e1; e++) {
                        const uint32_t u = esrc[e];
                        const WInt<NL> du = wi_load<NL>(dist + (size_t)u * NL);
                        if (wi_is_inf<NL>(du)) continue;
                        const WInt<NL> w = ewl ? wi_load<NL>(ewl + (size_t)e * NL) : wi_from_double<NL>(trunc(ew[e] * 1000.0));
                        const WInt<NL> cand = wi_add<NL>(du, w);
                        if (wi_lt<NL>(cand, best)) { best = cand; be = e; improved = true; }
                        else if (e < be && wi_eq<NL>(cand, best)) be = e;
                    }
                    moved = be != be0;
                }
                __syncthreads(); // every read of this iteration is done
                if (improved || moved) {
                    if (improved) { wi_store<NL>(dist + (size_t)v * NL, best); s_flag[it & 1] = 1; }
                    pedge[v] = be;
                }
                if (tid == 0) s_flag[(it + 1) & 1] = 0;
                __syncthreads();
                chg = s_flag[it & 1] != 0;
                it++;
                any = any || chg;
                if (++inner > NT + 8) { bad = true; break; } // a chunk of NT nodes converges in <= NT rounds unless a cycle is negative
            }
            if (bad) break;
        }
        if (++sweeps > V + 2) bad = true;
    }
    if (tid == 0) {
        meta->sweeps = sweeps;
        meta->sssp_iters = it;
        if (bad) meta->status = PHX_S_NEGCYCLE;
    }
}

template <int NL>
__global__ void k_path(DBatch b) {
    if (b.tot->overflow) return; // a buffer of this run is too small: the host grows it and runs again
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= b.n_contig) return;
    DMeta *meta = &b.meta[c];
    const int V = meta->n_node;
    if (meta->sssp_nl != NL || meta->sssp_mode != 0) return;
    meta->n_genes = 0; meta->n_path = 0; meta->gene_off = 0;
    if (meta->status < 0 || V <= 2) return; // phanotate.py:63: len(graph) > 2
    const uint64_t *dist = b.dist + (size_t)meta->node_off * b.dist_stride;
    emit_path_and_genes(b, meta, (const uint32_t *)(b.parent + meta->node_off), dist[(size_t)(V - 1) * NL + NL - 1] != WINF_TOP);
}

// ---- fast kernel: distances and the current window's in-edges live in LDS ----
// 1024 threads = 64 nodes x 16 lanes (one DPP row per node).  A window is 32 new nodes plus the nodes of
// the next 500 bp (every connector edge that points backwards spans < 500 bp, functions.py:372), so that
// after the window has converged its first 32 nodes are final in all but pathological cases; the outer
// sweep loop keeps the result exact regardless.  The relaxation tracks distances only; parents are chosen
// once at the end (lowest-index tight in-edge), which is also what makes ties schedule-independent.
#define SW_ADV 32
#define SW_MAX 64
#ifndef SW_ECAP
#define SW_ECAP 1024
#endif
#ifndef SW_LPN
#define SW_LPN 8 // lanes per node (4, 8 or 16; a DPP row has 16 lanes)
#endif
#define SW_THREADS (SW_MAX * SW_LPN)
//#define SW_PROFILE 1

// "infinity" that survives one addition of any edge weight without wrapping: 2^(64*NL-2).
// Real distances stay below 2^(64*NL-3) in magnitude (the host picks NL that way).
#define WBIG_TOP 0x4000000000000000ull
template <int NL>
__device__ __forceinline__ bool wi_unreached(const WInt<NL> &a) { return (int64_t)a.v[NL - 1] >= (int64_t)0x2000000000000000ull; }
// branch-free signed a < b
template <int NL>
__device__ __forceinline__ bool wi_lt_bf(const WInt<NL> &a, const WInt<NL> &b) {
    bool lt = a.v[0] < b.v[0];
#pragma unroll
    for (int i = 1; i < NL - 1; i++) lt = (a.v[i] < b.v[i]) | ((a.v[i] == b.v[i]) & lt);
    if (NL > 1) lt = ((int64_t)a.v[NL - 1] < (int64_t)b.v[NL - 1]) | ((a.v[NL - 1] == b.v[NL - 1]) & lt);
    return lt;
}
template <int NL>
__device__ __forceinline__ WInt<NL> wi_min_bf(const WInt<NL> &a, const WInt<NL> &b) {
    const bool lt = wi_lt_bf<NL>(b, a);
    WInt<NL> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = lt ? b.v[i] : a.v[i];
    return r;
}
template <int N, int NL>
__device__ __forceinline__ WInt<NL> wi_row_shr(const WInt<NL> &a) {
    WInt<NL> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t lo = dpp_row_shr<N>((uint32_t)a.v[i]), hi = dpp_row_shr<N>((uint32_t)(a.v[i] >> 32));
        r.v[i] = ((uint64_t)hi << 32) | lo;
    }
    return r;
}
// after this, the last lane of every SW_LPN-lane group holds the minimum of the group
template <int NL>
__device__ __forceinline__ WInt<NL> wi_row_min(WInt<NL> x, int sub) {
    WInt<NL> y;
    y = wi_min_bf<NL>(x, wi_row_shr<1, NL>(x)); if (sub >= 1) x = y;
    if (SW_LPN > 2) { y = wi_min_bf<NL>(x, wi_row_shr<2, NL>(x)); if (sub >= 2) x = y; }
    if (SW_LPN > 4) { y = wi_min_bf<NL>(x, wi_row_shr<4, NL>(x)); if (sub >= 4) x = y; }
    if (SW_LPN > 8) { y = wi_min_bf<NL>(x, wi_row_shr<8, NL>(x)); if (sub >= 8) x = y; }
    return x;
}
__device__ __forceinline__ uint32_t u32_row_min(uint32_t x, int sub) {
    uint32_t o;
    o = dpp_row_shr<1>(x); if (sub >= 1 && o < x) x = o;
    if (SW_LPN > 2) { o = dpp_row_shr<2>(x); if (sub >= 2 && o < x) x = o; }
    if (SW_LPN > 4) { o = dpp_row_shr<4>(x); if (sub >= 4 && o < x) x = o; }
    if (SW_LPN > 8) { o = dpp_row_shr<8>(x); if (sub >= 8 && o < x) x = o; }
    return x;
}

// Distances of the last SW_RING nodes (node ids are position-sorted, so this is a sliding window over the
// contig) are kept in an LDS ring; every improvement is also written through to the global array, which
// serves the rare reads outside the ring (an ORF edge longer than ~SW_RING nodes, or a backward edge from
// beyond a capped look-ahead).  LDS use is independent of the contig size.
//
// Window k starts at node 32k; its size (32 + look-ahead, <= 64) is planned once per contig.  While window k
// iterates out of LDS, the in-edge tile, the node types, the ring fill-in and the in-edge offsets of window
// k+1 are already in flight into registers, so global-memory latency stays off the critical path.
#ifndef SW_RING
#define SW_RING 1024
#endif
#define SW_EPT ((SW_ECAP + SW_THREADS - 1) / SW_THREADS) // tile edges prefetched per thread
#ifndef SW_RCA
#define SW_RCA 1 // in-edges per lane of a close node kept in registers
#endif
#ifndef SW_RCB
#define SW_RCB 1 // in-edges per lane of an open node kept in registers
#endif
#ifndef SW_WPS
#define SW_WPS 6 // wavefronts per SIMD the register allocation must allow (workgroups/CU = SW_WPS * 256 / SW_THREADS)
#endif
// fixed-size ring of distances + in-edge tile + one plan byte per window of SW_ADV nodes
__host__ __device__ inline size_t sssp_lds_bytes(int V, int nl) {
    return (size_t)(SW_RING + 1) * nl * 8 + (size_t)SW_ECAP * ((size_t)nl * 8 + 4) + (size_t)(V / SW_ADV + 1) + 64;
}
template <int NL>
__global__ __launch_bounds__(SW_THREADS, SW_WPS) void k_sssp_lds(DBatch b, int mode, int lds_given) {
    if (b.tot->overflow) return;
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_flag[2];
    __shared__ int s_np, s_nclose, s_viol;
    __shared__ uint32_t s_off[2][SW_MAX + 1];
    __shared__ uint8_t s_list[SW_MAX];
    DMeta *meta = &b.meta[blockIdx.x];
    const int V = meta->n_node;
    if (meta->status < 0 || V <= 2 || meta->sssp_nl != NL || meta->sssp_mode != mode) return;
    if (sssp_lds_bytes(V, NL) > (size_t)lds_given) return; // launched with less LDS than this contig needs: left unsolved (sweeps == 0), the host launches again
    const int tid = threadIdx.x;
    const int SRC = V - 2, TGT = V - 1, ncds = V - 2;
    const uint32_t *in_off = b.in_off + meta->node_off + blockIdx.x;
    const uint32_t *esrc = b.esrc + meta->edge_off;
    const double *ew = b.ew + meta->edge_off;
    const int32_t *npos = b.npos + meta->node_off;
    const int32_t *ninfo = b.ninfo + meta->node_off;
    uint64_t *gdist = b.dist + (size_t)meta->node_off * b.dist_stride;
    // LDS carve: ring (SW_RING+1)*NL u64 (slot SW_RING = the constant 0 of the source) | tile weights SW_ECAP*NL u64 |
    // tile source slots SW_ECAP u32 | window plan nW bytes.  After convergence the ring+tile area is reused for one
    // parent per node.
    uint64_t *ring = (uint64_t *)smem;
    uint64_t *tw = ring + (size_t)(SW_RING + 1) * NL;
    uint32_t *tsrc = (uint32_t *)(tw + (size_t)SW_ECAP * NL);
    uint8_t *plan = (uint8_t *)(tsrc + SW_ECAP);
    const size_t lds_words = (size_t)(SW_RING + 1) * NL * 2 + (size_t)SW_ECAP * NL * 2 + SW_ECAP; // 32-bit words before the plan
    const int nW = (V + SW_ADV - 1) / SW_ADV;
    for (int v = tid; v < V; v += SW_THREADS) {
        WInt<NL> d;
#pragma unroll
        for (int i = 0; i < NL; i++) d.v[i] = 0;
        if (v != SRC) d.v[NL - 1] = WBIG_TOP;
        wi_store<NL>(gdist + (size_t)v * NL, d);
    }
    if (tid < NL) ring[(size_t)SW_RING * NL + tid] = 0;
#ifdef SW_CENSUS
    if (tid == 0) { uint32_t c = atomicAdd(b.gene_total + 1, 1u) + 1; atomicMax(b.gene_total + 2, c); }
#endif
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    // ---- plan: size of every window = SW_ADV nodes to advance by + the nodes of the next 500 bp (<= SW_MAX, tile cap) ----
    for (int k = tid >> 6; k < nW; k += SW_THREADS / 64) {
        const int v0 = k * SW_ADV, lane = tid & 63;
        const int vadv = v0 + SW_ADV < V ? v0 + SW_ADV : V;
        const int idx = v0 + lane;
        bool ok = idx < V;
        if (ok && idx >= vadv) ok = idx < ncds && vadv - 1 < ncds && npos[idx] < npos[vadv - 1] + 500 && in_off[idx + 1] - in_off[v0] <= SW_ECAP;
        const uint64_t m = __ballot(ok);
        const int cnt = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;
        if (lane == 0) plan[k] = (uint8_t)cnt; // >= vadv - v0 >= 1
    }
    __syncthreads();
    const int node_l = tid / SW_LPN, sub = tid % SW_LPN;
#ifdef SW_PROFILE
    long long t_setup = 0, t_iter = 0, t_mark = wall_clock64();
#endif
    int sweeps = 0, it = 0;
    bool again = true, bad = false;
    uint32_t *gpe = (uint32_t *)(b.parent + meta->node_off);
    const bool ps_lds = (size_t)V <= lds_words; // else (very large contigs) the final walk chases parent edges in global memory
    uint32_t *psrc = (uint32_t *)smem;
    while (again && !bad) {
        int loaded = 0; // nodes [max(0, loaded - SW_RING), loaded) are in the ring
        if (tid < NL) ring[(size_t)SW_RING * NL + tid] = 0; // the constant-zero slot (the LDS is reused by the pass below)
        if (tid == 0) s_viol = 0;
        __syncthreads();
        // registers that carry window k+1's data while window k iterates
        uint32_t r_src[SW_EPT];
        double r_w[SW_EPT];
        uint32_t r_offn = 0;
        int r_type = 0;
        WInt<NL> r_ring;
        // prologue: window 0
        {
            const int nw0 = plan[0];
            if (tid <= nw0) s_off[0][tid] = in_off[tid];
            __syncthreads();
            const uint32_t e0n = s_off[0][0];
            const int nen = (int)(s_off[0][nw0] - e0n);
            r_type = tid < nw0 ? ninfo[tid] : 0;
            r_ring = wi_load<NL>(gdist + (size_t)(tid < nw0 ? tid : 0) * NL);
            r_offn = (nW > 1 && tid <= plan[1]) ? in_off[SW_ADV + tid] : 0u;
#pragma unroll
            for (int j = 0; j < SW_EPT; j++) {
                const int i = tid + j * SW_THREADS;
                const bool on = nen <= SW_ECAP && i < nen;
                r_src[j] = on ? esrc[e0n + i] : 0u;
                r_w[j] = on ? ew[e0n + i] : 0.0;
            }
        }
        for (int k = 0; k < nW && !bad; k++) {
            const int cur = k & 1;
            const int v0 = k * SW_ADV;
            const int nwin = plan[k];
            const int v1 = v0 + nwin;
            // ---- commit the prefetched registers of this window to LDS ----
            if (k + 1 < nW && tid <= plan[k + 1]) s_off[cur ^ 1][tid] = r_offn;
            if (tid < 64) { // split the window into close nodes (ORF-edge targets) and open nodes (connector targets)
                const int t = NTYPE(r_type), f = NFRAME(r_type);
                const bool isclose = tid < nwin && ((t == 1 && f > 0) || (t == 0 && f < 0));
                const uint64_t mc = __ballot(isclose);
                const uint64_t mo = __ballot(tid < nwin && !isclose);
                const uint64_t below = tid ? (~0ull >> (64 - tid)) : 0ull;
                const int nc = __popcll(mc);
                if (tid < nwin) s_list[isclose ? __popcll(mc & below) : nc + __popcll(mo & below)] = (uint8_t)tid;
                if (tid == 0) s_nclose = nc;
            }
            // nodes that enter the ring with this window bring their current distance from global memory
            if (loaded + tid < v1) wi_store<NL>(ring + (size_t)((loaded + tid) & (SW_RING - 1)) * NL, r_ring);
            loaded = v1 > loaded ? v1 : loaded;
            const uint32_t e0 = s_off[cur][0];
            const int ne = (int)(s_off[cur][nwin] - e0);
            const bool tiled = ne <= SW_ECAP; // false only if the SW_ADV advance nodes alone exceed the tile
            if (tiled) {
#pragma unroll
                for (int j = 0; j < SW_EPT; j++) {
                    const int i = tid + j * SW_THREADS;
                    if (i < ne) {
                        const uint32_t u = r_src[j];
                        WInt<NL> w = wi_from_double<NL>(trunc(r_w[j] * 1000.0));
                        // source slot: a ring slot, or the constant-zero slot (the source node; and sources outside the
                        // ring, whose distance cannot change while this window iterates and is folded into the weight)
                        uint32_t sl = SW_RING;
                        if (u != (uint32_t)SRC) {
                            if ((int)u < loaded && (int)u + SW_RING >= loaded) sl = u & (SW_RING - 1);
                            else w = wi_add<NL>(w, wi_load<NL>(gdist + (size_t)u * NL));
                        }
                        tsrc[i] = sl;
                        wi_store<NL>(tw + (size_t)i * NL, w);
                    }
                }
            }
            __syncthreads();
            // ---- put window k+1 in flight ----
            if (k + 1 < nW) {
                const int v0n = v0 + SW_ADV, nwn = plan[k + 1], v1n = v0n + nwn;
                const uint32_t e0n = s_off[cur ^ 1][0];
                const int nen = (int)(s_off[cur ^ 1][nwn] - e0n);
                r_type = tid < nwn ? ninfo[v0n + tid] : 0;
                r_ring = wi_load<NL>(gdist + (size_t)(loaded + tid < v1n ? loaded + tid : 0) * NL);
                r_offn = (k + 2 < nW && tid <= plan[k + 2]) ? in_off[v0n + SW_ADV + tid] : 0u;
#pragma unroll
                for (int j = 0; j < SW_EPT; j++) {
                    const int i = tid + j * SW_THREADS;
                    const bool on = nen <= SW_ECAP && i < nen;
                    r_src[j] = on ? esrc[e0n + i] : 0u;
                    r_w[j] = on ? ew[e0n + i] : 0.0;
                }
            }
#ifdef SW_PROFILE
            { long long t = wall_clock64(); t_setup += t - t_mark; t_mark = t; }
#endif
            // ---- iterate the window to its fixed point ----
            // The graph is bipartite: close nodes (forward stops, reverse starts) are reached by ORF edges from open
            // nodes only; open nodes (forward starts, reverse stops, target) by connector edges from close nodes and the
            // source only.  One round = phase A (all close nodes) then phase B (all open nodes): inside a phase nobody
            // reads what anybody writes, so results are written at once, and a round advances two hops.
            const int nclose = s_nclose, nopen = nwin - nclose;
            const bool actA = node_l < nclose, actB = node_l < nopen;
            const int lA = actA ? s_list[node_l] : 0, lB = actB ? s_list[nclose + node_l] : 0;
            const int iaA = actA ? (int)(s_off[cur][lA] - e0) + sub : 0, ibA = actA ? (int)(s_off[cur][lA + 1] - e0) : 0;
            const int iaB = actB ? (int)(s_off[cur][lB] - e0) + sub : 0, ibB = actB ? (int)(s_off[cur][lB + 1] - e0) : 0;
            uint64_t *slotA = ring + (size_t)((v0 + lA) & (SW_RING - 1)) * NL, *slotB = ring + (size_t)((v0 + lB) & (SW_RING - 1)) * NL;
            uint64_t *gA = gdist + (size_t)(v0 + lA) * NL, *gB = gdist + (size_t)(v0 + lB) * NL;
            // the first SW_RCA / SW_RCB in-edges of this lane stay in registers for all rounds of the window
            // (close nodes have 1-2 in-edges — the starts of one stop-group; open nodes ~14 — the connectors)
            uint32_t csA[SW_RCA], csB[SW_RCB];
            WInt<NL> cwA[SW_RCA], cwB[SW_RCB];
            {
                WInt<NL> big;
#pragma unroll
                for (int i = 0; i < NL; i++) big.v[i] = 0;
                big.v[NL - 1] = WBIG_TOP;
#pragma unroll
                for (int j = 0; j < SW_RCA; j++) {
                    const int ia = iaA + j * SW_LPN;
                    const bool oa = tiled && ia < ibA;
                    csA[j] = oa ? tsrc[ia] : (uint32_t)SW_RING;
                    cwA[j] = oa ? wi_load<NL>(tw + (size_t)ia * NL) : big;
                }
#pragma unroll
                for (int j = 0; j < SW_RCB; j++) {
                    const int ib = iaB + j * SW_LPN;
                    const bool ob = tiled && ib < ibB;
                    csB[j] = ob ? tsrc[ib] : (uint32_t)SW_RING;
                    cwB[j] = ob ? wi_load<NL>(tw + (size_t)ib * NL) : big;
                }
            }
            // A phase that changes nothing ends the window: the next phase would read exactly what it read last time.
            // (Exception: the window's very first phase A — phase B has not seen this window's close nodes yet.)
            int inner = 0;
            for (int ph = 0;; ph ^= 1) {
                const bool act = ph ? actB : actA;
                // node groups are compacted per phase: a wavefront whose first group is already past the phase's node
                // count has nothing to do and goes straight to the barrier (wave-uniform branch)
                if ((tid & ~63) / SW_LPN < (ph ? nopen : nclose)) {
                    const int ia = ph ? iaB : iaA, ib = ph ? ibB : ibA;
                    uint64_t *myslot = ph ? slotB : slotA;
                    WInt<NL> d0;
#pragma unroll
                    for (int i = 0; i < NL; i++) d0.v[i] = 0;
                    d0.v[NL - 1] = WBIG_TOP;
                    if (act) d0 = wi_load<NL>(myslot);
                    WInt<NL> best = d0;
                    if (tiled) {
                        int i;
                        if (ph) {
#pragma unroll
                            for (int j = 0; j < SW_RCB; j++) best = wi_min_bf<NL>(best, wi_add<NL>(wi_load<NL>(ring + (size_t)csB[j] * NL), cwB[j]));
                            i = ia + SW_RCB * SW_LPN;
                        } else {
#pragma unroll
                            for (int j = 0; j < SW_RCA; j++) best = wi_min_bf<NL>(best, wi_add<NL>(wi_load<NL>(ring + (size_t)csA[j] * NL), cwA[j]));
                            i = ia + SW_RCA * SW_LPN;
                        }
                        // the rare rest comes from the LDS tile; the source slot of the next edge is already on its way
                        uint32_t sl = i < ib ? tsrc[i] : 0u;
                        while (i < ib) {
                            const int in = i + SW_LPN;
                            const uint32_t sn = in < ib ? tsrc[in] : 0u;
                            best = wi_min_bf<NL>(best, wi_add<NL>(wi_load<NL>(ring + (size_t)sl * NL), wi_load<NL>(tw + (size_t)i * NL)));
                            sl = sn;
                            i = in;
                        }
                    } else {
                        for (int i = ia; i < ib; i += SW_LPN) {
                            const uint32_t u = esrc[e0 + i];
                            WInt<NL> du;
                            if (u == (uint32_t)SRC) du = wi_load<NL>(ring + (size_t)SW_RING * NL);
                            else if ((int)u < loaded && (int)u + SW_RING >= loaded) du = wi_load<NL>(ring + (size_t)(u & (SW_RING - 1)) * NL);
                            else du = wi_load<NL>(gdist + (size_t)u * NL);
                            best = wi_min_bf<NL>(best, wi_add<NL>(du, wi_from_double<NL>(trunc(ew[e0 + i] * 1000.0))));
                        }
                    }
                    best = wi_row_min<NL>(best, sub);
                    if (act && sub == SW_LPN - 1 && wi_lt_bf<NL>(best, d0)) {
                        wi_store<NL>(myslot, best);
                        wi_store<NL>(ph ? gB : gA, best); // write-through
                        s_flag[it & 1] = 1;
                    }
                }
                if (tid == 0) s_flag[(it + 1) & 1] = 0;
                __syncthreads();
                const bool chg = s_flag[it & 1] != 0;
                it++;
                if (!chg && (ph == 1 || inner > 0)) break;
                if (++inner > 2 * SW_MAX + 16) { bad = true; break; }
            }
#ifdef SW_PROFILE
            { long long t = wall_clock64(); t_iter += t - t_mark; t_mark = t; }
#endif
        }
        if (++sweeps > V + 2) bad = true;
        __syncthreads();
        // ---- verification + parents in one pass over every in-edge (distances from global memory) ----
        // A node that some in-edge could still improve means the sweep missed a backward dependency that reaches
        // beyond a window's look-ahead: sweep again.  Otherwise the distances are the fixed point and every node
        // takes its tight in-edge with the lowest index as parent (canonical tie-break).
        for (int vb = 0; vb < V; vb += SW_MAX) {
            const int v = vb + node_l;
            uint32_t be = PE_NONE;
            bool viol = false;
            if (v < V) {
                const WInt<NL> dv = wi_load<NL>(gdist + (size_t)v * NL);
                const uint32_t e1 = in_off[v + 1];
                for (uint32_t e = in_off[v] + sub; e < e1; e += SW_LPN) {
                    const WInt<NL> cand = wi_add<NL>(wi_load<NL>(gdist + (size_t)esrc[e] * NL), wi_from_double<NL>(trunc(ew[e] * 1000.0)));
                    if (wi_lt_bf<NL>(cand, dv)) viol = true;
                    if (wi_eq<NL>(cand, dv) && e < be && !wi_unreached<NL>(dv)) be = e;
                }
            }
            be = u32_row_min(be, sub);
            if (viol) s_viol = 1;
            if (v < V && sub == SW_LPN - 1) { gpe[v] = be; if (ps_lds) psrc[v] = be == PE_NONE ? PE_NONE : esrc[be]; }
        }
        __syncthreads();
        again = s_viol != 0;
        __syncthreads();
    }
    // ---- path (phanotate.py:64-67) and genes (phanotate.py:71-76, locus.py:29-37) ----
    int32_t *path = b.path + meta->node_off;
    if (tid == 0) {
        meta->sweeps = sweeps;
        meta->sssp_iters = it;
        meta->n_genes = 0; meta->n_path = 0; meta->gene_off = 0;
        int np = -1;
        if (bad) meta->status = PHX_S_NEGCYCLE;
        else if (wi_unreached<NL>(wi_load<NL>(gdist + (size_t)TGT * NL))) meta->status = PHX_S_NOPATH;
        else {
            int n = 0;
            for (int v = TGT; v != SRC && n <= V; v = ps_lds ? (int)psrc[v] : (int)esrc[gpe[v]]) n++;
            if (n > V) meta->status = PHX_S_NEGCYCLE;
            else {
                int k = n;
                for (int v = TGT;; v = ps_lds ? (int)psrc[v] : (int)esrc[gpe[v]]) { path[k--] = v; if (v == SRC || k < 0) break; }
                meta->n_path = n + 1;
                np = n / 2; // shortest_path[1:] taken two at a time (file_handling.pairwise)
                meta->n_genes = np;
                meta->gene_off = atomicAdd(b.gene_total, (uint32_t)np);
            }
        }
        s_np = np;
#ifdef SW_CENSUS
        atomicSub(b.gene_total + 1, 1u);
#endif
#ifdef SW_PROFILE
        { long long t = wall_clock64(); meta->pmax[0] = (uint32_t)t_setup; meta->pmin[0] = (uint32_t)t_iter; meta->sssp_why = (int32_t)(t - t_mark); }
#endif
    }
    __syncthreads();
    const int npairs = s_np;
    if (npairs > 0) {
        const int32_t *ninfo = b.ninfo + meta->node_off;
        const uint32_t *nlink = b.nlink + meta->node_off;
        const DOrf *orf = b.orf + meta->orf_off;
        const DGrp *grp = b.grp + meta->grp_off;
        const int64_t g0 = meta->gene_off;
        for (int i = tid; i < npairs; i += SW_THREADS) {
            const int a = path[2 * i + 1], bb = path[2 * i + 2];
            DGene g;
            g.left = npos[a];
            g.right = npos[bb] + 2; // locus.py:30
            g.frame = NFRAME(ninfo[a]);
            g.strand = g.frame < 0 ? -1 : 1;
            double w = 0.0; // Graph.weight, graphs.py:91-96
            const int ta = NTYPE(ninfo[a]);
            if (ta == 0 && g.frame > 0 && LINK_KIND(nlink[a]) == LINK_START) {
                const DOrf *r = &orf[LINK_IDX(nlink[a])];
                if (grp[r->grp].node == bb) w = r->weight;
            } else if (ta == 1 && g.frame < 0 && LINK_KIND(nlink[bb]) == LINK_START) {
                const DOrf *r = &orf[LINK_IDX(nlink[bb])];
                if (grp[r->grp].node == a) w = r->weight;
            }
            g.score = w;
            b.genes[g0 + i] = g;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Batch layout on the device.  One workgroup walks the contigs (a scan per 1024), so that no host round trip is
// needed between the counting and the emitting kernels when the buffers of the context are large enough.
#define LAYOUT_T 1024
// four running sums at once
__device__ __forceinline__ void layout_scan4(int64_t v[4], int64_t base[4], int64_t *s_part /* [4][LAYOUT_T/64 + 1] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t inc[4];
    for (int q = 0; q < 4; q++) {
        int64_t x = v[q];
        for (int d = 1; d < 64; d <<= 1) { const int64_t t = __shfl_up(x, d); if (lane >= d) x += t; }
        inc[q] = x;
        if (lane == 63) s_part[q * 17 + w] = x;
    }
    __syncthreads();
    if (threadIdx.x < 4) { int64_t a = 0; for (int i = 0; i < LAYOUT_T / 64; i++) { const int64_t t = s_part[threadIdx.x * 17 + i]; s_part[threadIdx.x * 17 + i] = a; a += t; } s_part[threadIdx.x * 17 + 16] = a; }
    __syncthreads();
    for (int q = 0; q < 4; q++) { const int64_t ex = base[q] + s_part[q * 17 + w] + inc[q] - v[q]; const int64_t tot = s_part[q * 17 + 16]; v[q] = ex; inc[q] = tot; }
    __syncthreads();
    for (int q = 0; q < 4; q++) base[q] += inc[q];
}
// after k_orf<false>: ORF / group / node / close-bitmap offsets (what phx_run did on the host between two syncs)
__global__ __launch_bounds__(LAYOUT_T) void k_layout1(DBatch b) {
    __shared__ int64_t s_part[4 * 17];
    int64_t base[4] = {0, 0, 0, 0};
    for (int i0 = 0; i0 < b.n_contig; i0 += LAYOUT_T) {
        const int i = i0 + (int)threadIdx.x;
        int64_t v[4] = {0, 0, 0, 0};
        int n_node = 0, ncw = 0;
        if (i < b.n_contig) {
            DMeta *m = &b.meta[i];
            n_node = m->status < 0 ? 0 : m->n_orf + m->n_grp + 2;
            ncw = n_node / 64 + 1;
            v[0] = m->n_orf; v[1] = m->n_grp; v[2] = n_node; v[3] = 2 * (int64_t)ncw;
        }
        layout_scan4(v, base, s_part);
        if (i < b.n_contig) {
            DMeta *m = &b.meta[i];
            m->orf_off = v[0]; m->grp_off = v[1]; m->node_off = v[2]; m->cb_off = v[3];
            m->n_node = n_node; m->ncw = ncw;
        }
    }
    if (threadIdx.x == 0) {
        DTotals *t = b.tot;
        t->orf = base[0]; t->grp = base[1]; t->node = base[2]; t->cb = base[3];
        if (base[0] > b.caps.orf || base[1] > b.caps.grp || base[2] > b.caps.node || base[3] > b.caps.cb) t->overflow |= 1;
    }
}


// after k_edges<false>: edge offsets; per contig the integer width its path sums need and the kernel that solves it
__global__ __launch_bounds__(LAYOUT_T) void k_layout2(DBatch b) {
    __shared__ int64_t s_part[4 * 17];
    __shared__ int s_nl, s_mask, s_vmax;
    __shared__ unsigned long long s_lds[4];
    if (b.tot->overflow) return;
    if (threadIdx.x == 0) { s_nl = 2; s_mask = 0; s_vmax = 0; for (int k = 0; k < 4; k++) s_lds[k] = 0; }
    __syncthreads();
    int64_t base[4] = {0, 0, 0, 0};
    const bool force_global = b.caps.flags & 1, no_wave = (b.caps.flags & 2) != 0;
    for (int i0 = 0; i0 < b.n_contig; i0 += LAYOUT_T) {
        const int i = i0 + (int)threadIdx.x;
        int64_t v[4] = {0, 0, 0, 0};
        if (i < b.n_contig) {
            DMeta *m = &b.meta[i];
            m->sssp_nl = 2; m->sssp_mode = 0; m->sssp_fb = 0;
            if (m->status < 0) m->n_edge = 0;
            else {
                v[0] = m->n_edge;
                // A tentative distance is the length of a walk that uses every ORF edge at most once (a shortest path is
                // simple; longer walks never win), so |dist| <= B = sum |w_orf| + (V/2) * max |w_connector|.  The connector
                // bound follows functions.py:26-46: overlap < 500 bp, gap <= 300 bp or bridge pow(.)+length; terminals are
                // smaller still.  A candidate d(u)+w needs one more bit, the sign another, the "unreached" pattern sits two
                // bits higher; one bit covers the rounding of the fp64 sum.
                int bits = 4096;
                if (m->maxexp < 2000) {
                    const double pst = contig_pstop(m->gc, m->L);
                    double cmax = 1.0 / pow(1.0 - pst, 500.0);
                    const double c2 = 1.0 / pow(1.0 - pst, 100.0);
                    cmax = (cmax > c2 ? cmax : c2) + 20.0;
                    const double c3 = (double)m->L + 21.0;
                    cmax = (cmax > c3 ? cmax : c3) * 1000.0;
                    const double bound = m->wsum + 0.5 * (double)(m->n_node > 2 ? m->n_node : 2) * cmax;
                    int eb = 0;
                    (void)frexp(bound, &eb);
                    bits = (eb > m->maxexp ? eb : m->maxexp) + 5;
                }
                if (bits > 17 * 64) m->status = PHX_S_OVERFLOW;
                else {
                    const int k = bits <= 128 ? 0 : bits <= 256 ? 1 : bits <= 512 ? 2 : 3;
                    const int nl = k == 0 ? 2 : k == 1 ? 4 : k == 2 ? 8 : 17;
                    m->sssp_nl = nl;
                    const size_t lds = sssp_lds_bytes(m->n_node, nl);
                    // the kernel a contig falls back to when the wavefront kernel hands it back (and the one it gets otherwise)
                    const int fb = force_global ? 0 : (lds <= 158 * 1024 ? 1 : 0);
                    const int mode = (!force_global && !no_wave && nl == 2) ? 2 : fb;
                    m->sssp_fb = fb; m->sssp_mode = mode;
                    atomicMax(&s_nl, nl);
                    if (m->n_node > 2) {
                        atomicOr(&s_mask, (1 << (4 * k + mode)) | (mode == 2 ? 1 << (4 * k + fb) : 0));
                        if (fb == 1) atomicMax(&s_lds[k], (unsigned long long)lds);
                        atomicMax(&s_vmax, m->n_node);
                    }
                }
            }
        }
        layout_scan4(v, base, s_part);
        if (i < b.n_contig) b.meta[i].edge_off = v[0];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        DTotals *t = b.tot;
        t->edge = base[0]; t->nlmax = s_nl; t->class_mask = s_mask; t->vmax = s_vmax;
        for (int k = 0; k < 4; k++) t->lds_need[k] = (int64_t)s_lds[k];
        if (base[0] + 1 > b.caps.edge || s_nl > b.caps.limbs) t->overflow |= 2;
    }
}

#include "phx_sssp_wave.inc"

// ------------------------------------------------------------------------------------------------
// launchers
extern "C" {
void phxk_features(const DBatch *b, const DTile *tiles, int n_tiles, void *stream) {
    if (n_tiles > 0) hipLaunchKernelGGL(k_features, dim3(n_tiles < 2048 ? n_tiles : 2048), dim3(PHX_FEAT_THREADS), 0, (hipStream_t)stream, *b, tiles, n_tiles);
}
void phxk_orf_count(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_orf<false>, dim3(b->n_contig), dim3(NT), 0, (hipStream_t)stream, *b); }
void phxk_orf_emit(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_orf<true>, dim3(b->n_contig, 6), dim3(NT), 0, (hipStream_t)stream, *b); }
void phxk_orf_stats(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_orf_stats, dim3(b->n_contig, 8), dim3(NT), 0, (hipStream_t)stream, *b); }
void phxk_train(const DBatch *, void *) {}
void phxk_score(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_score, dim3(b->n_contig), dim3(NT), 0, (hipStream_t)stream, *b); }
void phxk_nodes(const DBatch *b, void *stream) {
    hipLaunchKernelGGL(k_node_cov, dim3(b->n_contig, 4), dim3(NT), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_node_rank, dim3(b->n_contig), dim3(NT), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_node_build, dim3(b->n_contig, 4), dim3(NT), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_node_attr, dim3(b->n_contig, 4), dim3(NT), 0, (hipStream_t)stream, *b);
}
void phxk_edges_count(const DBatch *b, void *stream) {
    hipLaunchKernelGGL(k_edges<false>, dim3(b->n_contig, 4), dim3(NT), 0, (hipStream_t)stream, *b);
    hipLaunchKernelGGL(k_edges_scan, dim3(b->n_contig), dim3(NT), 0, (hipStream_t)stream, *b);
}
void phxk_edges_fill(const DBatch *b, int64_t n_edges, void *stream) {
    hipLaunchKernelGGL(k_edges<true>, dim3(b->n_contig, 4), dim3(NT), 0, (hipStream_t)stream, *b);
    if (b->defer_overlap && n_edges > 0)
        hipLaunchKernelGGL(k_edge_weights, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b->esrc, b->ew, (const DTotals *)b->tot);
}
// phx_solve: the relaxation alone, no path/gene emission (the caller walks the parent edges)
void phxk_sssp_only(const DBatch *b, int nl, void *stream) {
    dim3 g(b->n_contig), t(NT);
    hipStream_t s = (hipStream_t)stream;
    switch (nl) {
    case 2: hipLaunchKernelGGL(k_sssp<2>, g, t, 0, s, *b); break;
    case 4: hipLaunchKernelGGL(k_sssp<4>, g, t, 0, s, *b); break;
    case 8: hipLaunchKernelGGL(k_sssp<8>, g, t, 0, s, *b); break;
    default: hipLaunchKernelGGL(k_sssp<17>, g, t, 0, s, *b); break;
    }
}

size_t phxk_sssp_lds_bytes(int V, int nl) { return sssp_lds_bytes(V, nl); }
void phxk_layout1(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_layout1, dim3(1), dim3(LAYOUT_T), 0, (hipStream_t)stream, *b); }
void phxk_layout2(const DBatch *b, void *stream) { hipLaunchKernelGGL(k_layout2, dim3(1), dim3(LAYOUT_T), 0, (hipStream_t)stream, *b); }

int phxk_sssp_wave_ok(int nl) { return nl == 2; }

// mode 0: global-memory kernel (+ k_path); mode 1: workgroup-per-contig LDS kernel with `lds_bytes` of dynamic LDS;
// mode 2: wavefront-per-contig kernel (contigs it hands back carry their fallback mode in sssp_mode afterwards)
void phxk_sssp(const DBatch *b, int nl, int mode, size_t lds_bytes, void *stream) {
    dim3 g(b->n_contig), t(NT);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 2) {
        const size_t lb = wv_lds_bytes<2>();
        (void)hipFuncSetAttribute((const void *)k_sssp_wave<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
        hipLaunchKernelGGL(k_sssp_wave<2>, g, dim3(64), lb, s, *b);
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
