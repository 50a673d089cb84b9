// phx_feat_core.h — the per-position features of PHANOTATE's scan, position-bit-sliced in residue space.
//
// What it restates (reference: /root/reference/phanotate_modules/):
//   functions.py:158-171  per-base loop: g+c count after the counting remap, background RBS bins of dna[i:i+21] and of its rev_comp
//   functions.py:48-138   score_rbs: 43 first-match rules with non-increasing scores  ==  max score over all matching (motif, offset)
//   functions.py:198-215  codon classes in elif order (forward start, reverse start, forward stop, reverse stop)
//   gc_frame_plot.py:29-74, 7-28   W(q) = g+c among q + 3m, m in [-19, 20]; max_idx / min_idx as strict comparisons
//
// Layout.  Position p = 3 k + r belongs to residue stream r = p mod 3 at index k.  A lane holds, for each of the three streams, one
// 32-bit word = 32 consecutive indices k (96 positions), as bit planes: b0, b1 (base code a0 c1 t2 g3 — an ambiguity code as the base
// the reference counts it as), amb (not one of acgt; a position outside the contig is amb with code 0, a letter outside the IUPAC
// alphabet is amb with code 2).  In this space a codon of frame f is ONE bit index of three streams, the GC window of
// gc_frame_plot is a sliding sum over 40 consecutive bits of one stream, and every output bitmap (bit k of frame f <-> codon at
// f + 3 k) falls out without a stride-3 shuffle.  A shift of a plane by j positions is a funnel shift by <= ceil(j / 3) bits
// (v_alignbit_b32) with the neighbouring lane's word (DPP wave_shl / wave_shr), or nothing at all when only the stream changes.
//
// The code is written once over a word type V: uint32_t on the device (phx_features.inc), a 64-lane array on the host
// (tests/feat_core_host.cpp: the same source checked against the oracle without a GPU).  V needs & | ^ ~ and five functions that
// must be declared in namespace phxfc BEFORE this header is included:
//   V fsr(V cur, V next, int s)   (cur >> s) | (next << (32 - s))      V lane_prev(V)   the word of the lane before this one
//   V fsl(V cur, V prev, int s)   (cur << s) | (prev >> (32 - s))      V lane_next(V)   ... after this one
//   void popc_add(V &acc, V x)    acc += popcount(x), per lane
// Not a CPU path of the product: libphx.so contains only the device instance.
#pragma once
#include <stdint.h>

#ifndef PHX_FC_INLINE
#define PHX_FC_INLINE inline
#endif

namespace phxfc {

// ---- compile-time loops ----
template <int I> struct ic { static constexpr int value = I; };
template <class F> PHX_FC_INLINE void rule_seq(F &&f) {
    f(ic<0>()); f(ic<1>()); f(ic<2>()); f(ic<3>()); f(ic<4>()); f(ic<5>()); f(ic<6>()); f(ic<7>()); f(ic<8>()); f(ic<9>()); f(ic<10>()); f(ic<11>()); f(ic<12>()); f(ic<13>());
    f(ic<14>()); f(ic<15>()); f(ic<16>()); f(ic<17>()); f(ic<18>()); f(ic<19>()); f(ic<20>()); f(ic<21>()); f(ic<22>()); f(ic<23>()); f(ic<24>()); f(ic<25>()); f(ic<26>());
}

// ---- a plane over the three residue streams, with the neighbouring lanes' words ----
template <class V>
struct P3 {
    V c[3]; // this lane's word of stream 0..2
    V p[3]; // the previous lane's (indices k - 32 .. k - 1)
    V n[3]; // the next lane's
};
template <class V>
PHX_FC_INLINE P3<V> make_p3(const V &s0, const V &s1, const V &s2) {
    P3<V> x;
    x.c[0] = s0; x.c[1] = s1; x.c[2] = s2;
    for (int r = 0; r < 3; r++) { x.p[r] = lane_prev(x.c[r]); x.n[r] = lane_next(x.c[r]); } // (the unused side is dead code)
    return x;
}
// Y[k] = X at position 3 k + R + D
template <int R, int D, class V>
PHX_FC_INLINE V at(const P3<V> &x) {
    constexpr int t = R + D;
    constexpr int rr = ((t % 3) + 3) % 3;
    constexpr int s = (t - rr) / 3;
    static_assert(s > -32 && s < 32, "one neighbouring word is all a shift may reach");
    if constexpr (s == 0) return x.c[rr];
    else if constexpr (s > 0) return fsr(x.c[rr], x.n[rr], s);
    else return fsl(x.c[rr], x.p[rr], -s);
}

// ---- score_rbs ----
// Motif groups (every motif of a group has the same score in every offset class), anchored at the motif's first symbol x, read in the
// direction SG (forward windows read the strand leftwards: s[k] = dna[e - k], SG = -1; reverse windows s[k] = comp(dna[i + k]),
// SG = +1 on the complemented base planes):
enum { M_P1 /* ggagga */, M_P2 /* ggagg */, M_P3 /* gagga */, M_P4 /* gga[act]ga | gg[cgt]gga */, M_P4A /* gga[act]ga */, M_P5A /* ggag | gagg */,
       M_P5B /* agga */, M_P6 /* gg[cgt]gg */, M_P7 /* agg | gag | gga */, M_P8 /* ga[act]ga */, M_N };
template <int R, int SG, class V>
PHX_FC_INLINE void motif_planes(const P3<V> &G, const P3<V> &A, const P3<V> &nA, const P3<V> &nG, V m[M_N]) {
    const V g0 = at<R, 0>(G), g1 = at<R, SG * 1>(G), g2 = at<R, SG * 2>(G), g3 = at<R, SG * 3>(G), g4 = at<R, SG * 4>(G);
    const V a0 = at<R, 0>(A), a1 = at<R, SG * 1>(A), a2 = at<R, SG * 2>(A), a3 = at<R, SG * 3>(A), a4 = at<R, SG * 4>(A), a5 = at<R, SG * 5>(A);
    const V na2 = at<R, SG * 2>(nA), ng2 = at<R, SG * 2>(nG), ng3 = at<R, SG * 3>(nG);
    const V gga = g0 & g1 & a2, ggag = gga & g3;
    const V gag = g0 & a1 & g2, gagg = gag & g3;
    const V agg = a0 & g1 & g2;
    m[M_P2] = ggag & g4;
    m[M_P1] = m[M_P2] & a5;
    m[M_P3] = gagg & a4;
    m[M_P5A] = ggag | gagg;
    m[M_P5B] = agg & a3;
    m[M_P7] = agg | gag | gga;
    m[M_P6] = (g0 & g1 & na2) & g3 & g4;
    m[M_P8] = (g0 & a1 & ng2) & g3 & a4;
    m[M_P4A] = (gga & ng3) & g4 & a5;
    m[M_P4] = m[M_P4A] | (m[M_P6] & a5);
}

// The 27 scores in descending order: (score, motif group(s), offset class).  Offset classes: 0 = 3-4, 1 = 5-10, 2 = 11-12, 3 = 13-15.
// Source planes (S_*): the groups above and three unions.
enum { S_P1, S_P2, S_P3, S_P23, S_P4, S_P5A, S_P5B, S_P5, S_P6, S_P7, S_P8, S_X2 /* P7 | P4A | P6 */, S_N };
struct RbsRule { int score, src, cls; };
static constexpr RbsRule kRbsRules[27] = {
    {27, S_P1, 1}, {26, S_P1, 0}, {25, S_P1, 2}, {24, S_P2, 1}, {23, S_P2, 0}, {22, S_P3, 1}, {21, S_P3, 0}, {20, S_P23, 2}, {19, S_P4, 1},
    {18, S_P4, 0}, {17, S_P4, 2}, {16, S_P5A, 1}, {15, S_P5B, 1}, {14, S_P6, 1}, {13, S_P7, 1}, {12, S_P5, 2}, {11, S_P5, 0}, {10, S_P23, 3},
    {9, S_P8, 1}, {8, S_P6, 0}, {7, S_P6, 2}, {6, S_P7, 2}, {5, S_P8, 0}, {4, S_P8, 2}, {3, S_P5, 3}, {2, S_X2, 3}, {1, S_P7, 0}};

template <class V>
struct RbsSrc {
    P3<V> one[S_N]; // the group plane, anchored at x
    P3<V> two[S_N]; // | the same one position further along the reading direction: two consecutive offsets
};
// hits of source plane `src` in offset class `cls` for the windows of output stream R: OR over the class's offsets o of the plane at
// e + SG * o
template <int R, int SG, int CLS, class V>
PHX_FC_INLINE V class_hits(const P3<V> &one, const P3<V> &two) {
    if constexpr (CLS == 0) return at<R, SG * 3>(two);                                         // 3, 4
    else if constexpr (CLS == 1) return at<R, SG * 5>(two) | at<R, SG * 7>(two) | at<R, SG * 9>(two); // 5 .. 10
    else if constexpr (CLS == 2) return at<R, SG * 11>(two);                                   // 11, 12
    else return at<R, SG * 13>(two) | at<R, SG * 15>(one);                                     // 13, 14, 15
}

// Background bins of one strand: cnt[s] += number of valid windows of the lane whose bin is s (s = 0 .. 27); TAP: the bin of every
// window as five bit planes per output stream.
template <int SG, bool TAP, class V>
PHX_FC_INLINE void rbs_strand(const P3<V> &G, const P3<V> &A, const P3<V> &nA, const P3<V> &nG, const V valid[3], V cnt[28], V (*bin)[5]) {
    V m[3][M_N];
    motif_planes<0, SG>(G, A, nA, nG, m[0]);
    motif_planes<1, SG>(G, A, nA, nG, m[1]);
    motif_planes<2, SG>(G, A, nA, nG, m[2]);
    RbsSrc<V> src;
    auto put = [&](int id, const V &s0, const V &s1, const V &s2) {
        src.one[id] = make_p3(s0, s1, s2);
        const P3<V> &o = src.one[id];
        src.two[id] = make_p3(o.c[0] | at<0, SG>(o), o.c[1] | at<1, SG>(o), o.c[2] | at<2, SG>(o));
    };
    put(S_P1, m[0][M_P1], m[1][M_P1], m[2][M_P1]);
    put(S_P2, m[0][M_P2], m[1][M_P2], m[2][M_P2]);
    put(S_P3, m[0][M_P3], m[1][M_P3], m[2][M_P3]);
    put(S_P23, m[0][M_P2] | m[0][M_P3], m[1][M_P2] | m[1][M_P3], m[2][M_P2] | m[2][M_P3]);
    put(S_P4, m[0][M_P4], m[1][M_P4], m[2][M_P4]);
    put(S_P5A, m[0][M_P5A], m[1][M_P5A], m[2][M_P5A]);
    put(S_P5B, m[0][M_P5B], m[1][M_P5B], m[2][M_P5B]);
    put(S_P5, m[0][M_P5A] | m[0][M_P5B], m[1][M_P5A] | m[1][M_P5B], m[2][M_P5A] | m[2][M_P5B]);
    put(S_P6, m[0][M_P6], m[1][M_P6], m[2][M_P6]);
    put(S_P7, m[0][M_P7], m[1][M_P7], m[2][M_P7]);
    put(S_P8, m[0][M_P8], m[1][M_P8], m[2][M_P8]);
    put(S_X2, m[0][M_P7] | m[0][M_P4A] | m[0][M_P6], m[1][M_P7] | m[1][M_P4A] | m[1][M_P6], m[2][M_P7] | m[2][M_P4A] | m[2][M_P6]);

    auto stream = [&](auto RC) {
        constexpr int R = decltype(RC)::value;
        V found = ~valid[R]; // a window that is no background window never counts
        V b[5];
        if (TAP) for (int i = 0; i < 5; i++) b[i] = found & ~found; // 0
        auto rule = [&](auto IC) {
            constexpr int I = decltype(IC)::value;
            constexpr RbsRule ru = kRbsRules[I];
            const V s = class_hits<R, SG, ru.cls>(src.one[ru.src], src.two[ru.src]);
            const V h = s & ~found; // first match == highest score: the chain is in descending order
            popc_add(cnt[ru.score], h);
            if (TAP) for (int i = 0; i < 5; i++) if ((ru.score >> i) & 1) b[i] = b[i] | h;
            found = found | s;
        };
        rule_seq(rule);
        popc_add(cnt[0], ~found);
        if (TAP) for (int i = 0; i < 5; i++) bin[R][i] = b[i];
    };
    stream(ic<0>()); stream(ic<1>()); stream(ic<2>());
}

// ---- GC frame plot: W as six bit planes per stream, then the strict comparisons ----
template <int N, class V>
struct Num { V b[N]; }; // bit-sliced unsigned numbers, b[0] = least significant
// a (NA bits) + b (NB bits) -> NO bits (the caller knows the sum fits)
template <int NO, int NA, int NB, class V>
PHX_FC_INLINE Num<NO, V> add(const Num<NA, V> &a, const Num<NB, V> &b) {
    Num<NO, V> r;
    V carry = a.b[0] & b.b[0];
    r.b[0] = a.b[0] ^ b.b[0];
    for (int i = 1; i < NO; i++) {
        const bool ha = i < NA, hb = i < NB;
        if (ha && hb) { r.b[i] = a.b[i] ^ b.b[i] ^ carry; carry = (a.b[i] & b.b[i]) | (carry & (a.b[i] | b.b[i])); }
        else if (ha) { r.b[i] = a.b[i] ^ carry; carry = a.b[i] & carry; }
        else if (hb) { r.b[i] = b.b[i] ^ carry; carry = b.b[i] & carry; }
        else { r.b[i] = carry; carry = carry & ~carry; }
    }
    return r;
}
template <int S, int N, class V>
PHX_FC_INLINE Num<N, V> shifted(const Num<N, V> &a) { // r[k] = a[k + S]
    Num<N, V> r;
    for (int i = 0; i < N; i++) {
        if constexpr (S > 0) r.b[i] = fsr(a.b[i], lane_next(a.b[i]), S);
        else if constexpr (S < 0) r.b[i] = fsl(a.b[i], lane_prev(a.b[i]), -S);
        else r.b[i] = a.b[i];
    }
    return r;
}
// gt = a > b, lt = a < b
template <int N, class V>
PHX_FC_INLINE void compare(const Num<N, V> &a, const Num<N, V> &b, V &gt, V &lt) {
    gt = a.b[0] & ~b.b[0];
    lt = b.b[0] & ~a.b[0];
    for (int i = 1; i < N; i++) {
        const V ne = a.b[i] ^ b.b[i];
        gt = (a.b[i] & ~b.b[i]) | (~ne & gt);
        lt = (b.b[i] & ~a.b[i]) | (~ne & lt);
    }
}
// W[k] = sum of g over the indices k - 19 .. k + 20 of one stream (gc_frame_plot.py:44-59: the positions q + 3 m, m in [-19, 20])
template <class V>
PHX_FC_INLINE Num<6, V> gc_window(const V &g) {
    Num<1, V> t0; t0.b[0] = g;
    const Num<2, V> t1 = add<2>(t0, shifted<1>(t0));   // k .. k + 1
    const Num<3, V> t2 = add<3>(t1, shifted<2>(t1));   // k .. k + 3
    const Num<4, V> t3 = add<4>(t2, shifted<4>(t2));   // k .. k + 7
    const Num<5, V> t4 = add<5>(t3, shifted<8>(t3));   // k .. k + 15
    const Num<6, V> t5 = add<6>(t4, shifted<16>(t4));  // k .. k + 31
    return add<6>(shifted<-19>(t5), shifted<13>(t3));  // k - 19 .. k + 12, k + 13 .. k + 20
}

// ---- codon classes ----
// sets[c - 1]: the codons (c0 | c1 << 2 | c2 << 4) of class c = 1 .. 4 (CLS_FS, CLS_RS, CLS_FT, CLS_RT) after the elif priority of
// functions.py:198-215 (DParams.cls_tab): disjoint.  base[b] = the planes of base b (a, c, t, g: unambiguous only).
template <int F, class V>
PHX_FC_INLINE V codon_set(const P3<V> base[4], uint64_t set) {
    V acc = base[0].c[0] & ~base[0].c[0];
    const V c0[4] = {at<F, 0>(base[0]), at<F, 0>(base[1]), at<F, 0>(base[2]), at<F, 0>(base[3])};
    const V c1[4] = {at<F, 1>(base[0]), at<F, 1>(base[1]), at<F, 1>(base[2]), at<F, 1>(base[3])};
    const V c2[4] = {at<F, 2>(base[0]), at<F, 2>(base[1]), at<F, 2>(base[2]), at<F, 2>(base[3])};
    auto pick = [](const V x[4], int i) -> V { return i == 0 ? x[0] : (i == 1 ? x[1] : (i == 2 ? x[2] : x[3])); };
    for (uint64_t m = set; m; m &= m - 1) { // (uniform: a scalar loop on the device)
        const int code = __builtin_ctzll(m);
        acc = acc | (pick(c0, code & 3) & pick(c1, (code >> 2) & 3) & pick(c2, code >> 4));
    }
    return acc;
}
// the default codon tables (file_handling.py:51-53: atg / gtg / ttg, tag / tga / taa), as formulas
template <int F, class V>
PHX_FC_INLINE void codon_default(const P3<V> base[4], V out[4]) {
    const V a0 = at<F, 0>(base[0]), c0 = at<F, 0>(base[1]), t0 = at<F, 0>(base[2]), g0 = at<F, 0>(base[3]);
    const V a1 = at<F, 1>(base[0]), c1 = at<F, 1>(base[1]), t1 = at<F, 1>(base[2]), g1 = at<F, 1>(base[3]);
    const V a2 = at<F, 2>(base[0]), c2 = at<F, 2>(base[1]), t2 = at<F, 2>(base[2]), g2 = at<F, 2>(base[3]);
    out[0] = (a0 | g0 | t0) & t1 & g2;                   // atg gtg ttg
    out[1] = c0 & a1 & (t2 | c2 | a2);                   // cat cac caa
    out[2] = t0 & ((a1 & (g2 | a2)) | (g1 & a2));        // tag taa tga
    out[3] = a2 & (((c0 | t0) & t1) | (t0 & c1));        // cta tta tca
}
static constexpr uint64_t kDefaultSets[4] = {
    (1ull << (0 | 2 << 2 | 3 << 4)) | (1ull << (3 | 2 << 2 | 3 << 4)) | (1ull << (2 | 2 << 2 | 3 << 4)),
    (1ull << (1 | 0 << 2 | 2 << 4)) | (1ull << (1 | 0 << 2 | 1 << 4)) | (1ull << (1 | 0 << 2 | 0 << 4)),
    (1ull << (2 | 0 << 2 | 3 << 4)) | (1ull << (2 | 0 << 2 | 0 << 4)) | (1ull << (2 | 3 << 2 | 0 << 4)),
    (1ull << (1 | 2 << 2 | 0 << 4)) | (1ull << (2 | 2 << 2 | 0 << 4)) | (1ull << (2 | 1 << 2 | 0 << 4))};

// Background windows among the 96 positions of record w (32 indices of each stream) of a contig of L bases: vR = reverse windows
// dna[p : p + 21] start at every p < L; vF = forward windows dna[p - 20 : p + 1] exist for 20 <= p < L (the right-truncated ones of
// the last 20 starts are evaluated one by one: tail_windows)
static PHX_FC_INLINE void window_masks(int64_t L, int64_t w, uint32_t vF[3], uint32_t vR[3]) {
    for (int r = 0; r < 3; r++) {
        const int64_t nk = (L - r + 2) / 3;      // indices k with 3 k + r < L
        int64_t n = nk - 32 * w;
        n = n < 0 ? 0 : (n > 32 ? 32 : n);
        const uint32_t in = n >= 32 ? ~0u : ((1u << n) - 1u);
        int64_t lo = (20 - r + 2) / 3 - 32 * w;   // first index with 3 k + r >= 20
        lo = lo < 0 ? 0 : (lo > 32 ? 32 : lo);
        const uint32_t ge = lo >= 32 ? 0u : ~((1u << lo) - 1u);
        vR[r] = in;
        vF[r] = in & ge;
    }
}

// ---- the whole lane ----
template <class V>
struct FeatIn {
    V b0[3], b1[3], amb[3]; // the three planes of the three streams
    V vF[3], vR[3];         // background windows of the lane: forward window ending at p (20 <= p < L), reverse window starting at p (p < L)
};
enum { PL_FS, PL_RS, PL_FT, PL_RT, PL_GCF /* a > b, b > c, a > c */, PL_GCR = 7 /* c > b, b > a, c > a */, PL_ATGF = 10, PL_ATGR = 11, PL_N = 12 };
// Sink: plane(id, frame, V) for the twelve codon bitmaps of the three frames; tap(strand, stream, bit, V) for the bin planes (TAP)
template <bool TAP, bool DEFCOD, class V, class Sink>
PHX_FC_INLINE void feat_lane(const FeatIn<V> &in, const uint64_t sets[4], V cnt[28], V &gc, V &bad, Sink &sink) {
    // g + c after the counting remap (functions.py:159-163): bit 0 of the code, ambiguity codes included
    gc = in.b0[0] & ~in.b0[0];
    popc_add(gc, in.b0[0]); popc_add(gc, in.b0[1]); popc_add(gc, in.b0[2]);
    bad = (in.amb[0] & in.b1[0] & ~in.b0[0]) | (in.amb[1] & in.b1[1] & ~in.b0[1]) | (in.amb[2] & in.b1[2] & ~in.b0[2]);
    { // GC frame plot (gc_frame_plot.py:29-74, 7-28)
        const Num<6, V> W0 = gc_window(in.b0[0]), W1 = gc_window(in.b0[1]), W2 = gc_window(in.b0[2]);
        const Num<6, V> W0n = shifted<1>(W0), W1n = shifted<1>(W1); // the same streams one codon further: positions p + 3
        V gt01, lt01, gt12, lt12, gt02, lt02, gt20n, lt20n, gt10n, lt10n, gt21n, lt21n;
        compare(W0, W1, gt01, lt01);
        compare(W1, W2, gt12, lt12);
        compare(W0, W2, gt02, lt02);
        compare(W2, W0n, gt20n, lt20n);
        compare(W1, W0n, gt10n, lt10n);
        compare(W2, W1n, gt21n, lt21n);
        const V gt01n = fsr(gt01, lane_next(gt01), 1), lt01n = fsr(lt01, lane_next(lt01), 1); // W0n against W1n
        // frame 0: (a, b, c) = (W0, W1, W2); frame 1: (W1, W2, W0n); frame 2: (W2, W0n, W1n)
        sink.plane(PL_GCF + 0, 0, gt01); sink.plane(PL_GCF + 1, 0, gt12); sink.plane(PL_GCF + 2, 0, gt02);
        sink.plane(PL_GCR + 0, 0, lt12); sink.plane(PL_GCR + 1, 0, lt01); sink.plane(PL_GCR + 2, 0, lt02);
        sink.plane(PL_GCF + 0, 1, gt12); sink.plane(PL_GCF + 1, 1, gt20n); sink.plane(PL_GCF + 2, 1, gt10n);
        sink.plane(PL_GCR + 0, 1, lt20n); sink.plane(PL_GCR + 1, 1, lt12); sink.plane(PL_GCR + 2, 1, lt10n);
        sink.plane(PL_GCF + 0, 2, gt20n); sink.plane(PL_GCF + 1, 2, gt01n); sink.plane(PL_GCF + 2, 2, gt21n);
        sink.plane(PL_GCR + 0, 2, lt01n); sink.plane(PL_GCR + 1, 2, lt20n); sink.plane(PL_GCR + 2, 2, lt21n);
    }
    // the unambiguous bases
    V pa[3], pc[3], pt[3], pg[3], nac[3] /* not a, unambiguous */, ncc[3], ntc[3], ngc[3];
    for (int r = 0; r < 3; r++) {
        const V ok = ~in.amb[r];
        pa[r] = ok & ~in.b0[r] & ~in.b1[r];
        pc[r] = ok & in.b0[r] & ~in.b1[r];
        pt[r] = ok & ~in.b0[r] & in.b1[r];
        pg[r] = ok & in.b0[r] & in.b1[r];
        nac[r] = ok & (in.b0[r] | in.b1[r]);
        ngc[r] = ok & ~(in.b0[r] & in.b1[r]);
        ntc[r] = ok & (in.b0[r] | ~in.b1[r]);
        ncc[r] = ok & (~in.b0[r] | in.b1[r]);
    }
    const P3<V> A = make_p3(pa[0], pa[1], pa[2]), C = make_p3(pc[0], pc[1], pc[2]), T = make_p3(pt[0], pt[1], pt[2]), G = make_p3(pg[0], pg[1], pg[2]);
    { // codon classes and the literal 'atg' / 'cat' (Orf.start_codon() == 'atg', functions.py:263)
        const P3<V> base[4] = {A, C, T, G};
        auto frame = [&](auto FC) {
            constexpr int F = decltype(FC)::value;
            V cl[4];
            if (DEFCOD) codon_default<F>(base, cl);
            else for (int c = 0; c < 4; c++) cl[c] = codon_set<F>(base, sets[c]);
            sink.plane(PL_FS, F, cl[0]); sink.plane(PL_RS, F, cl[1]); sink.plane(PL_FT, F, cl[2]); sink.plane(PL_RT, F, cl[3]);
            sink.plane(PL_ATGF, F, at<F, 0>(A) & at<F, 1>(T) & at<F, 2>(G));
            sink.plane(PL_ATGR, F, at<F, 0>(C) & at<F, 1>(A) & at<F, 2>(T));
        };
        frame(ic<0>()); frame(ic<1>()); frame(ic<2>());
    }
    // background RBS bins (functions.py:168-169): forward windows on the strand as it is, read leftwards; reverse windows on the
    // complemented planes (g <-> c, a <-> t), read rightwards
    V binF[3][5], binR[3][5];
    rbs_strand<-1, TAP>(G, A, make_p3(nac[0], nac[1], nac[2]), make_p3(ngc[0], ngc[1], ngc[2]), in.vF, cnt, binF);
    rbs_strand<+1, TAP>(C, T, make_p3(ntc[0], ntc[1], ntc[2]), make_p3(ncc[0], ncc[1], ncc[2]), in.vR, cnt, binR);
    if (TAP)
        for (int r = 0; r < 3; r++)
            for (int i = 0; i < 5; i++) { sink.tap(0, r, i, binF[r][i]); sink.tap(1, r, i, binR[r][i]); }
}

// ---- score_rbs of ONE window, on linear masks (bit k <-> s[k], k = 0 .. 20; a symbol that is ambiguous or outside the contig is in
// no mask): the per-ORF bins of k_orf_stats, the right-truncated windows at the contig's end, and the cross-check of the planes above
static PHX_FC_INLINE uint32_t rbs_bin_linear(uint32_t G, uint32_t A, uint32_t nA, uint32_t nG) {
    const uint32_t gga = G & (G >> 1) & (A >> 2), ggag = gga & (G >> 3);
    const uint32_t gag = G & (A >> 1) & (G >> 2), gagg = gag & (G >> 3);
    const uint32_t agg = A & (G >> 1) & (G >> 2);
    uint32_t s[S_N];
    s[S_P2] = ggag & (G >> 4);
    s[S_P1] = s[S_P2] & (A >> 5);
    s[S_P3] = gagg & (A >> 4);
    s[S_P23] = s[S_P2] | s[S_P3];
    s[S_P5A] = ggag | gagg;
    s[S_P5B] = agg & (A >> 3);
    s[S_P5] = s[S_P5A] | s[S_P5B];
    s[S_P7] = agg | gag | gga;
    s[S_P6] = G & (G >> 1) & (nA >> 2) & (G >> 3) & (G >> 4);
    s[S_P8] = G & (A >> 1) & (nG >> 2) & (G >> 3) & (A >> 4);
    const uint32_t p4a = gga & (nG >> 3) & (G >> 4) & (A >> 5);
    s[S_P4] = p4a | (s[S_P6] & (A >> 5));
    s[S_X2] = s[S_P7] | p4a | s[S_P6];
    uint32_t hit = 1u;
    rule_seq([&](auto IC) {
        constexpr RbsRule ru = kRbsRules[decltype(IC)::value];
        constexpr uint32_t cm = ru.cls == 0 ? 0x18u : (ru.cls == 1 ? 0x7e0u : (ru.cls == 2 ? 0x1800u : 0xe000u)); // offsets 3-4, 5-10, 11-12, 13-15
        hit |= ((s[ru.src] & cm) ? 1u : 0u) << ru.score;
    });
    return 31u - (uint32_t)__builtin_clz(hit);
}

} // namespace phxfc
