// tests/feat_core_host.cpp — TEST HARNESS: phanotate_amd/csrc/phx_feat_core.h instantiated over a 64-lane array type, so that the
// bit-sliced feature code the GPU runs (same source, V = uint32_t + DPP there) is checked against the oracle in the CPU suite.
// Not part of libphx.so; built by tests/test_feat_core_host.py with g++.
#include <stdint.h>
#include <string.h>

#include <vector>

namespace phxfc {
struct HV { // one 32-bit word per lane of a simulated wavefront
    uint32_t l[64];
};
inline HV operator&(const HV &a, const HV &b) { HV r; for (int i = 0; i < 64; i++) r.l[i] = a.l[i] & b.l[i]; return r; }
inline HV operator|(const HV &a, const HV &b) { HV r; for (int i = 0; i < 64; i++) r.l[i] = a.l[i] | b.l[i]; return r; }
inline HV operator^(const HV &a, const HV &b) { HV r; for (int i = 0; i < 64; i++) r.l[i] = a.l[i] ^ b.l[i]; return r; }
inline HV operator~(const HV &a) { HV r; for (int i = 0; i < 64; i++) r.l[i] = ~a.l[i]; return r; }
inline HV fsr(const HV &c, const HV &n, int s) { HV r; for (int i = 0; i < 64; i++) r.l[i] = (c.l[i] >> s) | (n.l[i] << (32 - s)); return r; }
inline HV fsl(const HV &c, const HV &p, int s) { HV r; for (int i = 0; i < 64; i++) r.l[i] = (c.l[i] << s) | (p.l[i] >> (32 - s)); return r; }
inline HV lane_prev(const HV &a) { HV r; r.l[0] = 0; for (int i = 1; i < 64; i++) r.l[i] = a.l[i - 1]; return r; }  // DPP wave_shr:1 (lane 0: nothing)
inline HV lane_next(const HV &a) { HV r; r.l[63] = 0; for (int i = 0; i < 63; i++) r.l[i] = a.l[i + 1]; return r; } // DPP wave_shl:1
inline void popc_add(HV &acc, const HV &x) { for (int i = 0; i < 64; i++) acc.l[i] += (uint32_t)__builtin_popcount(x.l[i]); }
} // namespace phxfc

#include "../phanotate_amd/csrc/phx_feat_core.h"

using namespace phxfc;

namespace {
struct Sink {
    uint32_t *planes; // [12][3][nrec]
    uint32_t *tapbuf; // [nrec][2][3][5]
    int64_t nrec;
    int64_t v0;       // virtual word of lane 0
    void plane(int id, int f, const HV &x) {
        for (int i = 1; i < 63; i++) { const int64_t w = v0 + i; if (w >= 0 && w < nrec) planes[((int64_t)id * 3 + f) * nrec + w] = x.l[i]; }
    }
    void tap(int s, int r, int b, const HV &x) {
        if (!tapbuf) return;
        for (int i = 1; i < 63; i++) { const int64_t w = v0 + i; if (w >= 0 && w < nrec) tapbuf[((w * 2 + s) * 3 + r) * 5 + b] = x.l[i]; }
    }
};
} // namespace

// recs: nrec + 2 records of 9 words ([stream][b0, b1, amb]); record 0 and record nrec + 1 are "outside" pads, record 1 + w is word w of
// the (single) contig.  Outputs: planes [12][3][nrec], cnt[28], gc, bad; tap (may be null) [nrec][2][3][5].
extern "C" int feat_core_host(const uint32_t *recs, int64_t nrec, int64_t L, int defcod, const uint64_t *sets, uint32_t *planes, uint32_t *cnt28, uint32_t *gc_out,
                              uint32_t *bad_out, uint32_t *tap) {
    memset(cnt28, 0, 28 * 4);
    *gc_out = 0; *bad_out = 0;
    for (int64_t blk = 0; blk * 62 < nrec; blk++) {
        const int64_t v0 = blk * 62 - 1; // lane 0 is the halo in front
        FeatIn<HV> in;
        for (int i = 0; i < 64; i++) {
            int64_t w = v0 + i;
            const bool real = w >= 0 && w < nrec;
            int64_t rec = w + 1;
            if (rec < 0) rec = 0;
            if (rec > nrec + 1) rec = nrec + 1;
            const uint32_t *q = recs + rec * 9;
            uint32_t vF[3] = {0, 0, 0}, vR[3] = {0, 0, 0};
            if (real) window_masks(L, w, vF, vR);
            for (int r = 0; r < 3; r++) {
                in.b0[r].l[i] = q[r * 3 + 0]; in.b1[r].l[i] = q[r * 3 + 1]; in.amb[r].l[i] = q[r * 3 + 2];
                in.vF[r].l[i] = vF[r]; in.vR[r].l[i] = vR[r];
            }
        }
        HV cnt[28], gc, bad;
        memset(cnt, 0, sizeof cnt);
        Sink sink{planes, tap, nrec, v0};
        if (tap) { if (defcod) feat_lane<true, true>(in, sets, cnt, gc, bad, sink); else feat_lane<true, false>(in, sets, cnt, gc, bad, sink); }
        else { if (defcod) feat_lane<false, true>(in, sets, cnt, gc, bad, sink); else feat_lane<false, false>(in, sets, cnt, gc, bad, sink); }
        for (int i = 1; i < 63; i++) {
            const int64_t w = v0 + i;
            if (w < 0 || w >= nrec) continue;
            for (int s = 0; s < 28; s++) cnt28[s] += cnt[s].l[i];
            *gc_out += gc.l[i];
            if (bad.l[i]) *bad_out = 1;
        }
    }
    return 0;
}

// score_rbs of one window given as its symbols s[0 .. n) (codes a0 c1 t2 g3, 4 = ambiguous), n <= 21
extern "C" int rbs_bin_linear_host(const uint8_t *s, int n) {
    uint32_t G = 0, A = 0, nA = 0, nG = 0;
    for (int k = 0; k < n && k < 21; k++) {
        if (s[k] > 3) continue;
        if (s[k] == 3) G |= 1u << k;
        if (s[k] == 0) A |= 1u << k;
        if (s[k] != 0) nA |= 1u << k;
        if (s[k] != 3) nG |= 1u << k;
    }
    return (int)rbs_bin_linear(G, A, nA, nG);
}
