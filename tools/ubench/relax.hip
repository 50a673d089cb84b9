// Single-wavefront cost of one 8-edge relaxation group (128-bit distances) in three codings (development probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct W2 { uint64_t lo, hi; };
__device__ __forceinline__ W2 ld(const uint8_t *p) { const uint4 v = *(const uint4 *)p; W2 r; r.lo = ((uint64_t)v.y << 32) | v.x; r.hi = ((uint64_t)v.w << 32) | v.z; return r; }
__device__ __forceinline__ W2 add(W2 a, W2 b) { W2 r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull); return r; }
__device__ __forceinline__ bool lt(W2 a, W2 b) { return ((int64_t)a.hi < (int64_t)b.hi) | ((a.hi == b.hi) & (a.lo < b.lo)); }
__device__ __forceinline__ W2 sel(bool c, W2 a, W2 b) { W2 r; r.lo = c ? a.lo : b.lo; r.hi = c ? a.hi : b.hi; return r; }
typedef unsigned __int128 u128;

#define ITERS 2000
__global__ void k_tree(const uint64_t *in, uint64_t *out) {
    extern __shared__ __align__(16) uint8_t ring[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((uint64_t *)ring)[i] = in[i];
    __syncthreads();
    uint32_t cs[8]; W2 cw[8];
    for (int g = 0; g < 8; g++) { cs[g] = (uint32_t)(in[2048 + threadIdx.x * 8 + g] >> 20) & 8176; cw[g].lo = in[1024 + threadIdx.x * 16 + 2 * g]; cw[g].hi = in[1025 + threadIdx.x * 16 + 2 * g] & 0xff; }
    W2 m; m.lo = 0; m.hi = 0x6000000000000000ull; int arg = 0;
    uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        W2 x[8]; int xa[8];
#pragma unroll
        for (int g = 0; g < 8; g++) x[g] = ld(ring + cs[g]);
#pragma unroll
        for (int g = 0; g < 8; g++) { x[g] = add(x[g], cw[g]); xa[g] = g; }
#pragma unroll
        for (int w = 1; w < 8; w <<= 1)
#pragma unroll
            for (int g = 0; g + w < 8; g += 2 * w) { const bool l = lt(x[g + w], x[g]); x[g] = sel(l, x[g + w], x[g]); xa[g] = l ? xa[g + w] : xa[g]; }
        const bool l = lt(x[0], m); m = sel(l, x[0], m); arg = l ? xa[0] + it : arg;
        cs[it & 7] = (cs[it & 7] + 16) & 8176;
    }
    uint64_t t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    out[1 + threadIdx.x] = m.lo + m.hi + arg;
}
__global__ void k_chain(const uint64_t *in, uint64_t *out) {
    extern __shared__ __align__(16) uint8_t ring[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((uint64_t *)ring)[i] = in[i];
    __syncthreads();
    uint32_t cs[8]; u128 cw[8];
    for (int g = 0; g < 8; g++) { cs[g] = (uint32_t)(in[2048 + threadIdx.x * 8 + g] >> 20) & 8176; cw[g] = ((u128)(in[1025 + threadIdx.x * 16 + 2 * g] & 0xff) << 64) | in[1024 + threadIdx.x * 16 + 2 * g]; }
    u128 m = (u128)0x6000000000000000ull << 64; int arg = 0;
    uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        u128 x[8];
#pragma unroll
        for (int g = 0; g < 8; g++) x[g] = *(const u128 *)(ring + cs[g]);
#pragma unroll
        for (int g = 0; g < 8; g++) { const u128 c = x[g] + cw[g]; const bool l = (int32_t)(uint32_t)((c - m) >> 96) < 0; m = l ? c : m; arg = l ? g + it : arg; }
        cs[it & 7] = (cs[it & 7] + 16) & 8176;
    }
    uint64_t t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    out[1 + threadIdx.x] = (uint64_t)m + (uint64_t)(m >> 64) + arg;
}
// hand-scheduled: 4 independent add chains interleaved (distinct carry registers), pairwise minima by subtract + sign
#define ADD4(o, a, b) \
    "v_add_co_u32 %[" #o "0], s[20:21], %[" #a "0], %[" #b "0]\n"
__global__ void k_asm(const uint64_t *in, uint64_t *out) {
    extern __shared__ __align__(16) uint8_t ring[];
    for (int i = threadIdx.x; i < 1024; i += 64) ((uint64_t *)ring)[i] = in[i];
    __syncthreads();
    uint32_t cs[8]; uint32_t w[8][4];
    for (int g = 0; g < 8; g++) { cs[g] = (uint32_t)(in[2048 + threadIdx.x * 8 + g] >> 20) & 8176; const uint64_t a = in[1024 + threadIdx.x * 16 + 2 * g], b = in[1025 + threadIdx.x * 16 + 2 * g] & 0xff; w[g][0] = (uint32_t)a; w[g][1] = (uint32_t)(a >> 32); w[g][2] = (uint32_t)b; w[g][3] = (uint32_t)(b >> 32); }
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0x60000000u; int arg = 0;
    uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        uint4 x[8];
#pragma unroll
        for (int g = 0; g < 8; g++) x[g] = *(const uint4 *)(ring + cs[g]);
        // candidates: two chains per asm block, interleaved
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
            asm volatile(
                "v_add_co_u32 %0, s[20:21], %0, %8\n"
                "v_add_co_u32 %4, s[22:23], %4, %12\n"
                "s_nop 0\n"
                "v_addc_co_u32 %1, s[20:21], %1, %9, s[20:21]\n"
                "v_addc_co_u32 %5, s[22:23], %5, %13, s[22:23]\n"
                "s_nop 0\n"
                "v_addc_co_u32 %2, s[20:21], %2, %10, s[20:21]\n"
                "v_addc_co_u32 %6, s[22:23], %6, %14, s[22:23]\n"
                "s_nop 0\n"
                "v_addc_co_u32 %3, s[20:21], %3, %11, s[20:21]\n"
                "v_addc_co_u32 %7, s[22:23], %7, %15, s[22:23]\n"
                : "+v"(x[g].x), "+v"(x[g].y), "+v"(x[g].z), "+v"(x[g].w), "+v"(x[g + 1].x), "+v"(x[g + 1].y), "+v"(x[g + 1].z), "+v"(x[g + 1].w)
                : "v"(w[g][0]), "v"(w[g][1]), "v"(w[g][2]), "v"(w[g][3]), "v"(w[g + 1][0]), "v"(w[g + 1][1]), "v"(w[g + 1][2]), "v"(w[g + 1][3])
                : "s20", "s21", "s22", "s23");
        }
        int xa[8];
#pragma unroll
        for (int g = 0; g < 8; g++) xa[g] = g;
        // tree of minima: min(x[g], x[g+w]) -> x[g]; two pairs per block, interleaved; t = b - a, take b if t < 0
#pragma unroll
        for (int w2 = 1; w2 < 8; w2 <<= 1)
#pragma unroll
            for (int g = 0; g + w2 < 8; g += 4 * w2) {
                const int h = g + 2 * w2 < 8 ? g + 2 * w2 : g; // second pair (or the same one again at the last level)
                uint32_t t0_, t1_;
                asm volatile(
                    "v_sub_co_u32 %16, s[20:21], %4, %0\n"
                    "v_sub_co_u32 %17, s[22:23], %12, %8\n"
                    "s_nop 0\n"
                    "v_subb_co_u32 %16, s[20:21], %5, %1, s[20:21]\n"
                    "v_subb_co_u32 %17, s[22:23], %13, %9, s[22:23]\n"
                    "s_nop 0\n"
                    "v_subb_co_u32 %16, s[20:21], %6, %2, s[20:21]\n"
                    "v_subb_co_u32 %17, s[22:23], %14, %10, s[22:23]\n"
                    "s_nop 0\n"
                    "v_subb_co_u32 %16, s[20:21], %7, %3, s[20:21]\n"
                    "v_subb_co_u32 %17, s[22:23], %15, %11, s[22:23]\n"
                    "v_cmp_gt_i32 s[20:21], 0, %16\n"
                    "v_cmp_gt_i32 s[22:23], 0, %17\n"
                    "s_nop 0\n"
                    "v_cndmask_b32 %0, %0, %4, s[20:21]\n"
                    "v_cndmask_b32 %8, %8, %12, s[22:23]\n"
                    "v_cndmask_b32 %1, %1, %5, s[20:21]\n"
                    "v_cndmask_b32 %9, %9, %13, s[22:23]\n"
                    "v_cndmask_b32 %2, %2, %6, s[20:21]\n"
                    "v_cndmask_b32 %10, %10, %14, s[22:23]\n"
                    "v_cndmask_b32 %3, %3, %7, s[20:21]\n"
                    "v_cndmask_b32 %11, %11, %15, s[22:23]\n"
                    "v_cndmask_b32 %18, %18, %19, s[20:21]\n"
                    "v_cndmask_b32 %20, %20, %21, s[22:23]\n"
                    : "+v"(x[g].x), "+v"(x[g].y), "+v"(x[g].z), "+v"(x[g].w), "+v"(x[g + w2].x), "+v"(x[g + w2].y), "+v"(x[g + w2].z), "+v"(x[g + w2].w),
                      "+v"(x[h].x), "+v"(x[h].y), "+v"(x[h].z), "+v"(x[h].w), "+v"(x[h + w2].x), "+v"(x[h + w2].y), "+v"(x[h + w2].z), "+v"(x[h + w2].w),
                      "=&v"(t0_), "=&v"(t1_), "+v"(xa[g]), "+v"(xa[g + w2]), "+v"(xa[h]), "+v"(xa[h + w2])
                    :
                    : "s20", "s21", "s22", "s23");
            }
        // against the running minimum
        {
            uint32_t t;
            asm volatile(
                "v_sub_co_u32 %8, vcc, %4, %0\n s_nop 1\n v_subb_co_u32 %8, vcc, %5, %1, vcc\n s_nop 1\n v_subb_co_u32 %8, vcc, %6, %2, vcc\n s_nop 1\n v_subb_co_u32 %8, vcc, %7, %3, vcc\n"
                "v_cmp_gt_i32 vcc, 0, %8\n s_nop 1\n"
                "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc\n v_cndmask_b32 %9, %9, %10, vcc\n"
                : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(x[0].x), "+v"(x[0].y), "+v"(x[0].z), "+v"(x[0].w), "=&v"(t), "+v"(arg), "+v"(xa[0]) : : "vcc");
        }
        cs[it & 7] = (cs[it & 7] + 16) & 8176;
    }
    uint64_t t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    out[1 + threadIdx.x] = (uint64_t)m0 + m1 + m2 + m3 + arg;
}
int main() {
    uint64_t *in, *out, h[4096];
    for (int i = 0; i < 4096; i++) h[i] = (uint64_t)rand() * 2654435761ull;
    hipMalloc(&in, sizeof(h)); hipMalloc(&out, 65 * 8); hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    uint64_t t;
#define RUN(k) k<<<1, 64, 8192>>>(in, out); k<<<1, 64, 8192>>>(in, out); hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); printf("%-8s %7.1f clk per 8-edge group  (%5.1f per edge)\n", #k, (double)t / ITERS, (double)t / ITERS / 8);
    RUN(k_tree) RUN(k_chain) RUN(k_asm)
    return 0;
}
