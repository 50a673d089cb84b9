// Single-wavefront instruction cost probe for gfx950 (development aid): cycles per instruction for short
// instruction patterns when one wavefront has a SIMD to itself.   hipcc --offload-arch=gfx950 -O2 cpi.hip -o cpi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define ITERS 200
#define STR(x) #x
#define BENCH(name, body)                                                                    \
    __global__ void name(uint64_t *out, uint32_t *sink) {                                     \
        __shared__ uint32_t lds[1024];                                                       \
        for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 16) & 4095;               \
        __syncthreads();                                                                     \
        uint32_t a = threadIdx.x, b = 3, c = 5, d = 7, e = 11, f = 13, g = 17, h = 19;       \
        uint64_t A = a, B = 3, C = 5, D = 9;                                                 \
        uint32_t la = (threadIdx.x * 16) & 4095;                                             \
        uint64_t t0 = clock64();                                                             \
        for (int it = 0; it < ITERS; it++) {                                                 \
            _Pragma("unroll") for (int r = 0; r < REP; r++) { body }                         \
        }                                                                                    \
        uint64_t t1 = clock64();                                                             \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                              \
        sink[threadIdx.x] = a + b + c + d + e + f + g + h + (uint32_t)(A + B + C + D) + la;  \
    }
BENCH(k_dep_add32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
BENCH(k_ind_add32, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
BENCH(k_dep_add64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(A) : "v"(B));)
BENCH(k_ind_add64, asm volatile("v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2" : "+v"(A), "+v"(C) : "v"(B));)
BENCH(k_cmp64_cnd, asm volatile("v_cmp_lt_u64 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a) : "v"(A), "v"(B), "v"(b) : "vcc");)
BENCH(k_cmp32_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(c), "v"(b) : "vcc");)
BENCH(k_ind_cmp64, asm volatile("v_cmp_lt_u64 s[20:21], %0, %1\n v_cmp_lt_u64 s[22:23], %1, %0" : : "v"(A), "v"(B) : "s20", "s21", "s22", "s23");)
BENCH(k_carry4, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n s_nop 1\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n s_nop 1\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b) : "vcc");)
BENCH(k_salu_dep, asm volatile("s_add_u32 s20, s20, 1" : : : "s20");)
BENCH(k_snop, asm volatile("s_nop 0");)
BENCH(k_lds_chase, asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(la));)
BENCH(k_lds_chase128, asm volatile("ds_read_b128 v[40:43], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 4080, v40" : "+v"(la) : : "v40", "v41", "v42", "v43");)
BENCH(k_bperm, asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(la));)
BENCH(k_branch, asm volatile("s_cbranch_scc0 1f\n1:" : :);)
BENCH(k_dpp, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));)
BENCH(k_valu_salu_mix, asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(a) : "v"(b) : "s20");)
BENCH(k_readlane, asm volatile("v_readlane_b32 s20, %0, 3\n s_nop 3\n v_add_u32 %0, s20, %0" : "+v"(a) : : "s20");)
// fp64 operations of the solver's edge conversion
BENCH(k_dep_mul64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(A) : "v"(B));)
BENCH(k_ind_mul64, asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(A), "+v"(C) : "v"(B));)
BENCH(k_dep_trunc64, asm volatile("v_trunc_f64 %0, %0" : "+v"(A));)
BENCH(k_ind_trunc64, asm volatile("v_trunc_f64 %0, %0\n v_trunc_f64 %1, %1" : "+v"(A), "+v"(C));)
BENCH(k_ind_addf64, asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(A), "+v"(C) : "v"(B));)
BENCH(k_ind_minf64, asm volatile("v_min_f64 %0, %0, %2\n v_min_f64 %1, %1, %2" : "+v"(A), "+v"(C) : "v"(B));)
BENCH(k_dep_minf64, asm volatile("v_min_f64 %0, %0, %1" : "+v"(A) : "v"(B));)
BENCH(k_ind_cmpf64, asm volatile("v_cmp_nlt_f64 s[20:21], |%0|, %1\n v_cmp_nlt_f64 s[22:23], |%1|, %0" : : "v"(A), "v"(B) : "s20", "s21", "s22", "s23");)
BENCH(k_saveexec, asm volatile("s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(b) : "s20", "s21");)
BENCH(k_cmp_sor, asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n v_cmp_lt_u32 s[22:23], %1, %0\n s_or_b64 s[20:21], s[20:21], s[22:23]" : : "v"(a), "v"(b) : "s20", "s21", "s22", "s23");)
BENCH(k_cndmask_s, asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(a) : "v"(c), "v"(b) : "s20", "s21");)
BENCH(k_lds_read2_64, asm volatile("ds_read2_b64 v[40:43], %0 offset1:1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 4080, v40" : "+v"(la) : : "v40", "v41", "v42", "v43");)
int main() {
    uint64_t *out; uint32_t *sink;
    hipMalloc(&out, 8); hipMalloc(&sink, 256);
#define RUN(name, ninst)                                                                                         \
    { name<<<1, 64>>>(out, sink); name<<<1, 64>>>(out, sink); uint64_t t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);  \
      printf("%-18s %6.2f clk/pattern  (%d instr/pattern)  %6.2f clk/instr\n", STR(name), (double)t / (REP * ITERS), ninst, (double)t / (REP * ITERS) / ninst); }
    RUN(k_dep_add32, 1) RUN(k_ind_add32, 4) RUN(k_dep_add64, 1) RUN(k_ind_add64, 2) RUN(k_cmp64_cnd, 3) RUN(k_cmp32_cnd, 3) RUN(k_ind_cmp64, 2)
    RUN(k_carry4, 7) RUN(k_salu_dep, 1) RUN(k_snop, 1) RUN(k_lds_chase, 2) RUN(k_lds_chase128, 3) RUN(k_bperm, 2) RUN(k_branch, 1) RUN(k_dpp, 1) RUN(k_valu_salu_mix, 2) RUN(k_readlane, 3)
    RUN(k_dep_mul64, 1) RUN(k_ind_mul64, 2) RUN(k_dep_trunc64, 1) RUN(k_ind_trunc64, 2) RUN(k_ind_addf64, 2) RUN(k_dep_minf64, 1) RUN(k_ind_minf64, 2) RUN(k_ind_cmpf64, 2)
    RUN(k_saveexec, 3) RUN(k_cmp_sor, 3) RUN(k_cndmask_s, 3) RUN(k_lds_read2_64, 3)
    int dev; hipGetDevice(&dev); int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
    printf("clock64 counts shader clocks? device clock %d kHz; wall_clock64 is 100 MHz\n", khz);
    return 0;
}
