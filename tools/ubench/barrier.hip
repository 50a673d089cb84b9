// What would splitting a phase of k_sssp_wave over TWO wavefronts of one workgroup cost?  (development aid, gfx950)
// one: one wavefront does K units of phase-like work (an LDS read + a dependent 64-bit add + a min) and one LDS write + wait per iteration;
// two: two wavefronts (one workgroup, different SIMDs) do K/2 each, write their partial minimum, s_barrier, read the other's, combine.
//   hipcc --offload-arch=gfx950 -O2 barrier.hip -o barrier && ./barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2000
template <int K, int WAVES>
__global__ void k_phase(uint64_t *out, uint64_t *sink) {
    __shared__ uint64_t ring[1024];
    __shared__ uint64_t part[2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 1024; i += 64 * WAVES) ring[i] = (uint64_t)i * 7919u;
    __syncthreads();
    uint64_t acc = lane;
    uint32_t idx = (lane * 9) & 1023;
    const uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        uint64_t m = ~0ull;
#pragma unroll
        for (int k = 0; k < K / WAVES; k++) { // the lane's cached in-edges: ring read + weight add + min
            const uint64_t d = ring[(idx + 17 * (k * WAVES + w)) & 1023] + (acc & 0xff) + (uint64_t)k;
            m = d < m ? d : m;
        }
        if (WAVES == 2) {
            part[w][lane] = m;
            __syncthreads();
            const uint64_t o = part[w ^ 1][lane];
            m = o < m ? o : m;
        }
        // head lane writes the ring, everybody waits, vote
        ring[(idx + it) & 1023] = m;
        if (WAVES == 2) __syncthreads(); else __builtin_amdgcn_s_waitcnt(0xc07f);
        const uint64_t vote = __ballot(m != acc);
        acc = m + (vote & 1);
        idx = (idx + 1) & 1023;
    }
    const uint64_t t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main() {
    uint64_t *out, *sink; hipMalloc(&out, 8); hipMalloc(&sink, 8 * 128);
#define RUN(K, W) { k_phase<K, W><<<1, 64 * W>>>(out, sink); k_phase<K, W><<<1, 64 * W>>>(out, sink); uint64_t t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost); \
    printf("K=%3d in-edges per lane, %d wavefront(s): %7.1f clk (clock64 units; x 24 = %7.0f if they were 100 MHz ticks: they are shader clocks) per phase\n", K, W, (double)t / ITERS, (double)t / ITERS * 24.0); }
    RUN(4, 1) RUN(4, 2) RUN(16, 1) RUN(16, 2) RUN(32, 1) RUN(32, 2) RUN(64, 1) RUN(64, 2)
    return 0;
}
