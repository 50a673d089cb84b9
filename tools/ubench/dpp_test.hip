#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ int wv_max(int x) {
    uint32_t v = (uint32_t)x, o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;
    return __builtin_amdgcn_readlane((int)v, 63);
}
__global__ void k(const int *in, int *out) { out[blockIdx.x] = wv_max(in[blockIdx.x * 64 + threadIdx.x]); }
int main() {
    const int N = 2000; int h[N * 64], r[N], *d, *o;
    srand(1); int bad = 0;
    for (int i = 0; i < N * 64; i++) h[i] = (rand() % 7 == 0) ? rand() % 40 : 0;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<N, 64>>>(d, o); hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int b = 0; b < N; b++) { int m = 0; for (int i = 0; i < 64; i++) m = h[b * 64 + i] > m ? h[b * 64 + i] : m; if (m != r[b]) { if (bad < 5) { printf("block %d want %d got %d : ", b, m, r[b]); for (int i = 0; i < 64; i++) printf("%d ", h[b*64+i]); printf("\n"); } bad++; } }
    printf("wv_max mismatches: %d of %d\n", bad, N);
}
