// unit test: DPP wave_shl:1 (lane i takes lane i+1's value; lane 63 keeps `old`) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int *in, int *out) {
    const int v = in[threadIdx.x];
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-7, v, 0x130, 0xf, 0xf, false);
}
int main() {
    int h[64], r[64], *d, *o;
    for (int i = 0; i < 64; i++) h[i] = 100 + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) { const int want = i < 63 ? h[i + 1] : -7; if (r[i] != want) { if (bad < 8) printf("lane %d want %d got %d\n", i, want, r[i]); bad++; } }
    printf("wave_shl:1 mismatches: %d of 64\n", bad);
}
