// mall_pingpong.hip -- would a row-super-blocked propagation-blocking SpMV keep its value stream in the 256 MiB
// Infinity Cache? Model: S super-blocks; per block kernel A streams L/S bytes in and writes V/S bytes, kernel B reads
// those V/S bytes back plus R/S more (L = LCOL, V = VAL, R = LROW of R-MAT-26: 2.2 / 3.3 / 1.7 GB). All blocks use
// distinct memory. S = 1 is today's two-kernel SpMV; if large S is not clearly faster the cache does not help.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mall_pingpong tools/mall_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k_a(const float4 *__restrict__ l, uint64_t nl, float4 *__restrict__ v, uint64_t nv, float *sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    float s = 0;
    for (uint64_t i = t; i < nl; i += stride) { float4 a = l[i]; s += a.x + a.y + a.z + a.w; }
    for (uint64_t i = t; i < nv; i += stride) v[i] = make_float4(s, 1.f, 2.f, 3.f);
    if (s == 1.2345e30f) *sink = s;
}
__global__ void __launch_bounds__(1024) k_b(const float4 *__restrict__ v, uint64_t nv, const float4 *__restrict__ r, uint64_t nr, float *sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    float s = 0;
    for (uint64_t i = t; i < nv; i += stride) { float4 a = v[i]; s += a.x + a.y + a.z + a.w; }
    for (uint64_t i = t; i < nr; i += stride) { float4 a = r[i]; s += a.x + a.y + a.z + a.w; }
    if (s == 1.2345e30f) *sink = s;
}

int main() {
    const uint64_t LB = 2200ull << 20, VB = 3300ull << 20, RB = 1700ull << 20;
    float4 *L, *V, *R; float *sink;
    CK(hipMalloc(&L, LB)); CK(hipMalloc(&V, VB)); CK(hipMalloc(&R, RB)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(L, 0, LB)); CK(hipMemset(V, 0, VB)); CK(hipMemset(R, 0, RB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 2;
    for (int S : {1, 4, 8, 16, 24, 32, 48, 64, 128}) {
        const uint64_t nl = LB / 16 / S, nv = VB / 16 / S, nr = RB / 16 / S;
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            for (int s = 0; s < S; s++) {
                k_a<<<grid, 1024>>>(L + s * nl, nl, V + s * nv, nv, sink);
                k_b<<<grid, 1024>>>(V + s * nv, nv, R + s * nr, nr, sink);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("S=%3d  V block %6.1f MB  total %.3f ms  (%.2f TB/s over L+2V+R = %.1f GB)\n", S, VB / 1e6 / S, best,
               (LB + 2 * VB + RB) / 1e9 / best, (LB + 2 * VB + RB) / 1e9);
    }
    return 0;
}
