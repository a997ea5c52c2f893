// mall_reuse.hip -- does a value stream that is written and read back through the SAME small buffer stay in the 256 MiB
// Infinity Cache (no HBM write-back, no HBM read)?  tools/mall_pingpong.hip gave every super-block its own memory, so every
// dirty line still had to reach HBM; here S super-blocks reuse one buffer of V/S bytes: kernel A streams L/S bytes in and
// writes the buffer, kernel B reads the buffer back plus R/S more.  L, V, R as on R-MAT-26 with f32 messages: 2.2 / 1.6 / 0.8 GB.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mall_reuse tools/mall_reuse.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT>
__global__ void __launch_bounds__(1024) k_a(const float4 *__restrict__ l, uint64_t nl, float4 *__restrict__ v, uint64_t nv, float *sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    float s = 0;
    for (uint64_t i = t; i < nl; i += stride) { v4f a = NT ? __builtin_nontemporal_load((const v4f *)l + i) : ((const v4f *)l)[i]; s += a.x + a.y + a.z + a.w; }
    for (uint64_t i = t; i < nv; i += stride) v[i] = make_float4(s, 1.f, 2.f, 3.f);
    if (s == 1.2345e30f) *sink = s;
}
template <bool NT>
__global__ void __launch_bounds__(1024) k_b(const float4 *__restrict__ v, uint64_t nv, const float4 *__restrict__ r, uint64_t nr, float *sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    float s = 0;
    for (uint64_t i = t; i < nv; i += stride) { float4 a = v[i]; s += a.x + a.y + a.z + a.w; }
    for (uint64_t i = t; i < nr; i += stride) { v4f a = NT ? __builtin_nontemporal_load((const v4f *)r + i) : ((const v4f *)r)[i]; s += a.x + a.y + a.z + a.w; }
    if (s == 1.2345e30f) *sink = s;
}

int main() {
    const uint64_t LB = 2200ull << 20, VB = 1600ull << 20, RB = 800ull << 20;
    float4 *L, *V, *R; float *sink;
    CK(hipMalloc(&L, LB)); CK(hipMalloc(&V, VB)); CK(hipMalloc(&R, RB)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(L, 0, LB)); CK(hipMemset(V, 0, VB)); CK(hipMemset(R, 0, RB));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 2;
    for (int nt = 0; nt < 2; nt++)
    for (int reuse = 0; reuse < 2; reuse++)
    for (int S : {1, 4, 8, 12, 16, 24, 32, 64}) {
        const uint64_t nl = LB / 16 / S, nv = VB / 16 / S, nr = RB / 16 / S;
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            for (int s = 0; s < S; s++) {
                float4 *v = reuse ? V : V + s * nv;
                if (nt) { k_a<true><<<grid, 1024>>>(L + s * nl, nl, v, nv, sink); k_b<true><<<grid, 1024>>>(v, nv, R + s * nr, nr, sink); }
                else { k_a<false><<<grid, 1024>>>(L + s * nl, nl, v, nv, sink); k_b<false><<<grid, 1024>>>(v, nv, R + s * nr, nr, sink); }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%s %s S=%3d  V block %7.1f MB  total %.3f ms  (%.2f TB/s over L+2V+R = %.1f GB; L+R alone = %.1f GB -> %.2f TB/s)\n",
               nt ? "nt-streams" : "plain     ", reuse ? "one buffer reused " : "distinct buffers  ", S, VB / 1e6 / S, best,
               (LB + 2 * VB + RB) / 1e9 / best, (LB + 2 * VB + RB) / 1e9, (LB + RB) / 1e9, (LB + RB) / 1e9 / best);
        fflush(stdout);
    }
    return 0;
}
