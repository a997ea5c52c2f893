// hbm_ceiling.hip -- measured streaming ceilings of the box (read-only, write-only, copy, and
// segment-scattered writes shaped like PB phase 1), quoted in DESIGN.md next to the 8 TB/s spec.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_ceiling tools/hbm_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_read(const double2 *__restrict__ a, uint64_t n, double *out) {
    double s = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e300) *out = s;
}
__global__ void k_write(double2 *__restrict__ a, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a[i] = make_double2(1.0, 2.0);
}
__global__ void k_write8(double *__restrict__ a, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a[i] = 1.0;
}
__global__ void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) b[i] = a[i];
}
template <int U>
__global__ void k_copy_u(const double2 *__restrict__ a, double2 *__restrict__ b, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < n) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < n) b[i + u * stride] = v[u];
    }
}
// shaped like PB phase 1: 2-byte index stream -> LDS table lookup -> 8-byte value stream
template <int U>
__global__ void __launch_bounds__(1024) k_expand16(const uint16_t *__restrict__ idx, double *__restrict__ out, uint64_t n) {
    __shared__ double tab[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = i;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
        uint16_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (i + u * stride < n) ? idx[i + u * stride] : 0;
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < n) out[i + u * stride] = tab[v[u] & 8191];
    }
}
// f32 output, one entry per lane per access (2 B in, 4 B out) vs four entries per lane (8 B in, 16 B out)
template <int U>
__global__ void __launch_bounds__(1024) k_expand16_f32(const uint16_t *__restrict__ idx, float *__restrict__ out, uint64_t n) {
    __shared__ float tab[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = i;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
        uint16_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (i + u * stride < n) ? idx[i + u * stride] : 0;
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < n) out[i + u * stride] = tab[v[u] & 8191];
    }
}
template <int U>
__global__ void __launch_bounds__(1024) k_expand16x4_f32(const ushort4 *__restrict__ idx, float4 *__restrict__ out, uint64_t n4) {
    __shared__ float tab[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = i;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n4; i += stride * U) {
        ushort4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (i + u * stride < n4) ? idx[i + u * stride] : make_ushort4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < n4)
            out[i + u * stride] = make_float4(tab[v[u].x & 8191], tab[v[u].y & 8191], tab[v[u].z & 8191], tab[v[u].w & 8191]);
    }
}
// every wave writes runs of `run` doubles at pseudo-random run-aligned places (each place written once)
__global__ void k_scatter_runs(double *__restrict__ a, uint64_t nruns, uint32_t run, uint64_t mul) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < nruns; r += nw) {
        uint64_t dst = (r * mul) % nruns;   // mul coprime with nruns: a permutation
        for (uint32_t j = lane; j < run; j += 64) a[dst * run + j] = 1.0;
    }
}
int main() {
    const uint64_t bytes = 8ull << 30, n16 = bytes / 16, n8 = bytes / 8;
    double2 *a, *b; double *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double gb, auto launch) {
        float best = 1e30f;
        for (int it = 0; it < 5; it++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-34s %8.3f ms  %7.1f GB/s\n", name, best, gb / (best * 1e-3));
    };
    const int grid = 256 * 16, tpb = 256;
    timeit("read-only 16B/lane", bytes / 1e9, [&] { k_read<<<grid, tpb>>>(a, n16, out); });
    timeit("write-only 16B/lane", bytes / 1e9, [&] { k_write<<<grid, tpb>>>(a, n16); });
    timeit("write-only 8B/lane", bytes / 1e9, [&] { k_write8<<<grid, tpb>>>((double *)a, n8); });
    timeit("copy 16B/lane (r+w bytes)", 2 * bytes / 1e9, [&] { k_copy<<<grid, tpb>>>(a, b, n16); });
    timeit("copy 16B/lane x4 unrolled (r+w)", 2 * bytes / 1e9, [&] { k_copy_u<4><<<grid, tpb>>>(a, b, n16); });
    timeit("copy 16B/lane x8 unrolled (r+w)", 2 * bytes / 1e9, [&] { k_copy_u<8><<<256 * 8, 512>>>(a, b, n16); });
    timeit("expand u16->f64 via LDS, U=8 (r+w)", (bytes + bytes / 4) / 1e9, [&] { k_expand16<8><<<512, 1024>>>((const uint16_t *)b, (double *)a, n8); });
    timeit("expand u16->f64 via LDS, U=2 (r+w)", (bytes + bytes / 4) / 1e9, [&] { k_expand16<2><<<512, 1024>>>((const uint16_t *)b, (double *)a, n8); });
    timeit("expand u16->f64, U=8, 2048 WGs (r+w)", (bytes + bytes / 4) / 1e9, [&] { k_expand16<8><<<2048, 1024>>>((const uint16_t *)b, (double *)a, n8); });
    const uint64_t ne = 1ull << 30;   // 2^30 entries like R-MAT-26
    timeit("expand u16->f32 1/lane U=8", ne * 6 / 1e9, [&] { k_expand16_f32<8><<<512, 1024>>>((const uint16_t *)b, (float *)a, ne); });
    timeit("expand u16->f32 1/lane U=16", ne * 6 / 1e9, [&] { k_expand16_f32<16><<<512, 1024>>>((const uint16_t *)b, (float *)a, ne); });
    timeit("expand u16->f32 4/lane U=2", ne * 6 / 1e9, [&] { k_expand16x4_f32<2><<<512, 1024>>>((const ushort4 *)b, (float4 *)a, ne / 4); });
    timeit("expand u16->f32 4/lane U=4", ne * 6 / 1e9, [&] { k_expand16x4_f32<4><<<512, 1024>>>((const ushort4 *)b, (float4 *)a, ne / 4); });
    for (uint32_t run : {16u, 32u, 64u, 128u, 256u, 1024u}) {
        uint64_t nruns = n8 / run;
        char nm[64]; snprintf(nm, sizeof nm, "scattered write runs of %4u x 8B", run);
        timeit(nm, bytes / 1e9, [&] { k_scatter_runs<<<grid, tpb>>>((double *)a, nruns, run, 2654435761ull | 1); });
    }
    return 0;
}
