// diag.hip -- diagnostic entry points of the C ABI: this box's HBM streaming ceilings, measured with the same kernel shapes the
// DESIGN.md section 4.1 argument rests on (tools/hbm_ceiling2.hip), so that bench.py can put them into the driver's JSON line
// next to the SpMV's achieved rate instead of quoting a profile of another day and another box.
//
// Persistent grids (CUs x 8 workgroups of 256 threads), 16 bytes per lane, every workgroup owns a contiguous span:
//   mode 0  read only, non-temporal loads, 4 in flight per lane             (the guide's "nt stream": 7.0-7.2 TB/s on this pool)
//   mode 1  write only
//   mode 2  copy 1 : 1
//   mode 3  mix 3 reads : 2 writes   (phase 1 of the R-MAT-26 SpMV: 2.49 GB in, 1.59 GB out)
//   mode 4  mix 16 reads : 1 write, nt loads   (phase 2 with the lean applicator: 2.67 GB in, 0.17 GB out)
//   mode 5  mix 7 reads : 1 write, nt loads    (phase 2 with the full applicator)
// Not on any product path; no reference counterpart (the reference has no device).
#include <algorithm>
#include <cstdint>

#include "gt_internal.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U>
__global__ void __launch_bounds__(256) k_diag_read(const v4f *__restrict__ a, uint64_t n, float *out) {
    v4f s = {0, 0, 0, 0};
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += (uint64_t)blockDim.x * U) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const uint64_t j = i + (uint64_t)u * blockDim.x; if (j < hi) v[u] = __builtin_nontemporal_load(a + j); else v[u] = s; }
#pragma unroll
        for (int u = 0; u < U; u++) s += v[u];
    }
    if (s.x + s.y + s.z + s.w == 1.2345e30f) *out = s.x;
}
template <int U>
__global__ void __launch_bounds__(256) k_diag_write(v4f *__restrict__ a, uint64_t n) {
    const v4f one = {1, 2, 3, 4};
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += (uint64_t)blockDim.x * U)
#pragma unroll
        for (int u = 0; u < U; u++) { const uint64_t j = i + (uint64_t)u * blockDim.x; if (j < hi) a[j] = one; }
}
// NR 16-byte loads per NW 16-byte stores; both spans contiguous per workgroup (what phase 1 / phase 2 look like to the memory system)
template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) k_diag_mix(const v4f *__restrict__ a, v4f *__restrict__ b, uint64_t n_in) {
    const uint64_t per = (n_in / gridDim.x / (256 * NR)) * (256 * NR);
    const v4f *src = a + blockIdx.x * per;
    v4f *dst = b + blockIdx.x * (per / NR * NW);
    for (uint64_t k = 0; k * 256 * NR < per; k++) {
        const uint64_t i = k * 256 * NR + threadIdx.x;
        v4f v[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
        v4f s = v[0];
#pragma unroll
        for (int u = 1; u < NR; u++) s += v[u];
#pragma unroll
        for (int u = 0; u < NW; u++) dst[(k * NW + u) * 256 + threadIdx.x] = s;
    }
}

}  // namespace

extern "C" int gt_diag_hbm_ceiling(int mode, uint64_t bytes, double *gbps) {
    GT_REQUIRE(gbps && mode >= 0 && mode <= 5, GT_ERR_INVALID, "gt_diag_hbm_ceiling: mode 0..5 and a result pointer");
    GT_REQUIRE(bytes >= (64ull << 20) && bytes <= (32ull << 30), GT_ERR_INVALID, "gt_diag_hbm_ceiling: 64 MiB .. 32 GiB per buffer");
    *gbps = 0;
    hipDeviceProp_t prop;
    int dev = 0;
    GT_HIP(hipGetDevice(&dev));
    GT_HIP(hipGetDeviceProperties(&prop, dev));
    const int grid = prop.multiProcessorCount * 8;
    const uint64_t n = bytes / 16;
    v4f *a = nullptr, *b = nullptr; float *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&] { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (out) (void)hipFree(out); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); };
#define DIAG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { gt_set_error("gt_diag_hbm_ceiling: %s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return GT_ERR_HIP; } } while (0)
    DIAG_HIP(hipMalloc((void **)&a, bytes)); DIAG_HIP(hipMalloc((void **)&b, bytes)); DIAG_HIP(hipMalloc((void **)&out, 8));
    DIAG_HIP(hipMemset(a, 0, bytes)); DIAG_HIP(hipMemset(b, 0, bytes));
    DIAG_HIP(hipEventCreate(&e0)); DIAG_HIP(hipEventCreate(&e1));
    double moved = 0;   // bytes one launch moves
    auto launch = [&] {
        switch (mode) {
            case 0: k_diag_read<4><<<grid, 256>>>(a, n, out); moved = (double)bytes; break;
            case 1: k_diag_write<4><<<grid, 256>>>(a, n); moved = (double)bytes; break;
            case 2: k_diag_mix<1, 1, false><<<grid, 256>>>(a, b, n); moved = 2.0 * (double)((n / grid / 256) * 256) * grid * 16; break;
            case 3: k_diag_mix<3, 2, false><<<grid, 256>>>(a, b, n); moved = (double)((n / grid / 768) * 768) * grid * 16 * (5.0 / 3.0); break;
            case 4: k_diag_mix<16, 1, true><<<grid, 256>>>(a, b, n); moved = (double)((n / grid / 4096) * 4096) * grid * 16 * (17.0 / 16.0); break;
            default: k_diag_mix<7, 1, true><<<grid, 256>>>(a, b, n); moved = (double)((n / grid / 1792) * 1792) * grid * 16 * (8.0 / 7.0); break;
        }
    };
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {   // the first launch warms up
        DIAG_HIP(hipEventRecord(e0, 0));
        launch();
        DIAG_HIP(hipEventRecord(e1, 0));
        DIAG_HIP(hipEventSynchronize(e1));
        float ms = 0; DIAG_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    DIAG_HIP(hipGetLastError());
#undef DIAG_HIP
    cleanup();
    *gbps = moved / (best * 1e-3) / 1e9;
    return GT_OK;
}
