// lds_atomic_bench.hip -- cycles per wave-instruction of LDS atomics on gfx950 (design input for the staging row of
// pb.hip's phase 1): ds_add_f32 / ds_add_f64 / ds_add_u32 / ds_min_u32 without return, element stride 4/8/16 B,
// G lanes per address (1 = all distinct, 64 = one address).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_atomic_bench tools/lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class T, int OP> __device__ __forceinline__ void op(T *p, T v) {
    if constexpr (OP == 0) { if constexpr (sizeof(T) == 8 || __is_floating_point(T)) unsafeAtomicAdd(p, v); else atomicAdd(p, v); }
    else if constexpr (OP == 1) atomicMin(p, v);
    else *p = v;   // plain store for comparison
}
template <class T, int OP, int STRIDE /* bytes */>
__global__ void __launch_bounds__(1024) k(int group, int iters, unsigned long long *cycles, T *sink) {
    __shared__ char buf[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += blockDim.x) ((uint32_t *)buf)[i] = 0;
    __syncthreads();
    T *p = (T *)(buf + wave * 4096 + (lane / group) * STRIDE);
    const T v = (T)1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) op<T, OP>(p + (u & 1) * (1024 / sizeof(T)), v);
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (sink && *p == (T)12345) *sink = *p;
}
template <class T, int OP, int STRIDE> int run(const char *name, int waves) {
    unsigned long long *d; CK(hipMalloc(&d, 8 * 256));
    for (int group : {1, 2, 4, 16, 64}) {
        const int iters = 2000;
        k<T, OP, STRIDE><<<256, waves * 64>>>(group, iters, d, nullptr);
        CK(hipDeviceSynchronize());
        unsigned long long h[256]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        double c = 0; for (int i = 0; i < 256; i++) c += h[i];
        c /= 256;
        printf("%-28s stride %2d B, %2d waves/CU, %2d lanes per address: %7.1f cycles per wave-instruction per CU-slot (%.1f per instruction with %d waves interleaved)\n",
               name, STRIDE, waves, group, c / (iters * 8.0) , c / (iters * 8.0) / waves, waves);
    }
    CK(hipFree(d));
    return 0;
}
int main() {
    for (int waves : {4, 16}) {
        if (run<float, 0, 4>("ds_add_f32", waves)) return 1;
        if (run<float, 0, 8>("ds_add_f32", waves)) return 1;
        if (run<double, 0, 8>("ds_add_f64", waves)) return 1;
        if (run<double, 0, 16>("ds_add_f64", waves)) return 1;
        if (run<uint32_t, 0, 8>("ds_add_u32", waves)) return 1;
        if (run<uint32_t, 1, 8>("ds_min_u32", waves)) return 1;
        if (run<float, 2, 8>("ds_write_b32", waves)) return 1;
    }
    return 0;
}
