// tools/s_atomic_test.hip -- does gfx950 execute SCALAR memory atomics (s_atomic_add ... glc, returned through lgkmcnt, not vmcnt)?
// Every wave of a launch draws numbers from one counter per workgroup; the host checks that each number was handed out exactly once
// and prints the rate. (Round 4: a trip counter for the phase-1 kernels whose LDS is full; a VECTOR atomic's return sits in the
// in-order vmcnt queue of their software pipeline and drains it -- profiles/r04/ab_trip_counter_in_hbm_rejected.txt.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/s_atomic_test tools/s_atomic_test.hip && timeout -k 5 60 ./tools/s_atomic_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) k(unsigned *ctr, unsigned *seen, unsigned per_wg) {
    unsigned *c = ctr + blockIdx.x;
    for (;;) {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(c) : "memory");
        if (v >= per_wg) break;
        if ((threadIdx.x & 63) == 0) atomicAdd(&seen[blockIdx.x * per_wg + v], 1u);
    }
}
int main() {
    const unsigned wgs = 256, per = 4096;
    unsigned *ctr, *seen;
    hipMalloc(&ctr, wgs * 4); hipMalloc(&seen, (size_t)wgs * per * 4);
    hipMemset(ctr, 0, wgs * 4); hipMemset(seen, 0, (size_t)wgs * per * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k<<<wgs, 1024>>>(ctr, seen, per); hipEventRecord(b);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned> h((size_t)wgs * per); hipMemcpy(h.data(), seen, h.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (unsigned x : h) bad += (x != 1);
    printf("s_atomic_add: %zu of %zu numbers not handed out exactly once; %.3f ms for %u draws per workgroup x %u workgroups x 16 waves drawing (%.2f us per draw and wave)\n",
           bad, h.size(), ms, per, wgs, ms * 1e3 / (per / 16.0));
    return bad != 0;
}
