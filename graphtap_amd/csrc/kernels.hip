// kernels.hip -- generalized SpMV over the owned tile-row (K1/K5 of SURVEY 2.2).
//
// Replaces the inner loops of Vertex_Program::spmv_stationary / spmv_nonstationary
// (src/vp/vertex_program.hpp:1162-1173, 1490-1503): for every stored entry (r, c[, w])
//   plus : y[r] += x[c]                     min : if x[c] != INF: y[r] = min(y[r], x[c] (+ w))
//
// v0 "edge-parallel" kernel: one lane per stored entry, the entry's column id read from JI
// (4 B more per entry than TCSC's JA walk, but perfectly balanced under R-MAT skew), the
// accumulate is a device-scope atomic on y. It is the correctness baseline that the blocked
// kernels are checked against; see DESIGN.md for the roofline of each.
#include "gt_internal.h"

namespace {

constexpr int TPB = 256;

template <int SR> struct SemiringT;
template <> struct SemiringT<GT_PLUS_F64> { using T = double; };
template <> struct SemiringT<GT_PLUS_U32> { using T = uint32_t; };
template <> struct SemiringT<GT_MIN_U32> { using T = uint32_t; };
template <> struct SemiringT<GT_MINPLUS_U32> { using T = uint32_t; };

template <int SR, class T>
__device__ __forceinline__ void combine_one(T *__restrict__ y, uint32_t r, T xv, uint32_t w) {
    if constexpr (SR == GT_PLUS_F64) {
        unsafeAtomicAdd(&y[r], xv);  // global_atomic_add_f64, no CAS loop
    } else if constexpr (SR == GT_PLUS_U32) {
        atomicAdd(&y[r], xv);
    } else {
        if (xv == GT_INF) return;  // vp:1492
        T t = (SR == GT_MINPLUS_U32) ? xv + w : xv;
        if (t < y[r]) atomicMin(&y[r], t);  // plain read first: most candidates lose
    }
}

template <int SR>
__global__ void __launch_bounds__(TPB) k_spmv_edge(const uint32_t *__restrict__ IA, const uint32_t *__restrict__ JI,
                                                   const uint32_t *__restrict__ A, uint64_t nnz,
                                                   const uint32_t *__restrict__ xslot /* compressed column -> slot of x, or null */,
                                                   const typename SemiringT<SR>::T *__restrict__ x,
                                                   typename SemiringT<SR>::T *__restrict__ y) {
    using T = typename SemiringT<SR>::T;
    const uint64_t nvec = nnz >> 2;  // 16-byte loads: 4 entries per lane
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint4 *IA4 = reinterpret_cast<const uint4 *>(IA);
    const uint4 *JI4 = reinterpret_cast<const uint4 *>(JI);
    const uint4 *A4 = reinterpret_cast<const uint4 *>(A);
    for (uint64_t v = tid; v < nvec; v += nthreads) {
        uint4 r = IA4[v], c = JI4[v];
        if (xslot) c = make_uint4(xslot[c.x], xslot[c.y], xslot[c.z], xslot[c.w]);
        uint4 w = make_uint4(0, 0, 0, 0);
        if constexpr (SR == GT_MINPLUS_U32) w = A4[v];
        T x0 = x[c.x], x1 = x[c.y], x2 = x[c.z], x3 = x[c.w];
        combine_one<SR, T>(y, r.x, x0, w.x);
        combine_one<SR, T>(y, r.y, x1, w.y);
        combine_one<SR, T>(y, r.z, x2, w.z);
        combine_one<SR, T>(y, r.w, x3, w.w);
    }
    for (uint64_t e = (nvec << 2) + tid; e < nnz; e += nthreads)
        combine_one<SR, T>(y, IA[e], x[xslot ? xslot[JI[e]] : JI[e]], (SR == GT_MINPLUS_U32) ? A[e] : 0u);
}

}  // namespace

int gt_launch_spmv(const gt_graph *g, int semiring, const void *x, void *y, hipStream_t s, bool x_is_f32, const void *owner, uint64_t epoch,
                   uint32_t slice_lo, uint32_t slice_hi, unsigned phases, const gt_pr_epilogue *epi) {
    const uint32_t K = g->info.x_slices;
    if (slice_hi > K) slice_hi = K;
    if (g->spmv_variant == GT_SPMV_PB_F32MSG) return gt_pb_spmv(g, semiring, x, y, s, true, x_is_f32, owner, epoch, slice_lo, slice_hi, phases, epi);
    GT_REQUIRE(!x_is_f32, GT_ERR_STATE, "this program was created for GT_SPMV_PB_F32MSG (f32 message vector); select the SpMV variant before creating programs");
    if (g->spmv_variant == GT_SPMV_PB) return gt_pb_spmv(g, semiring, x, y, s, false, false, owner, epoch, slice_lo, slice_hi, phases, epi);
    GT_REQUIRE(epi == nullptr, GT_ERR_STATE, "the fused PageRank epilogue needs a propagation-blocking SpMV variant");
    if (phases ? !(phases & GT_PB_PHASE2) : slice_hi < K) return GT_OK;   // the edge kernel is not sliced: everything happens with the last stage
    return gt_launch_spmv_edge(g, semiring, x, y, s);
}

int gt_launch_spmv_edge(const gt_graph *g, int semiring, const void *x, void *y, hipStream_t s) {
    const uint64_t nnz = g->info.nnz_local;
    if (nnz == 0) return GT_OK;
    uint64_t blocks = ((nnz >> 2) + TPB - 1) / TPB;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    switch (semiring) {
        case GT_PLUS_F64:
            k_spmv_edge<GT_PLUS_F64><<<(unsigned)blocks, TPB, 0, s>>>(g->IA, g->JI, g->A, nnz, g->xslot, (const double *)x, (double *)y);
            break;
        case GT_PLUS_U32:
            k_spmv_edge<GT_PLUS_U32><<<(unsigned)blocks, TPB, 0, s>>>(g->IA, g->JI, g->A, nnz, g->xslot, (const uint32_t *)x, (uint32_t *)y);
            break;
        case GT_MIN_U32:
            k_spmv_edge<GT_MIN_U32><<<(unsigned)blocks, TPB, 0, s>>>(g->IA, g->JI, g->A, nnz, g->xslot, (const uint32_t *)x, (uint32_t *)y);
            break;
        case GT_MINPLUS_U32:
            GT_REQUIRE(g->A != nullptr, GT_ERR_INVALID, "min-plus SpMV needs a weighted graph (the reference builds sssp with -DHAS_WEIGHT, Makefile:26-27)");
            k_spmv_edge<GT_MINPLUS_U32><<<(unsigned)blocks, TPB, 0, s>>>(g->IA, g->JI, g->A, nnz, g->xslot, (const uint32_t *)x, (uint32_t *)y);
            break;
        default:
            gt_set_error("unknown semiring %d", semiring);
            return GT_ERR_INVALID;
    }
    GT_HIP(hipGetLastError());
    return GT_OK;
}
