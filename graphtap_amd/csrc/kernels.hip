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
#include <hipcub/hipcub.hpp>

#include <cstdlib>

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
                   uint32_t slice_lo, uint32_t slice_hi, unsigned phases, const gt_pr_epilogue *epi, bool skip_source, bool f64_messages,
                   uint32_t part_lo, uint32_t part_hi) {
    const uint32_t K = g->info.x_slices;
    if (slice_hi > K) slice_hi = K;
    if (g->spmv_variant == GT_SPMV_PB_F32MSG && f64_messages) {   // a converge-mode PageRank on a graph whose default is f32 messages (gt_program_prepare)
        GT_REQUIRE(!x_is_f32, GT_ERR_STATE, "f64 messages were asked for with an f32 message vector");
        return gt_pb_spmv(g, semiring, x, y, s, false, false, owner, epoch, slice_lo, slice_hi, phases, epi, skip_source, part_lo, part_hi);
    }
    if (g->spmv_variant == GT_SPMV_PB_F32MSG) return gt_pb_spmv(g, semiring, x, y, s, true, x_is_f32, owner, epoch, slice_lo, slice_hi, phases, epi, skip_source, part_lo, part_hi);
    GT_REQUIRE(!x_is_f32, GT_ERR_STATE, "this program was created for GT_SPMV_PB_F32MSG (f32 message vector); select the SpMV variant before creating programs");
    if (g->spmv_variant == GT_SPMV_PB) return gt_pb_spmv(g, semiring, x, y, s, false, false, owner, epoch, slice_lo, slice_hi, phases, epi, skip_source, part_lo, part_hi);
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

// ------------------------------------------------------------------ sparse frontier: SpMSpV of the min semirings
namespace {

// active columns (message != infinity()) of the message vector and the entries they hold
__global__ void k_frontier_count(const uint32_t *__restrict__ x, uint32_t x_len, const uint32_t *__restrict__ xcol, const uint32_t *__restrict__ JA,
                                 unsigned long long *__restrict__ out) {
    unsigned long long n = 0, e = 0;
    for (uint32_t sl = blockIdx.x * blockDim.x + threadIdx.x; sl < x_len; sl += gridDim.x * blockDim.x) {
        if (x[sl] == GT_INF) continue;
        const uint32_t c = xcol ? xcol[sl] : sl;
        if (c == 0xFFFFFFFFu) continue;
        n++; e += JA[c + 1] - JA[c];
    }
    for (int o = 32; o > 0; o >>= 1) { n += __shfl_down(n, o); e += __shfl_down(e, o); }
    __shared__ unsigned long long part[2][TPB / 64];   // one pair of atomics per workgroup: they all hit the same two words
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = n; part[1][threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; w++) { n += part[0][w]; e += part[1][w]; }
        if (n) { atomicAdd(&out[0], n); atomicAdd(&out[1], e); }
    }
}
__global__ void __launch_bounds__(TPB) k_frontier_list(const uint32_t *__restrict__ x, uint32_t x_len, const uint32_t *__restrict__ xcol, const uint32_t *__restrict__ JA,
                                unsigned int *__restrict__ cursor, uint32_t *__restrict__ col, uint32_t *__restrict__ val, uint32_t *__restrict__ deg) {
    // one reservation per WORKGROUP and round: every reservation hits the same word, and one per wave is 0.5 M of them on a
    // 33 M-slot x (5 ms by itself once most waves hold an active column)
    __shared__ unsigned wave_n[TPB / 64];
    __shared__ unsigned round_base;
    const uint32_t n_round = (x_len + TPB - 1) / TPB * TPB, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t sl = blockIdx.x * blockDim.x + threadIdx.x; sl < n_round; sl += gridDim.x * blockDim.x) {
        uint32_t c = 0xFFFFFFFFu, v = GT_INF;
        if (sl < x_len) { v = x[sl]; if (v != GT_INF) c = xcol ? xcol[sl] : sl; }
        const bool act = c != 0xFFFFFFFFu;
        const uint64_t b = __ballot(act);
        if (lane == 0) wave_n[wave] = (unsigned)__popcll((unsigned long long)b);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int w = 0; w < TPB / 64; w++) { const unsigned n = wave_n[w]; wave_n[w] = total; total += n; }
            round_base = total ? atomicAdd(cursor, total) : 0u;
        }
        __syncthreads();
        if (act) {
            const uint32_t o = round_base + wave_n[wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
            col[o] = c; val[o] = v; deg[o] = JA[c + 1] - JA[c];
        }
        __syncthreads();   // wave_n / round_base are rewritten by the next round
    }
}
// one thread per entry of an active column: the column is found by bisection of the entry offsets. EMIT: the rows whose
// accumulator this pass lowers -- the only rows apply() can change -- get their bit set in `mark_bits` (k_rows_from_marks turns
// the bits into the row list). `total_dev` != null: the number of entries is read from the device.
template <bool WEIGHTED, bool EMIT>
__global__ void __launch_bounds__(TPB) k_spmspv_min(const uint32_t *__restrict__ col, const uint32_t *__restrict__ val, const uint32_t *__restrict__ off, uint32_t nact,
                             uint64_t total, const unsigned long long *__restrict__ total_dev, const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA,
                             const uint32_t *__restrict__ A, uint32_t *__restrict__ y, uint32_t *__restrict__ mark_bits) {
    if (total_dev) total = *total_dev;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nact;   // last i with off[i] <= t
        while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (off[mid] <= t) lo = mid; else hi = mid; }
        const uint32_t e = JA[col[lo]] + (uint32_t)(t - off[lo]);
        const uint32_t r = IA[e];
        const uint32_t m = WEIGHTED ? val[lo] + A[e] : val[lo];
        if (m < y[r]) {
            const uint32_t old = atomicMin(&y[r], m);
            if (EMIT && m < old) atomicOr(&mark_bits[r >> 5], 1u << (r & 31u));
        }
    }
}
// eight lanes per active column (no bisection): the columns of a frontier are short -- 1.3 to 8 entries on average in the
// iterations this path is for; the few long ones (more than `big` entries) are left to the entry-parallel kernel
template <bool WEIGHTED>
__global__ void __launch_bounds__(TPB) k_spmspv_cols(const uint32_t *__restrict__ col, const uint32_t *__restrict__ val, const uint32_t *__restrict__ deg, uint32_t nact,
                              uint32_t big, const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA, const uint32_t *__restrict__ A,
                              uint32_t *__restrict__ y, uint32_t *__restrict__ mark_bits) {
    constexpr uint32_t LPC = 8, GPB = TPB / LPC;
    const uint32_t sub = threadIdx.x & (LPC - 1);
    for (uint32_t gi = blockIdx.x * GPB + threadIdx.x / LPC; gi < nact; gi += gridDim.x * GPB) {
        const uint32_t d = deg[gi];
        if (d == 0 || d > big) continue;
        const uint32_t e0 = JA[col[gi]], m0 = val[gi];
        for (uint32_t k = sub; k < d; k += LPC) {
            const uint32_t r = IA[e0 + k];
            const uint32_t m = WEIGHTED ? m0 + A[e0 + k] : m0;
            if (m < y[r]) {
                const uint32_t old = atomicMin(&y[r], m);
                if (m < old) atomicOr(&mark_bits[r >> 5], 1u << (r & 31u));
            }
        }
    }
}
// keeps the entry counts of the long columns only (the others are done) and adds them up
__global__ void __launch_bounds__(TPB) k_keep_big(uint32_t *__restrict__ deg, uint32_t nact, uint32_t big, unsigned long long *__restrict__ total_big) {
    unsigned long long e = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nact; i += gridDim.x * blockDim.x) {
        const uint32_t d = deg[i];
        if (d > big) e += d; else if (d) deg[i] = 0;
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
    __shared__ unsigned long long part[TPB / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < TPB / 64; w++) e += part[w]; if (e) atomicAdd(total_big, e); }
}
// mark bits -> ascending list of rows; clears the bits. One reservation per 4096 words.
__global__ void __launch_bounds__(TPB) k_rows_from_marks(uint32_t *__restrict__ mark_bits, uint32_t nwords, uint32_t *__restrict__ rows, unsigned int *__restrict__ rows_n) {
    constexpr uint32_t PER = 16, SPAN = PER * TPB;
    __shared__ unsigned wave_n[TPB / 64];
    __shared__ unsigned span_base;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nspan = (nwords + SPAN - 1) / SPAN;
    for (uint32_t sp = blockIdx.x; sp < nspan; sp += gridDim.x) {
        const uint32_t w0 = sp * SPAN + threadIdx.x * PER;   // 16 consecutive words per thread: its rows come out in order
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) if (w0 + k < nwords) cnt += (uint32_t)__popc(mark_bits[w0 + k]);
        uint32_t inc = cnt;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += t; }
        if (lane == 63) wave_n[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int w = 0; w < TPB / 64; w++) { const unsigned n = wave_n[w]; wave_n[w] = total; total += n; }
            span_base = total ? atomicAdd(rows_n, total) : 0u;
        }
        __syncthreads();
        if (cnt) {
            uint32_t o = span_base + wave_n[wave] + inc - cnt;
            for (uint32_t k = 0; k < PER && w0 + k < nwords; k++) {
                uint32_t m = mark_bits[w0 + k];
                if (!m) continue;
                mark_bits[w0 + k] = 0;
                while (m) { const uint32_t b = (uint32_t)__ffs((int)m) - 1; m &= m - 1; rows[o++] = (w0 + k) * 32 + b; }
            }
        }
        __syncthreads();
    }
}

__global__ void k_preload() {}

// ---- bottom-up BFS step on a SYMMETRIC graph: the in-neighbours of row r are the entries of the same vertex's column
// rows of the vertices BFS has not reached (+ the entries of their columns), one reservation per 4096 rows
// -- and, as by-products of the same pass, the bitmap of the rows ON the current level (bit r of `level_bits`, 64 rows per wave and
// trip; 27 M rows = 3.4 MB: the step's probes stay in L2) and the entries of their columns (entries[1]: how heavy the frontier is)
__global__ void __launch_bounds__(TPB) k_bu_collect(const uint32_t *__restrict__ IR, const uint32_t *__restrict__ R2C, const uint32_t *__restrict__ JA,
                                                    const uint32_t *__restrict__ hops, uint32_t level, uint32_t nr, uint32_t *__restrict__ list,
                                                    unsigned int *__restrict__ list_n, unsigned long long *__restrict__ entries,
                                                    uint32_t *__restrict__ level_bits) {
    constexpr uint32_t PER = 16, SPAN = PER * TPB;
    __shared__ unsigned wave_n[TPB / 64];
    __shared__ unsigned span_base;
    __shared__ unsigned long long wave_e[TPB / 64], wave_f[TPB / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nspan = (nr + SPAN - 1) / SPAN;
    unsigned long long e = 0, ef = 0;
    for (uint32_t sp = blockIdx.x; sp < nspan; sp += gridDim.x) {
        uint32_t mask = 0, cnt = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t r = sp * SPAN + k * TPB + threadIdx.x;
            const uint32_t h = r < nr ? hops[IR[r]] : 0u;
            const bool on_level = r < nr && h == level;
            if (r < nr && (h == GT_INF || on_level)) {
                const uint32_t c = R2C[r];
                if (c != 0xFFFFFFFFu) {
                    const uint32_t d = JA[c + 1] - JA[c];
                    if (on_level) ef += d; else { mask |= 1u << k; cnt++; e += d; }
                }
            }
            const uint64_t b = __ballot(on_level);   // rows r - lane .. r - lane + 63 (TPB is a multiple of 64)
            if (lane == 0) { const uint32_t w = (sp * SPAN + k * TPB + wave * 64) >> 5; level_bits[w] = (uint32_t)b; level_bits[w + 1] = (uint32_t)(b >> 32); }
        }
        uint32_t inc = cnt;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += t; }
        if (lane == 63) wave_n[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int w = 0; w < TPB / 64; w++) { const unsigned n = wave_n[w]; wave_n[w] = total; total += n; }
            span_base = total ? atomicAdd(list_n, total) : 0u;
        }
        __syncthreads();
        uint32_t o = span_base + wave_n[wave] + inc - cnt;
        for (uint32_t k = 0; mask; k++, mask >>= 1)
            if (mask & 1u) list[o++] = sp * SPAN + k * TPB + threadIdx.x;
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) { e += __shfl_down(e, o); ef += __shfl_down(ef, o); }
    if (lane == 0) { wave_e[wave] = e; wave_f[wave] = ef; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; w++) { e += wave_e[w]; ef += wave_f[w]; }
        if (e) atomicAdd(entries, e);
        if (ef) atomicAdd(entries + 1, ef);
    }
}
// 16 lanes per unreached row: y[r] = min id among its neighbours that are on the current level (bfs.h:52-54, 61-63: the
// messages of the active columns, min-combined). rows_out[i] = the row if it found one, ~0u if not: position for position,
// no compaction (nearly every unreached row finds a parent in the steps this is chosen for, and a compaction would be a
// reservation on one word per 16 rows)
__global__ void __launch_bounds__(TPB) k_bu_step(const uint32_t *__restrict__ list, const unsigned int *__restrict__ n_dev, const uint32_t *__restrict__ R2C,
                                                 const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA, const uint32_t *__restrict__ IR,
                                                 const uint32_t *__restrict__ hops, uint32_t level, uint32_t vid_base, gt_vidmap vm, uint32_t *__restrict__ y,
                                                 uint32_t *__restrict__ rows_out) {
    constexpr uint32_t LPR = 16, GPB = TPB / LPR;
    const uint32_t n = *n_dev, sub = threadIdx.x & (LPR - 1);
    for (uint32_t gi = blockIdx.x * GPB + threadIdx.x / LPR; gi < n; gi += gridDim.x * GPB) {
        uint32_t cand = GT_INF;
        const uint32_t r = list[gi], c = R2C[r];
        for (uint32_t e = JA[c] + sub, e1 = JA[c + 1]; e < e1; e += LPR) {
            const uint32_t u = IR[IA[e]];
            if (hops[u] == level) { const uint32_t id = gt_vid_of(vm, (uint64_t)vid_base + u); cand = id < cand ? id : cand; }
        }
        for (int o = LPR / 2; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(cand, o); cand = t < cand ? t : cand; }
        if (sub == 0) {
            const bool found = cand != GT_INF;
            if (found && cand < y[r]) y[r] = cand;
            rows_out[gi] = found ? r : 0xFFFFFFFFu;
        }
    }
}

// The same step with EARLY EXIT. The entries of a column are stored by ascending row, and rows ascend with the vertex ids
// (one rank, no id map): the FIRST neighbour found on the current level is the minimum the push sweep's min-combiner would
// leave (bfs.h:61-63) -- so a row stops at its first hit, which is what makes the step pay while the frontier is LARGE (the hub
// frontier of iteration 1 on R-MAT: ~2 probes per row instead of a sweep over every entry). A probe is one bit of `level_bits`.
// Two kernels: ONE THREAD per unreached row probes its first 4 neighbours (16 lanes per row
// were bound by the chain of dependent loads -- row, column, entry range, entries, bits -- with only 32 K rows in flight:
// 3.7 ms for the 31.8 M unreached rows of R-MAT-26's iteration 1, as long as the push sweep); the few rows that have more
// neighbours and found none among the first 4 go to a second list, which 16 lanes per row finish.
constexpr uint32_t BU_PROBE = 4;   // entries of a row the first kernel looks at
struct alignas(16) BuQuad { uint32_t v[4]; };
__device__ __forceinline__ bool bu_on_level(const uint32_t *__restrict__ level_bits, uint32_t nb) { return ((level_bits[nb >> 5] >> (nb & 31u)) & 1u) != 0; }
// FN[r] = the first four entries of row r's column (~0u beyond its end), 16 bytes per row, built once per program: read through IA
// the probes fetched a cache line per row for 16 useful bytes -- 1.1 ms of line traffic for the 31.8 M rows of that iteration
__global__ void k_bu_first_neighbours(const uint32_t *__restrict__ R2C, const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA, uint32_t nr, BuQuad *__restrict__ FN) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += gridDim.x * blockDim.x) {
        BuQuad q{{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};
        const uint32_t c = R2C[r];
        if (c != 0xFFFFFFFFu) { const uint32_t e0 = JA[c], e1 = JA[c + 1]; for (uint32_t j = 0; j < 4 && e0 + j < e1; j++) q.v[j] = IA[e0 + j]; }
        FN[r] = q;
    }
}
__global__ void __launch_bounds__(TPB) k_bu_first_probe(const uint32_t *__restrict__ list, const unsigned int *__restrict__ n_dev, const BuQuad *__restrict__ FN,
                                                        const uint32_t *__restrict__ IR, const uint32_t *__restrict__ level_bits, uint32_t vid_base,
                                                        uint32_t *__restrict__ y, uint32_t *__restrict__ rows_out, uint32_t *__restrict__ long_list,
                                                        unsigned int *__restrict__ long_n) {
    // the long rows are appended with ONE reservation per span of 2048 rows (one per wave on the one counter was 500 K same-address
    // atomics for the 31.8 M rows of R-MAT-26's iteration 1: 1.1 of the kernel's 1.2 ms)
    constexpr uint32_t PER = 8, SPAN = PER * TPB;
    __shared__ unsigned wave_n[TPB / 64];
    __shared__ unsigned span_base;
    const uint32_t n = *n_dev, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nspan = (n + SPAN - 1) / SPAN;
    for (uint32_t sp = blockIdx.x; sp < nspan; sp += gridDim.x) {
        uint32_t mask = 0, cnt = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t gi = sp * SPAN + k * TPB + threadIdx.x;
            const bool valid = gi < n;
            uint32_t r = 0, hitnb = 0xFFFFFFFFu;
            BuQuad q{{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}};
            if (valid) { r = list[gi]; q = FN[r]; }
            // the bits two at a time: ~57 % of the probes of a heavy frontier hit
            if (q.v[0] != 0xFFFFFFFFu) {
                const bool h0 = bu_on_level(level_bits, q.v[0]), h1 = q.v[1] != 0xFFFFFFFFu && bu_on_level(level_bits, q.v[1]);
                hitnb = h0 ? q.v[0] : h1 ? q.v[1] : 0xFFFFFFFFu;
                if (hitnb == 0xFFFFFFFFu && q.v[2] != 0xFFFFFFFFu) {
                    const bool h2 = bu_on_level(level_bits, q.v[2]), h3 = q.v[3] != 0xFFFFFFFFu && bu_on_level(level_bits, q.v[3]);
                    hitnb = h2 ? q.v[2] : h3 ? q.v[3] : 0xFFFFFFFFu;
                }
            }
            const bool found = hitnb != 0xFFFFFFFFu;
            if (found) { const uint32_t cand = vid_base + IR[hitnb]; if (cand < y[r]) y[r] = cand; }
            if (rows_out && valid) rows_out[gi] = found ? r : 0xFFFFFFFFu;   // (a long row: until the second kernel finds its parent)
            if (valid && !found && q.v[3] != 0xFFFFFFFFu) { mask |= 1u << k; cnt++; }   // four entries and no hit: there may be more
        }
        uint32_t inc = cnt;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += t; }
        if (lane == 63) wave_n[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int w = 0; w < TPB / 64; w++) { const unsigned c = wave_n[w]; wave_n[w] = total; total += c; }
            span_base = total ? atomicAdd(long_n, total) : 0u;
        }
        __syncthreads();
        uint32_t o = span_base + wave_n[wave] + inc - cnt;
        for (uint32_t k = 0; mask; k++, mask >>= 1)
            if (mask & 1u) long_list[o++] = sp * SPAN + k * TPB + threadIdx.x;
        __syncthreads();
    }
}
template <uint32_t LPR>
__global__ void __launch_bounds__(TPB) k_bu_first_rest(const uint32_t *__restrict__ long_list, const unsigned int *__restrict__ long_n, const uint32_t *__restrict__ list,
                                                       const uint32_t *__restrict__ R2C, const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA,
                                                       const uint32_t *__restrict__ IR, const uint32_t *__restrict__ level_bits, uint32_t vid_base,
                                                       uint32_t *__restrict__ y, uint32_t *__restrict__ rows_out) {
    constexpr uint32_t GPB = TPB / LPR;
    const uint32_t n = *long_n, sub = threadIdx.x & (LPR - 1), gbase = (threadIdx.x & 63u) & ~(LPR - 1);   // first lane of this row's lanes inside the wave
    for (uint32_t li = blockIdx.x * GPB + threadIdx.x / LPR; li < n; li += gridDim.x * GPB) {   // (uniform per row: its LPR lanes stay together)
        uint32_t cand = GT_INF;
        const uint32_t gi = long_list[li], r = list[gi], c = R2C[r];
        for (uint32_t base = JA[c] + BU_PROBE, e1 = JA[c + 1]; base < e1; base += LPR) {
            const uint32_t e = base + sub;
            uint32_t nb = 0; bool hit = false;
            if (e < e1) { nb = IA[e]; hit = bu_on_level(level_bits, nb); }
            const uint32_t m = (uint32_t)(__ballot(hit) >> gbase) & (LPR == 32 ? 0xFFFFFFFFu : ((1u << LPR) - 1u));
            if (m) {
                const uint32_t first = (uint32_t)__ffs((int)m) - 1u;
                cand = vid_base + IR[__shfl(nb, (int)(gbase + first))];
                break;
            }
        }
        if (sub == 0 && cand != GT_INF) {
            if (cand < y[r]) y[r] = cand;
            if (rows_out) rows_out[gi] = r;
        }
    }
}

// ---- the same step WITHOUT the collecting pass: BFS's apply kernels keep two row bitmaps (gt_internal.h, bu_reached): one pass over
// the rows, 2048 at a time -- a reached row costs its bit; an unreached one probes as above. The rows that found a parent are
// appended to `rows_out` (one reservation per 2048 rows; null: the full apply follows), the long ones to `long_list`.
__global__ void __launch_bounds__(TPB) k_bu_probe_all(uint32_t *reached_bits, const uint32_t *__restrict__ level_bits, uint32_t *__restrict__ next_bits,
                                                      const BuQuad *__restrict__ FN, const uint32_t *__restrict__ IR, uint32_t nr, uint32_t vid_base,
                                                      uint32_t *__restrict__ y, uint32_t *__restrict__ rows_out, unsigned int *__restrict__ rows_n,
                                                      uint32_t *__restrict__ long_list, unsigned int *__restrict__ long_n) {
    // The rows that find a parent ARE the next level (BFS's apply accepts every one of them): a wave writes the two words of its 64
    // rows into `next_bits` and ORs them into `reached_bits` (its own words: nobody else touches them), so the apply that follows
    // leaves the bitmaps alone (3.7 M atomic ORs cost the row-list apply of R-MAT-26's iteration 2 0.1 ms).
    constexpr uint32_t PER = 8, SPAN = PER * TPB;
    __shared__ unsigned wave_n[TPB / 64];
    __shared__ unsigned span_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nspan = (nr + SPAN - 1) / SPAN;
    auto reserve = [&](uint32_t cnt, unsigned int *cursor) -> uint32_t {   // exclusive position of this thread's first item; all threads call
        uint32_t inc = cnt;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += t; }
        if (lane == 63) wave_n[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int w = 0; w < TPB / 64; w++) { const unsigned c = wave_n[w]; wave_n[w] = total; total += c; }
            span_base = total ? atomicAdd(cursor, total) : 0u;
        }
        __syncthreads();
        const uint32_t o = span_base + wave_n[wave] + inc - cnt;
        __syncthreads();
        return o;
    };
    for (uint32_t sp = blockIdx.x; sp < nspan; sp += gridDim.x) {
        uint32_t mfound = 0, nfound = 0, mlong = 0, nlong = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t r = sp * SPAN + k * TPB + threadIdx.x;
            bool found = false;
            if (r < nr && !((reached_bits[r >> 5] >> (r & 31u)) & 1u)) {   // (read in place: this wave rewrites the word below)
                const BuQuad q = FN[r];
                uint32_t hitnb = 0xFFFFFFFFu;
                if (q.v[0] != 0xFFFFFFFFu) {
                    const bool h0 = bu_on_level(level_bits, q.v[0]), h1 = q.v[1] != 0xFFFFFFFFu && bu_on_level(level_bits, q.v[1]);
                    hitnb = h0 ? q.v[0] : h1 ? q.v[1] : 0xFFFFFFFFu;
                    if (hitnb == 0xFFFFFFFFu && q.v[2] != 0xFFFFFFFFu) {
                        const bool h2 = bu_on_level(level_bits, q.v[2]), h3 = q.v[3] != 0xFFFFFFFFu && bu_on_level(level_bits, q.v[3]);
                        hitnb = h2 ? q.v[2] : h3 ? q.v[3] : 0xFFFFFFFFu;
                    }
                }
                if (hitnb != 0xFFFFFFFFu) { const uint32_t cand = vid_base + IR[hitnb]; if (cand < y[r]) y[r] = cand; found = true; mfound |= 1u << k; nfound++; }
                else if (q.v[3] != 0xFFFFFFFFu) { mlong |= 1u << k; nlong++; }   // four entries and no hit: there may be more
            }
            const uint64_t b = __ballot(found);   // rows r - lane .. r - lane + 63
            if (lane == 0) {
                const uint32_t w = (sp * SPAN + k * TPB + wave * 64) >> 5;
                next_bits[w] = (uint32_t)b; next_bits[w + 1] = (uint32_t)(b >> 32);
                if (b) { reached_bits[w] |= (uint32_t)b; reached_bits[w + 1] |= (uint32_t)(b >> 32); }
            }
        }
        if (rows_out) {
            uint32_t o = reserve(nfound, rows_n);
            for (uint32_t k = 0; mfound; k++, mfound >>= 1) if (mfound & 1u) rows_out[o++] = sp * SPAN + k * TPB + threadIdx.x;
        }
        uint32_t o = reserve(nlong, long_n);
        for (uint32_t k = 0; mlong; k++, mlong >>= 1) if (mlong & 1u) long_list[o++] = sp * SPAN + k * TPB + threadIdx.x;
    }
}
template <uint32_t LPR>
__global__ void __launch_bounds__(TPB) k_bu_rest_rows(const uint32_t *__restrict__ long_list, const unsigned int *__restrict__ long_n, const uint32_t *__restrict__ R2C,
                                                      const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA, const uint32_t *__restrict__ IR,
                                                      const uint32_t *__restrict__ level_bits, uint32_t vid_base, uint32_t *__restrict__ y,
                                                      uint32_t *__restrict__ rows_out, unsigned int *__restrict__ rows_n, uint32_t *__restrict__ next_bits,
                                                      uint32_t *__restrict__ reached_bits) {
    constexpr uint32_t GPB = TPB / LPR;
    const uint32_t n = *long_n, sub = threadIdx.x & (LPR - 1), gbase = (threadIdx.x & 63u) & ~(LPR - 1);
    for (uint32_t li = blockIdx.x * GPB + threadIdx.x / LPR; li < n; li += gridDim.x * GPB) {
        uint32_t cand = GT_INF;
        const uint32_t r = long_list[li], c = R2C[r];
        for (uint32_t base = JA[c] + BU_PROBE, e1 = JA[c + 1]; base < e1; base += LPR) {
            const uint32_t e = base + sub;
            uint32_t nb = 0; bool hit = false;
            if (e < e1) { nb = IA[e]; hit = bu_on_level(level_bits, nb); }
            const uint32_t m = (uint32_t)(__ballot(hit) >> gbase) & (LPR == 32 ? 0xFFFFFFFFu : ((1u << LPR) - 1u));
            if (m) { cand = vid_base + IR[__shfl(nb, (int)(gbase + (uint32_t)__ffs((int)m) - 1u))]; break; }
        }
        if (sub == 0 && cand != GT_INF) {
            if (cand < y[r]) y[r] = cand;
            if (rows_out) rows_out[atomicAdd(rows_n, 1u)] = r;   // (few rows come this far)
            atomicOr(&next_bits[r >> 5], 1u << (r & 31u)); atomicOr(&reached_bits[r >> 5], 1u << (r & 31u));
        }
    }
}
__global__ void k_bu_maps_root(const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ IV, uint32_t v, uint32_t *__restrict__ level_bits, uint32_t *__restrict__ reached_bits) {
    if (IJ[v] & 1u) { const uint32_t r = IV[v]; level_bits[r >> 5] = 1u << (r & 31u); reached_bits[r >> 5] = 1u << (r & 31u); }
}
// entries of the columns of a frontier list's vertices (how heavy the frontier is)
__global__ void k_list_entries(const uint32_t *__restrict__ list, uint32_t n, const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ JV, const uint32_t *__restrict__ JA,
                               unsigned long long *__restrict__ out) {
    unsigned long long e = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t v = list[i];
        if (IJ[v] & 2u) { const uint32_t c = JV[v]; e += JA[c + 1] - JA[c]; }
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
    __shared__ unsigned long long ws[TPB / 64];
    if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < TPB / 64; w++) e += ws[w]; if (e) atomicAdd(out, e); }
}

// ---- frontier lists (vertices changed by the last apply)
__device__ __forceinline__ uint32_t slot_of_vertex(const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ JV, const uint32_t *__restrict__ xslot, uint32_t v) {
    if (!(IJ[v] & 2u)) return 0xFFFFFFFFu;   // no column: the vertex sends nothing
    const uint32_t c = JV[v];
    return xslot ? xslot[c] : c;
}
__global__ void k_list_reset_x(uint32_t *__restrict__ x, const uint32_t *__restrict__ list, const unsigned int *__restrict__ n_dev,
                               const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ JV, const uint32_t *__restrict__ xslot) {
    const uint32_t n = *n_dev;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t sl = slot_of_vertex(IJ, JV, xslot, list[i]);
        if (sl != 0xFFFFFFFFu) x[sl] = GT_INF;
    }
}
__global__ void k_list_msg(uint32_t *__restrict__ x, const uint32_t *__restrict__ list, const unsigned int *__restrict__ n_dev,
                           const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ JV, const uint32_t *__restrict__ xslot,
                           const uint32_t *__restrict__ s0, uint32_t vid_base, gt_vidmap vm, int kind) {
    const uint32_t n = *n_dev;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t v = list[i], sl = slot_of_vertex(IJ, JV, xslot, v);
        if (sl != 0xFFFFFFFFu) x[sl] = (kind == GT_BFS) ? gt_vid_of(vm, (uint64_t)vid_base + v) : s0[v];   // bfs.h:52-54, sssp.h:44-46, cc.h:38-40
    }
}
// the frontier's columns, messages and entry counts straight from the list (a vertex without a column counts zero entries)
__global__ void k_list_frontier(const uint32_t *__restrict__ list, uint32_t n, const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ JV,
                                const uint32_t *__restrict__ JA, const uint32_t *__restrict__ s0, uint32_t vid_base, gt_vidmap vm, int kind,
                                uint32_t *__restrict__ col, uint32_t *__restrict__ val, uint32_t *__restrict__ deg, unsigned long long *__restrict__ out) {
    unsigned long long e = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t v = list[i];
        uint32_t c = 0, d = 0;
        if (IJ[v] & 2u) { c = JV[v]; d = JA[c + 1] - JA[c]; }
        col[i] = c; deg[i] = d; val[i] = (kind == GT_BFS) ? gt_vid_of(vm, (uint64_t)vid_base + v) : s0[v];
        e += d;
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
    __shared__ unsigned long long part[TPB / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) { for (int w = 1; w < TPB / 64; w++) e += part[w]; if (e) atomicAdd(&out[1], e); }
}

// ---- the tail of a min program in ONE launch (one rank, converge mode): while the frontier is a short list an iteration is
// ~20 small launches and a host round trip, 0.17-0.2 ms for microseconds of work. One workgroup runs whole iterations instead --
// messages of the list's vertices, the SpMSpV over their columns (atomicMin on y, rows de-duplicated through the mark bits),
// apply of the rows it lowered, the next list -- and loops on the device until nothing changes, or the list or its columns'
// entries outgrow what one workgroup should do (it then leaves everything as an ordinary iteration would find it).
constexpr uint32_t TAIL_N = 8192;          // longest list the kernel takes (LDS: 3 x 32 KiB)
constexpr int TAIL_THREADS = 1024;
struct TailOut { uint32_t iterations, active, status, cur; unsigned long long changed; };
enum { TAIL_CONVERGED = 0, TAIL_LIST_GREW = 1, TAIL_ENTRIES = 2, TAIL_MAX_ITERATIONS = 3 };
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // past this CU's L1

template <bool BFS, bool WEIGHTED>
__global__ void __launch_bounds__(TAIL_THREADS) k_tail(uint32_t *__restrict__ list0, uint32_t *__restrict__ list1, unsigned int *__restrict__ d_fl, int cur, uint32_t list_cap,
                                                       uint32_t n_cap, uint32_t ent_cap, const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ JV,
                                                       const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA, const uint32_t *__restrict__ A,
                                                       const uint32_t *__restrict__ IR, uint32_t *__restrict__ y, uint32_t *__restrict__ s0, uint32_t *__restrict__ s1,
                                                       uint8_t *__restrict__ C, uint32_t *__restrict__ marks, uint32_t *__restrict__ rows, uint32_t iteration,
                                                       uint32_t vid_base, gt_vidmap vm, uint32_t max_it, TailOut *__restrict__ out) {
    __shared__ uint32_t off[TAIL_N + 1];   // degrees, then exclusive entry offsets of the list's columns
    __shared__ uint32_t start[TAIL_N];     // first entry of each column
    __shared__ uint32_t val[TAIL_N];       // its message
    __shared__ uint32_t wsum[TAIL_THREADS / 64];
    __shared__ uint32_t s_rows, s_next;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t n = d_fl[cur], done = 0, status = TAIL_CONVERGED;
    unsigned long long changed = 0;
    if (tid == 0) { s_rows = 0; s_next = 0; }
    __syncthreads();
    for (;;) {
        if (n > n_cap) { status = TAIL_LIST_GREW; break; }
        if (done == max_it) { status = TAIL_MAX_ITERATIONS; break; }
        uint32_t *__restrict__ L = cur ? list1 : list0, *__restrict__ Ln = cur ? list0 : list1;
        for (uint32_t i = tid; i < TAIL_N; i += TAIL_THREADS) {
            uint32_t d = 0, st = 0, m = 0;
            if (i < n) {
                const uint32_t v = L[i];
                if (IJ[v] & 2u) { const uint32_t c = JV[v]; st = JA[c]; d = JA[c + 1] - st; }
                m = BFS ? gt_vid_of(vm, (uint64_t)vid_base + v) : ld_agent(s0 + v);   // bfs.h:52-54, sssp.h:44-46, cc.h:38-40
            }
            off[i] = d; start[i] = st; val[i] = m;
        }
        __syncthreads();
        // exclusive scan of the degrees: eight consecutive items per thread, a shuffle scan per wave, the waves' totals through LDS
        uint32_t loc[TAIL_N / TAIL_THREADS], sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < TAIL_N / TAIL_THREADS; j++) { loc[j] = sum; sum += off[tid * (TAIL_N / TAIL_THREADS) + j]; }
        uint32_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (uint32_t w = 0; w < TAIL_THREADS / 64; w++) { const uint32_t t = wsum[w]; if (w < wave) base += t; total += t; }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < TAIL_N / TAIL_THREADS; j++) off[tid * (TAIL_N / TAIL_THREADS) + j] = base + inc - sum + loc[j];
        if (tid == 0) off[TAIL_N] = total;
        __syncthreads();
        if (total > ent_cap) { status = TAIL_ENTRIES; break; }   // nothing has been touched yet: the ordinary path takes this iteration
        for (uint32_t i = tid; i < n; i += TAIL_THREADS) C[L[i]] = 0;   // the flags of the vertices that were active (apply sets the new ones)
        // SpMSpV: one thread per entry of the list's columns (vp:1475-1489)
        for (uint32_t t = tid; t < total; t += TAIL_THREADS) {
            uint32_t lo = 0, hi = n;   // last i with off[i] <= t
            while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (off[mid] <= t) lo = mid; else hi = mid; }
            const uint32_t e = start[lo] + (t - off[lo]);
            const uint32_t r = IA[e];
            const uint32_t m = WEIGHTED ? val[lo] + A[e] : val[lo];
            if (m < y[r]) {   // (a stale, higher y only costs an atomic)
                const uint32_t old = atomicMin(&y[r], m);
                if (m < old) {
                    const uint32_t bit = 1u << (r & 31u);
                    if (!(atomicOr(&marks[r >> 5], bit) & bit)) rows[atomicAdd(&s_rows, 1u)] = r;   // the first to lower the row lists it
                }
            }
        }
        __threadfence();
        __syncthreads();
        const uint32_t nrows = s_rows;
        // apply over the rows that were lowered (the only ones that can change), the next list
        for (uint32_t i = tid; i < nrows; i += TAIL_THREADS) {
            const uint32_t r = rows[i];
            atomicAnd(&marks[r >> 5], ~(1u << (r & 31u)));
            const uint32_t v = IR[r], yv = ld_agent(y + r);
            bool c;
            if (BFS) { c = ld_agent(s1 + v) == GT_INF && yv != GT_INF; if (c) { s1[v] = iteration + done + 1; s0[v] = yv; } }   // bfs.h:65-77
            else { const uint32_t tmp = ld_agent(s0 + v), nv = yv < tmp ? yv : tmp; c = nv != tmp; if (c) s0[v] = nv; }       // sssp.h:57-65, cc.h:51-55
            C[v] = c;
            if (c) { const uint32_t kx = atomicAdd(&s_next, 1u); if (kx < list_cap) Ln[kx] = v; }
        }
        __threadfence();
        __syncthreads();
        n = s_next; changed += n; done++; cur ^= 1;
        __syncthreads();
        if (tid == 0) { s_rows = 0; s_next = 0; }
        __syncthreads();
        if (n == 0) { status = TAIL_CONVERGED; break; }
    }
    if (tid == 0) {
        if (done) { d_fl[cur] = n; d_fl[cur ^ 1] = 0; d_fl[2] = 0; }   // (nothing ran: the lists are as the caller left them)
        out->iterations = done; out->active = n; out->status = status; out->cur = (uint32_t)cur; out->changed = changed;
    }
}

}  // namespace

// Called after an apply() that left a short list (converge mode, one rank): runs iterations on the device until the program
// converges or the frontier outgrows the kernel. *converged: the last apply activated nothing (the caller finishes the run).
int gt_tail_try(gt_program *p, hipStream_t s, bool *converged, uint32_t *iterations_run) {
    *converged = false; *iterations_run = 0;
    const gt_graph *g = p->g;
    const int enabled = gt_cfg(p, "GRAPHTAP_TAIL_KERNEL") ? atoi(gt_cfg(p, "GRAPHTAP_TAIL_KERNEL")) : 1;   // 0: never
    const uint32_t n_cap = std::min<uint32_t>(TAIL_N, gt_cfg(p, "GRAPHTAP_TAIL_LIST") ? (uint32_t)atoi(gt_cfg(p, "GRAPHTAP_TAIL_LIST")) : 4096u);   // longest list it takes
    const uint32_t ent_cap = gt_cfg(p, "GRAPHTAP_TAIL_ENTRIES") ? (uint32_t)atoi(gt_cfg(p, "GRAPHTAP_TAIL_ENTRIES")) : (1u << 17);          // most entries of its columns
    const char *senv = gt_cfg(p, "GRAPHTAP_SPMSPV");
    if (senv && atoi(senv) == 0) return GT_OK;   // the sparse paths are switched off: this is one
    if (!enabled || !p->fl_enabled || !p->fl_cur_valid || gt_has_exchange(g) || p->fl_cur_n == 0 || p->fl_cur_n > n_cap || !p->d_tail) return GT_OK;
    if (p->semiring != GT_MIN_U32 && p->semiring != GT_MINPLUS_U32) return GT_OK;
    const uint32_t vb = g->info.rank * g->info.tile_height;
    const bool bfs = p->prm.kind == GT_BFS, w = p->semiring == GT_MINPLUS_U32;
    TailOut *out = (TailOut *)p->d_tail;
#define GT_TAIL_LAUNCH(B, W)                                                                                                              \
    k_tail<B, W><<<1, TAIL_THREADS, 0, s>>>(p->fl_v[0], p->fl_v[1], p->d_fl, p->fl_cur, p->fl_cap, std::min(n_cap, p->fl_cap), ent_cap, g->IJ, g->JV, g->JA, g->IA, \
                                            g->A, g->IR, (uint32_t *)p->y, p->s0, p->s1, p->C, (uint32_t *)p->row_mark, p->fl_rows, p->iteration, vb,        \
                                            gt_vidmap_of(g), 1u << 20, out)
    if (bfs) GT_TAIL_LAUNCH(true, false); else if (w) GT_TAIL_LAUNCH(false, true); else GT_TAIL_LAUNCH(false, false);
#undef GT_TAIL_LAUNCH
    GT_HIP(hipGetLastError());
    TailOut h{};
    { int st = gt_read_back(p, &h, out, sizeof(h), s); if (st != GT_OK) return st; }
    if (h.iterations == 0) return GT_OK;   // too many entries in the very first list: nothing was touched
    p->iteration += h.iterations;
    p->spmspv_iters += h.iterations; p->list_iters += h.iterations; p->tail_iters += h.iterations;
    p->fl_prev_valid = false; p->fl_prev_n = 0;   // the messages of the lists in between were never written: the next messenger rewrites x
    p->x_stale = true; p->x_fresh = false; p->x_deferred = false;
    p->fl_cur = (int)h.cur; p->fl_cur_n = h.active; p->fl_cur_valid = h.active <= p->fl_cap;
    p->fl_rows_valid = false;
    p->last_active = h.active;
    if (p->prm.kind == GT_BFS) p->bfs_settled += h.changed;
    *iterations_run = h.iterations;
    *converged = (h.status == TAIL_CONVERGED);
    return GT_OK;
}

namespace {
}  // namespace

// buffers of the sparse path, sized for frontiers of up to `nact` columns (called at initialize() so that no allocation
// falls into the iteration loop)
int gt_spmspv_reserve(gt_program *p, uint32_t nact) {
    if (!p->d_frontier) { GT_HIP(hipMalloc((void **)&p->d_frontier, 4 * sizeof(unsigned long long))); p->spmspv_allocs++; }
    if (p->fr_cap < nact + 1) {
        p->spmspv_allocs += 3;
        for (void *q : {(void *)p->fr_col, (void *)p->fr_val, (void *)p->fr_off}) if (q) GT_HIP(hipFree(q));
        p->fr_col = p->fr_val = p->fr_off = nullptr; p->fr_cap = 0;
        const uint64_t cap = (uint64_t)nact + nact / 2 + 1024;
        GT_HIP(hipMalloc((void **)&p->fr_col, cap * 4)); GT_HIP(hipMalloc((void **)&p->fr_val, cap * 4)); GT_HIP(hipMalloc((void **)&p->fr_off, cap * 4));
        p->fr_cap = (uint32_t)cap;
    }
    size_t tb = 0;
    GT_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, p->fr_off, p->fr_off, p->fr_cap, (hipStream_t)0));
    if (tb > p->fr_tmp_bytes) { if (p->fr_tmp) GT_HIP(hipFree(p->fr_tmp)); p->fr_tmp = nullptr; GT_HIP(hipMalloc(&p->fr_tmp, tb)); p->fr_tmp_bytes = tb; p->spmspv_allocs++; }
    return GT_OK;
}

int gt_bu_first_neighbours(const gt_graph *g, uint32_t *FN, hipStream_t s) {
    const uint32_t nr = g->info.nnzrows;
    if (!nr) return GT_OK;
    k_bu_first_neighbours<<<(unsigned)std::min<uint64_t>(((uint64_t)nr + TPB - 1) / TPB, 256u * 64u), TPB, 0, s>>>(g->R2C, g->JA, g->IA, nr, reinterpret_cast<BuQuad *>(FN));
    GT_HIP(hipGetLastError());
    return GT_OK;
}

static int spmspv_from_list(gt_program *p, hipStream_t s, bool force, bool *done);

// CC, iteration 0, on a symmetric graph held by one rank: EVERY vertex is active and sends its own id (cc.h:33-40), so what the
// min-combiner leaves in y[r] is the smallest id among r's neighbours -- and the neighbours of r are the entries of the same
// vertex's column, stored by ascending row = ascending id: the column's FIRST entry. One load chain per row instead of a pass over
// every entry (R-MAT-26 symmetrised: 2.10 G entries, 2.3 ms). A self-loop is an entry like any other, as in the sweep.
__global__ void k_cc_first_neighbour(const uint32_t *__restrict__ R2C, const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA, const uint32_t *__restrict__ IR,
                                     uint32_t nr, uint32_t vid_base, uint32_t *__restrict__ y) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += gridDim.x * blockDim.x) {
        const uint32_t c = R2C[r];
        if (c == 0xFFFFFFFFu) continue;
        const uint32_t e0 = JA[c];
        if (e0 == JA[c + 1]) continue;
        const uint32_t m = vid_base + IR[IA[e0]];
        if (m < y[r]) y[r] = m;
    }
}
// (scatter_gather asks too: the step reads no messages, so iteration 0's messenger -- a pass over every slot of x -- is left out;
// the full apply that follows writes every message anew, combine runs the messenger after all if the step is not taken)
// scatter_gather asks: the current frontier is a list that can qualify for the SpMSpV, which takes its messages from the list and
// never looks at x. The messenger is then left to combine, which runs it (all of x: it is stale by then) only if the streaming
// pass is taken after all. BFS R-MAT-26: the 0.2-ms pass over 33 M slots before the first list iteration after the bottom-up
// steps; CC: 0.18 ms for resetting the 3.7 M slots of the previous frontier before a 13 K-vertex one.
bool gt_list_spmspv_likely(const gt_program *p) {
    const char *senv = gt_cfg(p, "GRAPHTAP_SPMSPV");
    if (gt_has_exchange(p->g) || !p->fl_enabled || !p->fl_cur_valid || (senv && atoi(senv) == 0)) return false;
    return gt_frontier_list_worth(p, p->fl_cur_n);
}
bool gt_cc_first_likely(const gt_program *p) {
    const gt_graph *g = p->g;
    const char *e = gt_cfg(p, "GRAPHTAP_CC_FIRST");   // 0: iteration 0 sweeps like every other
    const char *senv = gt_cfg(p, "GRAPHTAP_SPMSPV");  // 0: no sparse path of any kind
    return p->prm.kind == GT_CC && p->iteration == 0 && !p->converged && !g->flags.directed && !gt_has_exchange(g) && g->info.nnzrows != 0 && g->info.nnz_local != 0 &&
           !(e && atoi(e) == 0) && !(senv && atoi(senv) == 0);
}
static int cc_first_iteration_try(gt_program *p, hipStream_t s, bool *done) {
    const gt_graph *g = p->g;
    if (!gt_cc_first_likely(p)) return GT_OK;
    const uint32_t nr = g->info.nnzrows;
    k_cc_first_neighbour<<<(unsigned)std::min<uint64_t>(((uint64_t)nr + TPB - 1) / TPB, 256u * 64u), TPB, 0, s>>>(g->R2C, g->JA, g->IA, g->IR, nr, g->info.rank * g->info.tile_height,
                                                                                                                 (uint32_t *)p->y);
    GT_HIP(hipGetLastError());
    p->spmspv_iters++;
    p->fl_rows_valid = false;   // every row may change: the full apply
    *done = true;
    return GT_OK;
}

// Is a list of the n vertices an apply changed of any use on one rank? Only a frontier that can qualify for the SpMSpV (the
// rule at the head of spmspv_from_list) or for the tail kernel is worth collecting: the messages of a large frontier were written
// by the full apply itself, and its pass streams. Collecting 5 M changed vertices from the flags took 0.08 ms per iteration of
// SSSP on R-MAT-24 (+ 0.02-0.04 for looking at the list again), four times in a 3.4-ms run.
bool gt_frontier_list_worth(const gt_program *p, uint64_t n) {
    const char *env = gt_cfg(p, "GRAPHTAP_SPMSPV");
    if (gt_has_exchange(p->g) || (env && atoi(env) == 1)) return true;   // several ranks: the list travels as pairs; forced SpMSpV: any list
    const uint64_t frac0 = gt_cfg(p, "GRAPHTAP_SPMSPV_FRACTION") ? (uint64_t)atoll(gt_cfg(p, "GRAPHTAP_SPMSPV_FRACTION")) : 32;
    return n <= 4096 || n * 8 <= p->g->info.nnz_local / frac0;
}

// A bottom-up step instead of the push sweep (Beamer's direction switch, with the reference's labels: every neighbour is looked
// at, the minimum id wins). Only for BFS on a symmetric graph (built with directed = false) on one rank, and only when fewer
// rows are unreached than vertices are active and the unreached rows' columns hold few entries (counted exactly first).
bool gt_bfs_bottom_up_likely(const gt_program *p) {
    const gt_graph *g = p->g;
    const char *menv = gt_cfg(p, "GRAPHTAP_BFS_BOTTOM_UP");
    const int mode = menv ? atoi(menv) : -1;   // 0 never, 1 whenever possible, unset: by size
    const char *senv = gt_cfg(p, "GRAPHTAP_SPMSPV");
    if (mode == 0 || (senv && atoi(senv) == 0) || !p->fl_enabled || !p->bu_rows || p->prm.kind != GT_BFS || g->flags.directed || g->info.nnzrows == 0) return false;
    const uint64_t nr = g->info.nnzrows, unreached = nr > p->bfs_settled ? nr - p->bfs_settled : 0;
    if (mode == 1 || p->last_active == ~0ull) return mode == 1;
    // few rows left (the classic switch), or a frontier large enough that a row's first probes are likely to hit (the step stops
    // at a row's first hit; the collecting pass then decides with exact counts)
    return unreached < p->last_active || (p->last_active > 65536 && p->last_active * 64 >= nr);
}

int gt_bu_maps_init(gt_program *p, hipStream_t s) {
    const gt_graph *g = p->g;
    const char *e = getenv("GRAPHTAP_BFS_BU_MAPS");   // 0: every bottom-up step collects the unreached rows and the level from the vertex states
    p->bu_maps_valid = false;
    if (!p->bu_first || !p->bu_reached || !p->bu_next || (e && atoi(e) == 0)) return GT_OK;
    for (uint32_t *m : {p->bu_bits, p->bu_next, p->bu_reached}) GT_HIP(hipMemsetAsync(m, 0, (size_t)p->bu_words * 4, s));
    p->bu_step_emitted = false;
    if (p->root_here) k_bu_maps_root<<<1, 1, 0, s>>>(g->IJ, g->IV, p->root_local, p->bu_bits, p->bu_reached);
    GT_HIP(hipGetLastError());
    p->bu_maps_valid = true;
    return GT_OK;
}

// the bottom-up step from the row bitmaps (no collecting pass). The decision uses the host's counts: rows still unreached
// (nnzrows - vertices reached so far) and, for a frontier that is a list, the entries of its columns.
static int bfs_bottom_up_maps_try(gt_program *p, hipStream_t s, bool *done) {
    const gt_graph *g = p->g;
    const char *menv = gt_cfg(p, "GRAPHTAP_BFS_BOTTOM_UP");
    const int mode = menv ? atoi(menv) : -1;
    const bool dbg = getenv("GRAPHTAP_PB_STATS") != nullptr;
    const uint64_t nr = g->info.nnzrows, nnz = g->info.nnz_local;
    const uint64_t unreached = nr > p->bfs_settled ? nr - p->bfs_settled : 0;
    constexpr uint32_t LPR = 16;
    bool take = mode == 1 || (p->last_active != ~0ull && unreached < p->last_active);   // few rows left: the classic switch
    uint64_t fe = 0;
    if (!take) {   // a heavy frontier: a row's first probes are likely to hit
        if (p->fl_cur_valid && p->fl_cur_n) {
            GT_HIP(hipMemsetAsync(p->d_frontier, 0, 4 * sizeof(unsigned long long), s));
            k_list_entries<<<(unsigned)std::min<uint64_t>(((uint64_t)p->fl_cur_n + TPB - 1) / TPB, 4096), TPB, 0, s>>>(p->fl_v[p->fl_cur], p->fl_cur_n, g->IJ, g->JV, g->JA, p->d_frontier + 1);
            unsigned long long h = 0;
            { int st = gt_read_back(p, &h, p->d_frontier + 1, sizeof(h), s); if (st != GT_OK) return st; }
            fe = h;
        } else if (p->last_active != ~0ull && p->last_active * 8 >= nr) fe = nnz / 2;   // not even a list: most of the graph is on the level
        const uint64_t per_row = fe ? std::max<uint64_t>(LPR, nnz / fe) : ~0ull;
        take = fe != 0 && per_row < nnz && unreached * per_row * 2 <= nnz;
    }
    if (dbg) fprintf(stderr, "[bfs] iteration %u: %llu rows unreached, %llu vertices active, the frontier's columns hold %llu entries: %s\n", p->iteration,
                     (unsigned long long)unreached, (unsigned long long)p->last_active, (unsigned long long)fe, take ? "bottom-up step (row bitmaps)" : "push sweep");
    if (!take) return GT_OK;
    const bool rows_list = unreached * 8 <= nr;   // else the full apply (the row-list apply of 31.8 M rows took 1.5 ms, the full one 0.27)
    GT_HIP(hipMemsetAsync(p->d_fl + 2, 0, sizeof(unsigned int), s));
    GT_HIP(hipMemsetAsync(p->d_frontier + 3, 0, sizeof(unsigned long long), s));
    unsigned int *long_n = (unsigned int *)(p->d_frontier + 3);
    uint32_t *rows_out = rows_list ? p->fl_rows : nullptr;
    const uint32_t vb = g->info.rank * g->info.tile_height;
    k_bu_probe_all<<<(unsigned)std::min<uint64_t>((nr + 2047) / 2048, 256u * 64u), TPB, 0, s>>>(p->bu_reached, p->bu_bits, p->bu_next, reinterpret_cast<const BuQuad *>(p->bu_first), g->IR,
                                                                                                (uint32_t)nr, vb, (uint32_t *)p->y, rows_out, p->d_fl + 2, p->bu_long, long_n);
    k_bu_rest_rows<LPR><<<4096, TPB, 0, s>>>(p->bu_long, long_n, g->R2C, g->JA, g->IA, g->IR, p->bu_bits, vb, (uint32_t *)p->y, rows_out, p->d_fl + 2, p->bu_next, p->bu_reached);
    GT_HIP(hipGetLastError());
    std::swap(p->bu_bits, p->bu_next);   // the rows that found a parent are the next level; the apply of this iteration leaves the bitmaps alone
    p->bu_step_emitted = true;
    p->bottom_up_iters++; p->spmspv_iters++;
    p->fl_rows_valid = rows_list;
    *done = true;
    return GT_OK;
}

static int bfs_bottom_up_try(gt_program *p, hipStream_t s, bool *done) {
    const gt_graph *g = p->g;
    const char *menv = gt_cfg(p, "GRAPHTAP_BFS_BOTTOM_UP");
    const int mode = menv ? atoi(menv) : -1;
    const char *eenv = getenv("GRAPHTAP_BFS_BU_EARLY");   // 0: every neighbour is looked at (A/B, tests)
    const bool early = !(eenv && atoi(eenv) == 0);
    const bool dbg = getenv("GRAPHTAP_PB_STATS") != nullptr;
    if (!gt_bfs_bottom_up_likely(p)) return GT_OK;
    if (p->bu_maps_valid && early) return bfs_bottom_up_maps_try(p, s, done);
    const uint32_t nr = g->info.nnzrows;
    if (dbg) fprintf(stderr, "[bfs] iteration %u: %llu rows unreached (estimate), %llu vertices active\n", p->iteration,
                     (unsigned long long)(nr > p->bfs_settled ? nr - p->bfs_settled : 0), (unsigned long long)p->last_active);
    GT_HIP(hipMemsetAsync(p->d_fl + 3, 0, sizeof(unsigned int), s));
    GT_HIP(hipMemsetAsync(p->d_frontier, 0, 4 * sizeof(unsigned long long), s));
    k_bu_collect<<<(unsigned)std::min<uint64_t>(((uint64_t)nr + 4095) / 4096, 4096), TPB, 0, s>>>(g->IR, g->R2C, g->JA, p->s1, p->iteration, nr, p->bu_rows, p->d_fl + 3,
                                                                                                  p->d_frontier + 1, p->bu_bits);
    unsigned long long cnt[2] = {0, 0}; unsigned int n = 0;   // entries of the unreached rows' columns, of the frontier's columns
    GT_HIP(hipMemcpyAsync(cnt, p->d_frontier + 1, sizeof(cnt), hipMemcpyDeviceToHost, s));
    GT_HIP(hipMemcpyAsync(&n, p->d_fl + 3, sizeof(n), hipMemcpyDeviceToHost, s));
    { int st = gt_stream_wait_deadline(s, "the bottom-up collecting pass"); if (st != GT_OK) return st; }
    const uint64_t nnz = g->info.nnz_local, entries = cnt[0], fe = cnt[1];
    // Looking at every entry of the unreached rows pays when they are few (nnz / 8: the push sweep costs a pass over the active
    // windows). With early exit a row probes ~nnz / fe entries before its first hit (fe = entries of the frontier's columns = the
    // chance that a neighbour is on the level, entry-weighted), 16 at a time: taken when that is well under a pass.
    constexpr uint32_t LPR = 16;
    const uint64_t per_row = fe ? std::max<uint64_t>(LPR, nnz / fe) : ~0ull;
    const bool few = entries <= nnz / 8, heavy = early && fe != 0 && per_row < nnz && (uint64_t)n * per_row * 2 <= nnz;
    if (dbg) fprintf(stderr, "[bfs] iteration %u: %u unreached rows hold %llu entries, the frontier's columns %llu: %s\n", p->iteration, n, (unsigned long long)entries, (unsigned long long)fe,
                     (mode == 1 || few || heavy) ? (early ? "bottom-up step (first hit)" : "bottom-up step") : "push sweep");
    if (mode != 1 && !few && !heavy) return GT_OK;   // the push sweep is the cheaper one
    // the rows that found a parent go to apply as a list (position for position, ~0u for the others) -- unless they are a large
    // share of all rows: the row-list apply of 31.8 M rows took 1.5 ms where the full apply takes 0.27
    const bool rows_list = !early || (uint64_t)n * 8 <= nr;
    if (rows_list) GT_HIP(hipMemcpyAsync(p->d_fl + 2, p->d_fl + 3, sizeof(unsigned int), hipMemcpyDeviceToDevice, s));   // one slot of fl_rows per unreached row
    if (n) {
        uint32_t *rows_out = rows_list ? p->fl_rows : nullptr;
        const uint32_t vb = g->info.rank * g->info.tile_height;
        if (early) {
            unsigned int *long_n = (unsigned int *)(p->d_frontier + 3);   // (zeroed with the counters above)
            k_bu_first_probe<<<(unsigned)std::min<uint64_t>(((uint64_t)n + 2047) / 2048, 256u * 64u), TPB, 0, s>>>(p->bu_rows, p->d_fl + 3, reinterpret_cast<const BuQuad *>(p->bu_first), g->IR, p->bu_bits, vb,
                                                                                                                      (uint32_t *)p->y, rows_out, p->bu_long, long_n);
            k_bu_first_rest<LPR><<<4096, TPB, 0, s>>>(p->bu_long, long_n, p->bu_rows, g->R2C, g->JA, g->IA, g->IR, p->bu_bits, vb, (uint32_t *)p->y, rows_out);
        } else {
            const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)n + TPB / 16 - 1) / (TPB / 16), 256u * 64u);
            k_bu_step<<<grid, TPB, 0, s>>>(p->bu_rows, p->d_fl + 3, g->R2C, g->JA, g->IA, g->IR, p->s1, p->iteration, vb, gt_vidmap_of(g), (uint32_t *)p->y, rows_out);
        }
        GT_HIP(hipGetLastError());
    }
    p->bottom_up_iters++; p->spmspv_iters++;
    p->fl_rows_valid = rows_list;
    *done = true;
    return GT_OK;
}

int gt_spmspv_try(gt_program *p, hipStream_t s, bool *done) {
    *done = false;
    const gt_graph *g = p->g;
    if (p->semiring != GT_MIN_U32 && p->semiring != GT_MINPLUS_U32) return GT_OK;
    const char *env = gt_cfg(p, "GRAPHTAP_SPMSPV");               // "0": never, "1": whenever there is a frontier, unset: by size
    if (env && atoi(env) == 0) return GT_OK;
    const uint64_t nnz = g->info.nnz_local;
    if (nnz == 0) return GT_OK;
    const bool force = env && atoi(env) == 1;
    p->fl_rows_valid = false;
    { int st = cc_first_iteration_try(p, s, done); if (st != GT_OK || *done) return st; }
    // BFS with a large frontier and few unreached rows: the bottom-up step first (R-MAT-26, 3.7 M active columns: 0.24 ms against
    // 0.39 ms for the SpMSpV from the list); small frontiers: the list first
    const bool bottom_up_first = p->last_active != ~0ull && p->last_active > 65536;
    if (bottom_up_first) { int st = bfs_bottom_up_try(p, s, done); if (st != GT_OK || *done) return st; }
    // (under the exchange layout the frontier is what the peers sent, not this rank's list: dist.hip runs that SpMSpV from the pairs)
    const bool own_list = p->fl_enabled && p->fl_cur_valid && !gt_has_exchange(g);
    if (own_list) {
        int st = spmspv_from_list(p, s, force, done);
        if (st != GT_OK || *done) return st;
    }
    if (!bottom_up_first) { int st = bfs_bottom_up_try(p, s, done); if (st != GT_OK || *done) return st; }
    if (own_list) return GT_OK;
    // counting the frontier costs a pass over x and a device round trip: only worth it when the previous apply() (whose
    // count the converge-mode driver reads anyway) activated few vertices
    static const uint64_t max_active = getenv("GRAPHTAP_SPMSPV_MAX_ACTIVE") ? (uint64_t)atoll(getenv("GRAPHTAP_SPMSPV_MAX_ACTIVE")) : 16384;
    if (!force && p->last_active > max_active) return GT_OK;
    if (p->x_deferred) { int st = gt_min_messenger(p); if (st != GT_OK) return st; }   // this path scans x
    if (!p->d_frontier) { int st = gt_spmspv_reserve(p, (uint32_t)std::min<uint64_t>(p->x_elems, 1u << 20)); if (st != GT_OK) return st; }
    GT_HIP(hipMemsetAsync(p->d_frontier, 0, 4 * sizeof(unsigned long long), s));
    const uint32_t x_len = (uint32_t)p->x_elems;
    const unsigned grid = (unsigned)std::min<uint64_t>((x_len + TPB - 1) / TPB + 1, 4096);
    k_frontier_count<<<grid, TPB, 0, s>>>((const uint32_t *)p->x, x_len, g->xcol, g->JA, p->d_frontier);
    unsigned long long h[2] = {0, 0};
    GT_HIP(hipMemcpyAsync(h, p->d_frontier, sizeof(h), hipMemcpyDeviceToHost, s));
    { int st = gt_stream_wait_deadline(s, "the frontier count"); if (st != GT_OK) return st; }
    if (h[0] == 0 || h[1] == 0) { *done = true; return GT_OK; }  // empty frontier (or active columns without entries here): y keeps its running minima
    // The reference switches at 0.6 of the columns (vp:769). Here the streaming pass already skips every window without an
    // active column (pb.hip), so the frontier-driven kernels only pay for small frontiers; this scan-based entry is what is left
    // when no frontier list exists (several ranks without slices, lists switched off): at most nnz / 1024 entries, and the
    // frontier is only counted when the previous apply activated <= 16 384 vertices.
    const uint64_t frac = gt_cfg(p, "GRAPHTAP_SPMSPV_FRACTION") ? (uint64_t)atoll(gt_cfg(p, "GRAPHTAP_SPMSPV_FRACTION")) : 1024;
    if (!force && h[1] > nnz / frac) return GT_OK;
    GT_REQUIRE(h[0] < 0xFFFFFFFFull && h[1] < (1ull << 40), GT_ERR_UNSUPPORTED, "frontier too large for the sparse path");
    const uint32_t nact = (uint32_t)h[0];
    if (p->fr_cap < nact + 1) { int st = gt_spmspv_reserve(p, nact); if (st != GT_OK) return st; }
    GT_REQUIRE(h[1] < 0xFFFFFFFFull, GT_ERR_UNSUPPORTED, "frontier entry offsets exceed 32 bits");
    unsigned int *cursor = (unsigned int *)(p->d_frontier + 2);
    k_frontier_list<<<grid, TPB, 0, s>>>((const uint32_t *)p->x, x_len, g->xcol, g->JA, cursor, p->fr_col, p->fr_val, p->fr_off);
    {   // degrees -> exclusive offsets, in place
        size_t tb = 0;
        GT_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, p->fr_off, p->fr_off, nact, s));
        if (tb > p->fr_tmp_bytes) { if (p->fr_tmp) GT_HIP(hipFree(p->fr_tmp)); p->fr_tmp = nullptr; GT_HIP(hipMalloc(&p->fr_tmp, tb)); p->fr_tmp_bytes = tb; p->spmspv_allocs++; }
        GT_HIP(hipcub::DeviceScan::ExclusiveSum(p->fr_tmp, tb, p->fr_off, p->fr_off, nact, s));
    }
    const unsigned g2 = (unsigned)std::min<uint64_t>((h[1] + TPB - 1) / TPB, 256u * 64u);
    if (p->semiring == GT_MINPLUS_U32)
        k_spmspv_min<true, false><<<g2, TPB, 0, s>>>(p->fr_col, p->fr_val, p->fr_off, nact, h[1], nullptr, g->JA, g->IA, g->A, (uint32_t *)p->y, nullptr);
    else
        k_spmspv_min<false, false><<<g2, TPB, 0, s>>>(p->fr_col, p->fr_val, p->fr_off, nact, h[1], nullptr, g->JA, g->IA, nullptr, (uint32_t *)p->y, nullptr);
    GT_HIP(hipGetLastError());
    p->spmspv_iters++;
    *done = true;
    return GT_OK;
}

// The same from the frontier LIST (gt_internal.h): no pass over x; the rows it lowers are left in fl_rows for apply().
// A frontier of ONE vertex (iteration 0 of BFS / SSSP: the root, whose column holds a million entries on R-MAT-26): its column's entries
// straight into y, the rows it lowers straight into the row list -- one launch instead of the eight of the general path (list ->
// frontier arrays, short / long split, scan, two kernels, bitmap -> rows: 0.16 ms).
template <bool WEIGHTED>
__global__ void __launch_bounds__(TPB) k_spmspv_one(const uint32_t *__restrict__ list, const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ JV, const uint32_t *__restrict__ JA,
                                                    const uint32_t *__restrict__ IA, const uint32_t *__restrict__ A, const uint32_t *__restrict__ s0, uint32_t vid_base, gt_vidmap vm,
                                                    int kind, uint32_t *__restrict__ y, uint32_t *__restrict__ rows_out, unsigned int *__restrict__ rows_n) {
    __shared__ unsigned wave_n[TPB / 64];
    __shared__ unsigned base;
    const uint32_t v = list[0];
    if (!(IJ[v] & 2u)) return;   // no column: the vertex sends nothing
    const uint32_t c = JV[v], e0 = JA[c], e1 = JA[c + 1];
    const uint32_t m = (kind == GT_BFS) ? gt_vid_of(vm, (uint64_t)vid_base + v) : s0[v];   // bfs.h:52-54, sssp.h:44-46, cc.h:38-40
    if (m == GT_INF) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n = e1 - e0, n_round = (n + TPB - 1) / TPB * TPB;
    for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < n_round; i += gridDim.x * TPB) {
        bool low = false; uint32_t r = 0;
        if (i < n) { r = IA[e0 + i]; const uint32_t val = WEIGHTED ? m + A[e0 + i] : m; low = val < atomicMin(&y[r], val); }
        const uint64_t b = __ballot(low);
        if (lane == 0) wave_n[wave] = (unsigned)__popcll((unsigned long long)b);
        __syncthreads();
        if (threadIdx.x == 0) { unsigned t = 0; for (int w = 0; w < TPB / 64; w++) { const unsigned k = wave_n[w]; wave_n[w] = t; t += k; } base = t ? atomicAdd(rows_n, t) : 0u; }
        __syncthreads();
        if (low) rows_out[base + wave_n[wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = r;
        __syncthreads();
    }
}

static int spmspv_from_list(gt_program *p, hipStream_t s, bool force, bool *done) {
    const gt_graph *g = p->g;
    const uint32_t nact = p->fl_cur_n;
    if (nact == 1 && !(getenv("GRAPHTAP_SPMSPV_ONE") && atoi(getenv("GRAPHTAP_SPMSPV_ONE")) == 0)) {
        GT_HIP(hipMemsetAsync(p->d_fl + 2, 0, sizeof(unsigned int), s));
        const uint32_t vb = g->info.rank * g->info.tile_height;
        if (p->semiring == GT_MINPLUS_U32) k_spmspv_one<true><<<1024, TPB, 0, s>>>(p->fl_v[p->fl_cur], g->IJ, g->JV, g->JA, g->IA, g->A, p->s0, vb, gt_vidmap_of(g), p->prm.kind, (uint32_t *)p->y, p->fl_rows, p->d_fl + 2);
        else k_spmspv_one<false><<<1024, TPB, 0, s>>>(p->fl_v[p->fl_cur], g->IJ, g->JV, g->JA, g->IA, nullptr, p->s0, vb, gt_vidmap_of(g), p->prm.kind, (uint32_t *)p->y, p->fl_rows, p->d_fl + 2);
        GT_HIP(hipGetLastError());
        p->spmspv_iters++; p->fl_rows_valid = true; *done = true;
        return GT_OK;
    }
    if (nact == 0) { *done = true; p->fl_rows_valid = true; GT_HIP(hipMemsetAsync(p->d_fl + 2, 0, sizeof(unsigned int), s)); return GT_OK; }   // nothing is active: y keeps its minima
    if (!force) {   // counting costs a pass over the list and a round trip (0.15 ms at 8 M vertices): not for a frontier that cannot
                    // qualify -- the mid-run frontiers hold 8+ entries per vertex, only the tail ones fewer (1.3)
        const uint64_t frac0 = gt_cfg(p, "GRAPHTAP_SPMSPV_FRACTION") ? (uint64_t)atoll(gt_cfg(p, "GRAPHTAP_SPMSPV_FRACTION")) : 32;
        if ((uint64_t)nact * 8 > g->info.nnz_local / frac0) return GT_OK;
    }
    if (p->fr_cap < nact + 1) { int st = gt_spmspv_reserve(p, nact); if (st != GT_OK) return st; }
    GT_HIP(hipMemsetAsync(p->d_frontier, 0, 4 * sizeof(unsigned long long), s));
    const unsigned grid = (unsigned)std::min<uint64_t>((nact + TPB - 1) / TPB, 4096);
    k_list_frontier<<<grid, TPB, 0, s>>>(p->fl_v[p->fl_cur], nact, g->IJ, g->JV, g->JA, p->s0, g->info.rank * g->info.tile_height, gt_vidmap_of(g),
                                         p->prm.kind, p->fr_col, p->fr_val, p->fr_off, p->d_frontier);
    GT_HIP(hipMemsetAsync(p->d_fl + 2, 0, sizeof(unsigned int), s));
    // a few thousand columns: no need to know their entries first (even if they were all hubs the entry-parallel kernel takes
    // them in a fraction of a streaming pass) -- the count costs a host round trip, a quarter of such an iteration
    static const uint32_t count_from = getenv("GRAPHTAP_SPMSPV_COUNT_FROM") ? (uint32_t)atoi(getenv("GRAPHTAP_SPMSPV_COUNT_FROM")) : 4096;
    if (nact <= count_from) {
        int st = gt_spmspv_run_frontier(p, nact, s);
        if (st != GT_OK) return st;
        *done = true;
        return GT_OK;
    }
    unsigned long long h[2] = {0, 0};
    { int st = gt_read_back(p, h, p->d_frontier, sizeof(h), s); if (st != GT_OK) return st; }
    if (h[1] == 0) { *done = true; p->fl_rows_valid = true; return GT_OK; }
    // From the list, with eight lanes per column, the sparse pass wins up to ~nnz/32 entries (tools/spmspv_sweep.sh on R-MAT-26:
    // 10 M entries of 1.27 M columns 0.41 against 1.49 ms, 4.7 M of 3.7 M columns 0.41 against 0.95 ms, but 102 M entries of
    // 8.1 M columns 2.1 against 1.8 ms)
    const uint64_t frac = gt_cfg(p, "GRAPHTAP_SPMSPV_FRACTION") ? (uint64_t)atoll(gt_cfg(p, "GRAPHTAP_SPMSPV_FRACTION")) : 32;
    if (!force && h[1] > g->info.nnz_local / frac) return GT_OK;   // the streaming pass does it (x is complete either way)
    GT_REQUIRE(h[1] < 0xFFFFFFFFull, GT_ERR_UNSUPPORTED, "frontier entry offsets exceed 32 bits");
    int st = gt_spmspv_run_frontier(p, nact, s);
    if (st != GT_OK) return st;
    *done = true;
    return GT_OK;
}

// fr_col / fr_val hold the frontier's columns and messages, fr_off their entry counts (nact of them); d_fl[2] (the length of
// fl_rows) must be zero. The total entry count stays on the device.
int gt_spmspv_run_frontier(gt_program *p, uint32_t nact, hipStream_t s) {
    const gt_graph *g = p->g;
    const unsigned grid = (unsigned)std::max<uint64_t>(std::min<uint64_t>(((uint64_t)nact + TPB - 1) / TPB, 4096), 1);
    const bool weighted = p->semiring == GT_MINPLUS_U32;
    uint32_t *marks = (uint32_t *)p->row_mark;
    // the short columns: eight lanes each
    constexpr uint32_t BIG = 2048;
    const unsigned gc = (unsigned)std::max<uint64_t>(std::min<uint64_t>(((uint64_t)nact + TPB / 8 - 1) / (TPB / 8), 256u * 64u), 1);
    if (weighted) k_spmspv_cols<true><<<gc, TPB, 0, s>>>(p->fr_col, p->fr_val, p->fr_off, nact, BIG, g->JA, g->IA, g->A, (uint32_t *)p->y, marks);
    else k_spmspv_cols<false><<<gc, TPB, 0, s>>>(p->fr_col, p->fr_val, p->fr_off, nact, BIG, g->JA, g->IA, nullptr, (uint32_t *)p->y, marks);
    // the long ones (hubs): one thread per entry
    GT_HIP(hipMemsetAsync(p->d_frontier + 2, 0, sizeof(unsigned long long), s));
    k_keep_big<<<grid, TPB, 0, s>>>(p->fr_off, nact, BIG, p->d_frontier + 2);
    {
        size_t tb = 0;
        GT_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, p->fr_off, p->fr_off, nact, s));
        if (tb > p->fr_tmp_bytes) { if (p->fr_tmp) GT_HIP(hipFree(p->fr_tmp)); p->fr_tmp = nullptr; GT_HIP(hipMalloc(&p->fr_tmp, tb)); p->fr_tmp_bytes = tb; p->spmspv_allocs++; }
        GT_HIP(hipcub::DeviceScan::ExclusiveSum(p->fr_tmp, tb, p->fr_off, p->fr_off, nact, s));
    }
    const unsigned g2 = 256u * 16u;   // the kernel reads the number of long-column entries from the device
    if (weighted) k_spmspv_min<true, true><<<g2, TPB, 0, s>>>(p->fr_col, p->fr_val, p->fr_off, nact, 0, p->d_frontier + 2, g->JA, g->IA, g->A, (uint32_t *)p->y, marks);
    else k_spmspv_min<false, true><<<g2, TPB, 0, s>>>(p->fr_col, p->fr_val, p->fr_off, nact, 0, p->d_frontier + 2, g->JA, g->IA, nullptr, (uint32_t *)p->y, marks);
    // the rows either kernel lowered, in ascending order
    const uint32_t nwords = g->info.nnzrows / 32 + 1;
    k_rows_from_marks<<<(unsigned)std::min<uint64_t>(((uint64_t)nwords + 4095) / 4096, 4096), TPB, 0, s>>>(marks, nwords, p->fl_rows, p->d_fl + 2);
    GT_HIP(hipGetLastError());
    p->spmspv_iters++;
    p->fl_rows_valid = true;
    return GT_OK;
}

// HIP loads the code object of a translation unit at the first launch of one of its kernels (2.4 ms for this one): initialize()
// calls this so that the load does not fall into the first iteration of execute()
int gt_kernels_preload(hipStream_t s) {
    k_preload<<<1, 64, 0, s>>>();
    GT_HIP(hipGetLastError());
    return GT_OK;
}

int gt_frontier_messages(gt_program *p, hipStream_t s) {
    const gt_graph *g = p->g;
    const unsigned int *n_prev = p->d_fl + (p->fl_cur ^ 1), *n_cur = p->d_fl + p->fl_cur;
    auto grid = [](uint32_t n) { return (unsigned)std::max<uint64_t>(std::min<uint64_t>(((uint64_t)n + TPB - 1) / TPB, 4096), 1); };
    uint32_t *xm = (uint32_t *)(p->xseg ? p->xseg : p->x);   // several ranks: the owned columns' messages (packed per destination afterwards)
    if (p->fl_prev_n) k_list_reset_x<<<grid(p->fl_prev_n), TPB, 0, s>>>(xm, p->fl_v[p->fl_cur ^ 1], n_prev, g->IJ, g->JV, g->xslot);
    if (p->fl_cur_n) k_list_msg<<<grid(p->fl_cur_n), TPB, 0, s>>>(xm, p->fl_v[p->fl_cur], n_cur, g->IJ, g->JV, g->xslot, p->s0,
                                                                   g->info.rank * g->info.tile_height, gt_vidmap_of(g), p->prm.kind);
    GT_HIP(hipGetLastError());
    return GT_OK;
}
