// engine.hip -- vertex-program engine + the extern "C" ABI of include/graphtap_amd.h.
//
// Replaces Vertex_Program<> (src/vp/vertex_program.hpp): state vectors V / C live in HBM as
// struct-of-arrays over the owned vertex segment; messenger (K7/K8), applicator (K10/K11) and
// the convergence count (K12) are kernels; the five apps' hooks (src/apps/*.h) are op-codes.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gt_internal.h"

// ------------------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";
void gt_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
constexpr int TPB = 256;
inline unsigned grid_for(uint64_t n) {
    uint64_t b = (n + TPB - 1) / TPB;
    if (b < 1) b = 1;
    if (b > 256u * 16u) b = 256u * 16u;
    return (unsigned)b;
}
}  // namespace

// ------------------------------------------------------------------ messenger kernels (K7/K8)
// scatter_gather_stationary vp:688-708 / _nonstationary vp:711-758 over the owned segment's
// non-empty columns: x[j] = messenger(V[JC[j]]), C-gated to INF for the min programs. `JC` is the slot -> vertex map
// of the message vector's layout (gt_x_vertex: JC itself, or XV under the hubs-first layout of pb.hip).
__global__ void k_msg_deg(uint32_t *__restrict__ x, uint32_t nc) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) x[j] = 1u;  // deg.h:35-37
}
template <class TX>
__global__ void k_msg_pr(TX *__restrict__ x, const uint32_t *__restrict__ JC, uint32_t nc,
                         const uint32_t *__restrict__ deg, const double *__restrict__ rank) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        uint32_t v = JC[j];
        if (v == 0xFFFFFFFFu) continue;              // unused slot of the hubs-first layout
        uint32_t d = deg[v];
        x[j] = (TX)(d ? rank[v] / (double)d : 0.0);  // pr.h:31-33
    }
}
__global__ void k_msg_min(uint32_t *__restrict__ x, const uint32_t *__restrict__ JC, uint32_t nc,
                          const uint8_t *__restrict__ C, const uint32_t *__restrict__ s0, uint32_t vid_base, gt_vidmap vm, int kind) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        uint32_t v = JC[j];
        if (v == 0xFFFFFFFFu) continue;              // unused slot of the hubs-first layout: stays infinity()
        // bfs.h:52-54 (vid), sssp.h:44-46 (distance), cc.h:38-40 (label); inactive -> infinity() vp:749-750
        x[j] = C[v] ? (kind == GT_BFS ? gt_vid_of(vm, (uint64_t)vid_base + v) : s0[v]) : GT_INF;
    }
}

// Deg in _COL_ order (apps/pr.cpp:40-42; vp:1174-1184 with x == 1): y[c] = entries in column c. Several ranks: the
// tile-row's local columns scatter into the global [segment][seg_stride] space (zero-filled by the caller).
__global__ void k_col_counts(const uint32_t *__restrict__ JA, uint32_t ncols, const uint32_t *__restrict__ loc2glob, uint32_t *__restrict__ y) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x) {
        const uint32_t o = loc2glob ? loc2glob[c] : c;
        if (o != 0xFFFFFFFFu) y[o] = JA[c + 1] - JA[c];
    }
}
// compressed-column order -> slot order of the message vector (gt_spmv under the hubs-first layout)
template <class TX>
__global__ void k_to_slots(const TX *__restrict__ x, const uint32_t *__restrict__ xslot, uint32_t nc, TX *__restrict__ out) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) out[xslot[c]] = x[c];
}
// messages of the owned columns -> send buffer, block (slice k, destination d) after block (ingest.hip: k_send_list)
template <class TX>
__global__ void k_pack_send(const TX *__restrict__ xseg, const uint32_t *__restrict__ send_idx, uint64_t n, TX *__restrict__ send) {
    // (bound by the line traffic of the gathers, not by their latency: xseg of a tile-row of 8 of R-MAT-26 is 13.5 MB, every XCD's L2 sees
    // all of it, and 12.6 M gathers pull ~1.6 GB of lines out of the Infinity Cache in 73-87 us; four independent chains per thread
    // were no faster -- rocprofv3, profiles/r03/tilerow3_of_8_kernel_stats.csv; what helped is fewer sweeps: the blocks' tail in column order, ingest.hip)
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) send[i] = xseg[send_idx[i]];
}

// ------------------------------------------------------------------ applicator kernels (K10/K11)
// Local term of has_converged (K12): one atomic per BLOCK (same-address atomics serialise: one per wave cost
// 0.19 ms on a 3.4 M-row tile-row), and none at all when the caller does not ask for the count (d_active == null).
__device__ __forceinline__ void count_active(unsigned act, unsigned long long *d_active) {
    if (d_active == nullptr) return;
    __shared__ unsigned wave_sums[TPB / 64];
    for (int o = 32; o > 0; o >>= 1) act += __shfl_down(act, o);
    if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = act;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < TPB / 64; w++) t += wave_sums[w];
        if (t) atomicAdd(d_active, (unsigned long long)t);
    }
}

// rows without an accumulator slot: C[i] = applicator(state) = false (vp:1666-1667, 1735-1736)
__global__ void k_clear_empty_rows(uint8_t *__restrict__ C, const uint8_t *__restrict__ IJ, uint32_t H) {
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < H; v += gridDim.x * blockDim.x)
        if (!(IJ[v] & 1)) C[v] = 0;
}
// BFS / SSSP: the root is the only vertex initialize() left active (bfs.h:37-50, sssp.h:33-42), so it is the only one the pass above
// could change (85 us for the 67 M vertices of R-MAT-26, in iteration 0 of every run)
__global__ void k_clear_if_empty_row(uint8_t *__restrict__ C, const uint8_t *__restrict__ IJ, uint32_t v) { if (!(IJ[v] & 1)) C[v] = 0; }

__global__ void k_apply_deg_row(const uint32_t *__restrict__ y, const uint32_t *__restrict__ IR, uint32_t nr,
                                uint32_t *__restrict__ deg, uint8_t *__restrict__ C) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += gridDim.x * blockDim.x) {
        uint32_t v = IR[r];
        deg[v] = y[r]; C[v] = 0;  // deg.h:47-50
    }
}
__global__ void k_apply_deg_col(const uint32_t *__restrict__ y, const uint32_t *__restrict__ JC, uint32_t nc,
                                uint32_t *__restrict__ deg, uint8_t *__restrict__ C) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        uint32_t v = JC[j];
        deg[v] = y[j]; C[v] = 0;
    }
}
// PageRank in compressed-row space: applicator (pr.h:43-47) of iteration t fused with the messenger
// (pr.h:31-33) of iteration t+1 and with the zero-fill of y (K13): one sequential pass over the rows.
// A vertex without in-edges has no row: its degree is 0 after initialize(other) (vp:476-483), so its
// message is the constant 0 written once at initialize; only rows ever produce non-zero messages.
template <class TX>
__global__ void k_pr_apply_msg(double *__restrict__ y, const uint32_t *__restrict__ R2C, uint32_t nr,
                               double *__restrict__ rank_c, const uint32_t *__restrict__ deg_c, uint8_t *__restrict__ C_c,
                               TX *__restrict__ x, double alpha, double tol, int cf, int last,
                               unsigned long long *d_active, const uint32_t *__restrict__ bins, uint32_t nbins_listed, int state) {
    // bins == null: every row. Otherwise only the rows of the listed row bins (the ones phase 2 did not apply itself, pb.hip).
    constexpr uint32_t RB = GT_PB_ROW_BIN_BITS, RR = 1u << RB;
    const uint64_t n = bins ? (uint64_t)nbins_listed << RB : nr;
    unsigned act = 0;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = bins ? bins[t >> RB] * RR + (uint32_t)(t & (RR - 1)) : (uint32_t)t;
        if (r >= nr) continue;
        const uint32_t c = R2C[r];
        const bool source = (c == 0xFFFFFFFFu);
        const double yr = y[r];
        y[r] = 0.0;
        if (cf && source && !last) continue;   // vp:1671-1691
        const double nv = alpha + (1.0 - alpha) * yr;
        const double tmp = state == 0 ? rank_c[r] : 0.0;   // state: gt_program::pr_state (1, 2: nobody can see this iteration's rank / changed flag)
        if (state != 2) rank_c[r] = nv;
        if (state == 0) {
            const uint8_t ch = fabs(nv - tmp) > tol;
            C_c[r] = ch;
            act += (ch && !(cf && source));
        }
        if (!source) { const uint32_t d = deg_c[r]; x[c] = (TX)(d ? nv / (double)d : 0.0); }
    }
    count_active(act, d_active);
}
__global__ void k_pr_pack(const uint32_t *__restrict__ IR, uint32_t nr, const double *__restrict__ rank, const uint32_t *__restrict__ deg,
                          const uint8_t *__restrict__ C, double *__restrict__ rank_c, uint32_t *__restrict__ deg_c, uint8_t *__restrict__ C_c) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += gridDim.x * blockDim.x) {
        const uint32_t v = IR[r];
        rank_c[r] = rank[v]; deg_c[r] = deg[v]; C_c[r] = C[v];
    }
}
__global__ void k_pr_unpack(const uint32_t *__restrict__ IR, uint32_t nr, const double *__restrict__ rank_c, const uint8_t *__restrict__ C_c,
                            double *__restrict__ rank, uint8_t *__restrict__ C) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += gridDim.x * blockDim.x) {
        const uint32_t v = IR[r];
        rank[v] = rank_c[r]; C[v] = C_c[r];
    }
}
__global__ void k_pr_cf_tail_c(const uint32_t *__restrict__ R2C, uint32_t nr, double *__restrict__ rank_c, double alpha) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += gridDim.x * blockDim.x)
        if (R2C[r] == 0xFFFFFFFFu) rank_c[r] = alpha + (1.0 - alpha) * 0.0;
}
// All threads of a workgroup call (kernels.hip has the same helper for its lists): one reservation per workgroup.
__device__ __forceinline__ void block_append(bool want, uint32_t value, uint32_t *__restrict__ list, unsigned int *__restrict__ cursor, uint32_t cap) {
    __shared__ unsigned wave_n[TPB / 64];
    __shared__ unsigned round_base;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t b = __ballot(want);
    if (lane == 0) wave_n[wave] = (unsigned)__popcll((unsigned long long)b);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
        for (int w = 0; w < TPB / 64; w++) { const unsigned n = wave_n[w]; wave_n[w] = total; total += n; }
        round_base = total ? atomicAdd(cursor, total) : 0u;
    }
    __syncthreads();
    if (want) {
        const uint32_t o = round_base + wave_n[wave] + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
        if (o < cap) list[o] = value;
    }
    __syncthreads();
}
// applicator of one row: bfs.h:65-77 / sssp.h:57-65 (HAS_WEIGHT), cc.h:51-55; returns "changed"
template <bool BFS>
__device__ __forceinline__ bool apply_row(uint32_t v, uint32_t yv, uint32_t *__restrict__ s0, uint32_t *__restrict__ s1, uint32_t iteration) {
    if constexpr (BFS) {
        if (s1[v] == GT_INF && yv != GT_INF) { s1[v] = iteration + 1; s0[v] = yv; return true; }   // s1 = hops, s0 = parent
        return false;
    } else {
        const uint32_t tmp = s0[v], nv = yv < tmp ? yv : tmp;
        s0[v] = nv;
        return nv != tmp;
    }
}
// every row. MSG: the applicator also writes the NEXT iteration's message of the row's vertex -- its new value if it changed,
// infinity() otherwise (vp:749-750) -- through the row -> slot map, as PageRank's fused applicator does: the full messenger pass
// of the next scatter_gather (a gather of C and the state through slot -> vertex: 0.14-0.4 ms in the iterations that run a full
// apply) disappears.
template <bool BFS, bool MSG>
__global__ void __launch_bounds__(TPB) k_apply_rows(const uint32_t *__restrict__ y, const uint32_t *__restrict__ IR, uint32_t nr,
                                                    uint32_t *__restrict__ s0, uint32_t *__restrict__ s1, uint8_t *__restrict__ C, uint32_t iteration,
                                                    unsigned long long *d_active, uint32_t *__restrict__ x, const uint32_t *__restrict__ R2X,
                                                    uint32_t vid_base, gt_vidmap vm, uint32_t *__restrict__ level_bits = nullptr,
                                                    uint32_t *__restrict__ reached_bits = nullptr) {
    unsigned act = 0;
    // (whole waves: with the row bitmaps of BFS's bottom-up steps a wave's 64 rows are two words of each -- gt_internal.h, bu_reached)
    const uint32_t n64 = (nr + 63u) & ~63u;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n64; r += gridDim.x * blockDim.x) {
        bool c = false;
        if (r < nr) {
            const uint32_t v = IR[r];
            c = apply_row<BFS>(v, y[r], s0, s1, iteration);
            C[v] = c; act += c;
            if constexpr (MSG) {
                const uint32_t sl = R2X[r];
                if (sl != 0xFFFFFFFFu) x[sl] = c ? (BFS ? gt_vid_of(vm, (uint64_t)vid_base + v) : s0[v]) : GT_INF;   // bfs.h:52-54, sssp.h:44-46, cc.h:38-40
            }
        }
        if constexpr (BFS) {
            if (level_bits) {   // wave-uniform
                const uint64_t b = __ballot(c);
                if ((threadIdx.x & 63u) == 0) {
                    const uint32_t w = r >> 5;
                    level_bits[w] = (uint32_t)b; level_bits[w + 1] = (uint32_t)(b >> 32);
                    if (b) { reached_bits[w] |= (uint32_t)b; reached_bits[w + 1] |= (uint32_t)(b >> 32); }
                }
            }
        }
    }
    count_active(act, d_active);
}
// once per initialize(), with the first fused apply: the columns of vertices WITHOUT a row (no in-edges: no apply ever visits
// them) sent what they had to send in iteration 0 (the root; every vertex for CC) and are inactive from then on
__global__ void k_reset_rowless_messages(uint32_t *__restrict__ x, const uint32_t *__restrict__ XV, uint32_t nslots, const uint8_t *__restrict__ IJ) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nslots; j += gridDim.x * blockDim.x) {
        const uint32_t v = XV[j];
        if (v != 0xFFFFFFFFu && !(IJ[v] & 1u)) x[j] = GT_INF;
    }
}
// the list of the vertices a full apply changed, from their flags -- run only when the count it returned fits a list (appending
// inside the full apply costs two barriers and a reservation per 256 rows: 1.5 instead of 0.3 ms when millions change). A
// workgroup reserves once per 4096 rows: reservations all hit one word, and 10^5 of them are a millisecond.
__global__ void __launch_bounds__(TPB) k_list_from_flags(const uint32_t *__restrict__ IR, uint32_t nr, const uint8_t *__restrict__ C,
                                                         uint32_t *__restrict__ next, unsigned int *__restrict__ next_n, uint32_t cap) {
    constexpr uint32_t PER = 16, SPAN = PER * TPB;
    __shared__ unsigned wave_n[TPB / 64];
    __shared__ unsigned span_base;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nspan = (nr + SPAN - 1) / SPAN;
    for (uint32_t sp = blockIdx.x; sp < nspan; sp += gridDim.x) {
        uint32_t mask = 0, cnt = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t r = sp * SPAN + k * TPB + threadIdx.x;
            if (r < nr && C[IR[r]]) { mask |= 1u << k; cnt++; }
        }
        uint32_t inc = cnt;   // inclusive prefix over the wave
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += t; }
        if (lane == 63) wave_n[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned total = 0;
            for (int w = 0; w < TPB / 64; w++) { const unsigned n = wave_n[w]; wave_n[w] = total; total += n; }
            span_base = total ? atomicAdd(next_n, total) : 0u;
        }
        __syncthreads();
        uint32_t o = span_base + wave_n[wave] + inc - cnt;
        for (uint32_t k = 0; mask; k++, mask >>= 1)
            if (mask & 1u) { if (o < cap) next[o] = IR[sp * SPAN + k * TPB + threadIdx.x]; o++; }
        __syncthreads();
    }
}
// only the rows the SpMSpV lowered (fl_rows); C of the vertices that were active is cleared by k_list_clear_flags first
template <bool BFS>
__global__ void __launch_bounds__(TPB) k_apply_list(const uint32_t *__restrict__ rows, const unsigned int *__restrict__ n_dev, const uint32_t *__restrict__ y,
                                                    const uint32_t *__restrict__ IR, uint32_t *__restrict__ s0, uint32_t *__restrict__ s1,
                                                    uint8_t *__restrict__ C, uint32_t iteration, unsigned long long *d_active,
                                                    uint32_t *__restrict__ next, unsigned int *__restrict__ next_n, uint32_t cap,
                                                    uint32_t *__restrict__ level_bits = nullptr, uint32_t *__restrict__ reached_bits = nullptr) {
    unsigned act = 0;
    const uint32_t n = *n_dev, n_round = (n + TPB - 1) / TPB * TPB;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_round; t += gridDim.x * blockDim.x) {
        bool c = false; uint32_t v = 0;
        if (t < n) {
            const uint32_t r = rows[t];
            if (r != 0xFFFFFFFFu) {   // ~0u: a bottom-up row that found nobody
                v = IR[r]; c = apply_row<BFS>(v, y[r], s0, s1, iteration); C[v] = c; act += c;
                if constexpr (BFS) if (c && level_bits) { atomicOr(&level_bits[r >> 5], 1u << (r & 31u)); atomicOr(&reached_bits[r >> 5], 1u << (r & 31u)); }   // (level_bits zeroed by the caller)
            }
        }
        if (__syncthreads_or(c)) block_append(c, v, next, next_n, cap);
    }
    count_active(act, d_active);
}
__global__ void k_list_clear_flags(uint8_t *__restrict__ C, const uint32_t *__restrict__ list, const unsigned int *__restrict__ n_dev) {
    const uint32_t n = *n_dev;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) C[list[i]] = 0;
}
// the TCSC_CF converged tail (vp:425-428 -> 1683-1691): source rows get alpha + (1-alpha) * 0 (k_pr_cf_tail_c above)

// ------------------------------------------------------------------ init kernels
template <class T> __global__ void k_fill(T *__restrict__ p, uint64_t n, T v) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}
// initializer(vid, state): bfs.h:37-50, sssp.h:33-42, cc.h:33-36
__global__ void k_init_min(int kind, uint32_t H, uint32_t vid_base, gt_vidmap vm, uint32_t root, uint32_t *__restrict__ s0,
                           uint32_t *__restrict__ s1, uint8_t *__restrict__ C) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < H; i += gridDim.x * blockDim.x) {
        const uint32_t vid = gt_vid_of(vm, (uint64_t)vid_base + i);   // ~0u for padding slots: never equals a root, never a label that wins
        if (kind == GT_BFS) { s0[i] = (vid == root) ? vid : 0; s1[i] = (vid == root) ? 0 : GT_INF; C[i] = (vid == root); }
        else if (kind == GT_SSSP) { s0[i] = (vid == root) ? 0 : GT_INF; C[i] = (vid == root); }
        else { s0[i] = vid; C[i] = 1; }
    }
}
// initialize(other), vp:476-483 with pr.h:24-28: degree copied only where the row is non-empty
__global__ void k_pr_take_degree(uint32_t H, const uint8_t *__restrict__ IJ, const uint32_t *__restrict__ other_deg,
                                 uint32_t *__restrict__ deg) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < H; i += gridDim.x * blockDim.x)
        deg[i] = (IJ[i] & 1) ? other_deg[i] : 0;
}

// ------------------------------------------------------------------ R-MAT generator
// counter based; bit-identical to graphtap_amd/rmat.py (SURVEY 8d parameters)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_rmat(uint32_t *__restrict__ out, int scale, uint64_t seed, int weighted, uint64_t first, uint64_t count) {
    const uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;
    const int stride = weighted ? 3 : 2;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t base = mix64(seed * GOLDEN + (first + i));
        uint32_t src = 0, dst = 0;
        for (int l = 0; l < scale; l++) {
            uint32_t u = (uint32_t)(mix64(base + (uint64_t)(l + 1) * GOLDEN) >> 32);
            uint32_t rbit = (u >= 3264175144u);
            uint32_t cbit = ((u >= 2448131358u) & (u < 3264175144u)) | (u >= 4080218931u);
            src = (src << 1) | rbit; dst = (dst << 1) | cbit;
        }
        out[i * stride] = src; out[i * stride + 1] = dst;
        if (weighted) out[i * stride + 2] = (uint32_t)(mix64(base ^ 0xD1B54A32D192ED03ull) % 128ull) + 1u;
    }
}

// =============================================================================== C ABI
extern "C" {

int gt_abi_version(void) { return GT_ABI_VERSION; }
const char *gt_last_error(void) { return g_err; }

int gt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int gt_set_device(int device) {
    GT_REQUIRE(gt_device_count() > 0, GT_ERR_NO_DEVICE, "no HIP device visible: graphtap_amd has no CPU fallback");
    GT_HIP(hipSetDevice(device));
    return GT_OK;
}

int gt_malloc(void **p, uint64_t bytes) { GT_HIP(hipMalloc(p, bytes ? bytes : 1)); return GT_OK; }
int gt_free(void *p) { GT_HIP(hipFree(p)); return GT_OK; }
int gt_memcpy_h2d(void *d, const void *h, uint64_t bytes) { GT_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return GT_OK; }
int gt_memcpy_d2h(void *h, const void *d, uint64_t bytes) { GT_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); return GT_OK; }
int gt_memset(void *d, int v, uint64_t bytes) { GT_HIP(hipMemset(d, v, bytes)); return GT_OK; }
int gt_device_synchronize(void) { GT_HIP(hipDeviceSynchronize()); return GT_OK; }

int gt_rmat_generate(void *dev_out, int scale, uint64_t seed, int weighted, uint64_t first, uint64_t count, void *hip_stream) {
    GT_REQUIRE(dev_out && scale > 0 && scale <= 31, GT_ERR_INVALID, "gt_rmat_generate: bad arguments");
    if (count) k_rmat<<<grid_for(count), TPB, 0, (hipStream_t)hip_stream>>>((uint32_t *)dev_out, scale, seed, weighted, first, count);
    GT_HIP(hipGetLastError());
    return GT_OK;
}

// ---- graph
int gt_graph_free(gt_graph *g) {
    if (!g) return GT_OK;
    void *ptrs[] = {g->JA, g->IA, g->A, g->JI, g->JC, g->IR, g->IJ, g->IV, g->JV, g->R2C, g->loc2glob, g->send_idx, g->xslot, g->xcol, g->XV, g->R2X, g->x_scratch};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    gt_pb_free(g->pb);
    gt_pb_free(g->pb_wide);
    gt_tcsc_cf_free(g->cf);
    delete g;
    return GT_OK;
}

static int graph_build_impl(gt_graph **out, const void *edges, uint64_t m, int edges_on_device, int weighted,
                            uint32_t num_vertices, const gt_graph_flags *flags, int rank, int nranks, gt_dist *dist, const gt_graph_options *opt);
int gt_graph_build(gt_graph **out, const void *edges, uint64_t m, int edges_on_device, int weighted,
                   uint32_t num_vertices, const gt_graph_flags *flags, int rank, int nranks) {
    return graph_build_impl(out, edges, m, edges_on_device, weighted, num_vertices, flags, rank, nranks, nullptr, nullptr);
}
// Matrix::distribute (mat/matrix.hpp:693-810): every rank brings a SHARE of the records (any split of the list); the build moves
// each record to the owner of its row(s) and gets the global pieces from collectives over `dist`. Collective: every rank of the
// communicator calls it. The result is the graph gt_graph_build gives for the same rank from the full list.
int gt_graph_build_distributed(gt_graph **out, gt_dist *dist, const void *edges_share, uint64_t m_share, int edges_on_device, int weighted,
                               uint32_t num_vertices, const gt_graph_flags *flags) {
    GT_REQUIRE(dist, GT_ERR_INVALID, "gt_graph_build_distributed: null communicator");
    return graph_build_impl(out, edges_share, m_share, edges_on_device, weighted, num_vertices, flags, gt_dist_rank(dist), gt_dist_nranks(dist), dist, nullptr);
}
// ---- handle-level configuration (include/graphtap_amd.h): a field that is set becomes an override of the environment variable it
// names, stored in the handle (gt_overrides); every site that reads such a knob asks gt_cfg(handle, name)
void gt_graph_options_init(gt_graph_options *o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->size = (uint32_t)sizeof(*o); o->spmv_variant = -1; o->force_exchange = -1; o->hubs_first = -1; o->wide_windows = -1;
}
void gt_program_options_init(gt_program_options *o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->size = (uint32_t)sizeof(*o);
    o->frontier_lists = o->spmspv = o->tail_kernel = o->bfs_bottom_up = o->cc_first = o->fuse_apply = o->lean_state = o->hybrid = -1;
}
int gt_graph_build_opt(gt_graph **out, gt_dist *dist, const void *edges, uint64_t m, int edges_on_device, int weighted,
                       uint32_t num_vertices, const gt_graph_flags *flags, int rank, int nranks, const gt_graph_options *opt) {
    GT_REQUIRE(!opt || opt->size >= 8, GT_ERR_INVALID, "gt_graph_options: size is not set (gt_graph_options_init)");
    if (dist) { rank = gt_dist_rank(dist); nranks = gt_dist_nranks(dist); }
    return graph_build_impl(out, edges, m, edges_on_device, weighted, num_vertices, flags, rank, nranks, dist, opt);
}
static int graph_build_impl(gt_graph **out, const void *edges, uint64_t m, int edges_on_device, int weighted,
                            uint32_t num_vertices, const gt_graph_flags *flags, int rank, int nranks, gt_dist *dist, const gt_graph_options *opt_in) {
    GT_REQUIRE(out && flags && (edges || m == 0), GT_ERR_INVALID, "gt_graph_build: null argument");
    GT_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, GT_ERR_INVALID, "gt_graph_build: rank %d of %d", rank, nranks);
    GT_REQUIRE(num_vertices < 0xFFFFFFF0u - (uint32_t)nranks, GT_ERR_INVALID, "num_vertices too large for 32-bit vertex ids");
    GT_REQUIRE(gt_device_count() > 0, GT_ERR_NO_DEVICE, "no HIP device visible: graphtap_amd has no CPU fallback");
    *out = nullptr;
    gt_graph *g = new gt_graph();
    { static std::atomic<uint64_t> serials{0}; g->serial = ++serials; }
    if (opt_in) {   // only the fields the caller's struct has (its `size`), only the ones that are set
        gt_graph_options o; gt_graph_options_init(&o);
        memcpy(&o, opt_in, std::min<size_t>(opt_in->size, sizeof(o)));
        if (o.spmv_variant >= 0) {
            if (o.spmv_variant > GT_SPMV_PB_F32MSG) { delete g; gt_set_error("gt_graph_options: unknown SpMV variant %d", o.spmv_variant); return GT_ERR_INVALID; }
            g->cfg.set("GRAPHTAP_SPMV", o.spmv_variant == GT_SPMV_EDGE ? "edge" : o.spmv_variant == GT_SPMV_PB_F32MSG ? "pb_f32msg" : "pb");
        }
        if (o.force_exchange >= 0) g->cfg.set("GRAPHTAP_FORCE_EXCHANGE", o.force_exchange ? "1" : "0");
        if (o.x_slices) g->cfg.set("GRAPHTAP_X_SLICES", std::to_string(o.x_slices));
        if (o.hubs_first >= 0) g->cfg.set("GRAPHTAP_PB_HUBS", o.hubs_first ? "1" : "0");
        if (o.hub_min_degree) g->cfg.set("GRAPHTAP_PB_HUB_DEG", std::to_string(o.hub_min_degree));
        if (o.exchange_hub_min) g->cfg.set("GRAPHTAP_EXCHANGE_HUB_MIN", std::to_string(o.exchange_hub_min));
        if (o.chunk_log2) g->cfg.set("GRAPHTAP_PB_CH", std::to_string(o.chunk_log2));
        if (o.wide_windows >= 0) g->cfg.set("GRAPHTAP_PB_WIDE", o.wide_windows ? "1" : "0");
    }
    g->flags = *flags;
    g->info.num_vertices = num_vertices;
    g->info.nrows = num_vertices + 1;                          // mat/graph.hpp:89-90
    g->info.nranks = (uint32_t)nranks; g->info.rank = (uint32_t)rank;
    // One rank: the reference's id space as is. Several ranks: contiguous id ranges of an R-MAT (or any
    // degree-skewed) graph are badly unbalanced -- tile-row 0 of 8 holds 44 % of R-MAT-26 -- so ranks own
    // contiguous ranges of a multiplicatively hashed INTERNAL id space (a bijection on the next power of
    // two; results are reported in original ids, see gt_graph_vertex_ids).
    g->nint = g->info.nrows;
    { const char *fe = gt_cfg(g, "GRAPHTAP_FORCE_EXCHANGE"); g->force_exchange = fe != nullptr && atoi(fe) != 0; }
    if (nranks > 1) {
        uint32_t M = 1; while (M < g->info.nrows && M < 0x80000000u) M <<= 1;
        GT_REQUIRE(M >= g->info.nrows, GT_ERR_UNSUPPORTED, "more than 2^31 vertices on several ranks");
        g->perm_a = 0x9E3779B1u; g->perm_mask = M - 1; g->nint = M;
        uint32_t inv = g->perm_a;   // Newton: inverse of an odd number modulo 2^32
        for (int it = 0; it < 5; it++) inv *= 2u - g->perm_a * inv;
        g->perm_ainv = inv;
    }
    g->info.tile_height = g->nint / (uint32_t)nranks + 1;  // mat/matrix.hpp:193, over the (internal) id space
    g->info.weighted = weighted ? 1 : 0;
    const void *dev_edges = edges;
    void *staged = nullptr;
    const uint64_t bytes = m * (weighted ? 12ull : 8ull);
    if (!edges_on_device && m) {
        if (hipMalloc(&staged, bytes) != hipSuccess) { delete g; gt_set_error("out of device memory staging %llu edge bytes", (unsigned long long)bytes); return GT_ERR_HIP; }
        if (hipMemcpy(staged, edges, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(staged); delete g; gt_set_error("H2D copy of the edge list failed"); return GT_ERR_HIP; }
        dev_edges = staged;
    }
    const bool verbose = getenv("GRAPHTAP_PB_STATS") != nullptr;
    auto tick = [&](const char *what, std::chrono::steady_clock::time_point &t) {
        if (!verbose) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        gt_alloc_clock &c = gt_alloc_clock_ref();
        fprintf(stderr, "[build] %s: %.1f ms (scratch: %llu hipMalloc %.1f ms, %.2f GB; %llu hipFree %.1f ms)\n", what,
                std::chrono::duration<double, std::milli>(now - t).count(), (unsigned long long)c.mallocs, c.malloc_ms, c.bytes / 1e9,
                (unsigned long long)c.frees, c.free_ms);
        c = gt_alloc_clock{};
        t = now;
    };
    auto tb = std::chrono::steady_clock::now();
    if (dist && nranks == 1 && !g->force_exchange) dist = nullptr;   // one rank without the exchange layout: nothing to distribute
    int st = gt_ingest(g, dev_edges, m, weighted, dist);
    tick("ingest (TCSC tile-row)", tb);
    if (staged) (void)hipFree(staged);
    if (st != GT_OK) { gt_scratch_release(); gt_graph_free(g); return st; }
    const char *env = gt_cfg(g, "GRAPHTAP_SPMV");
    // default: propagation blocking with f64 messages. pb_f32msg (PageRank's messages rounded to f32 in fixed-count runs: 5e-8 of
    // relative rank error, tolerance 1e-6) stays opt-in: made the default in round 3 it flipped the sixth decimal of one printed
    // rank between two layouts (1.425146 / 1.425145) -- the mains must print the reference's lines digit for digit
    g->spmv_variant = (env && strcmp(env, "edge") == 0) ? GT_SPMV_EDGE : (env && strcmp(env, "pb_f32msg") == 0) ? GT_SPMV_PB_F32MSG : GT_SPMV_PB;
    st = gt_layout_build(g);
    if (st != GT_OK) { gt_scratch_release(); gt_graph_free(g); return st; }
    tick("layout of x", tb);
    if (g->spmv_variant != GT_SPMV_EDGE) {
        st = gt_pb_build(g);
        if (st != GT_OK) { gt_scratch_release(); gt_graph_free(g); return st; }
        tick("propagation-blocking streams", tb);
    }
    gt_scratch_release();   // the build's scratch pool goes back to the driver
    tick("scratch pool released", tb);
    *out = g;
    return GT_OK;
}

int gt_graph_select_spmv(gt_graph *g, int variant) {
    GT_REQUIRE(g, GT_ERR_INVALID, "null argument");
    GT_REQUIRE(variant >= GT_SPMV_EDGE && variant <= GT_SPMV_PB_F32MSG, GT_ERR_INVALID, "unknown SpMV variant %d", variant);
    const int before = g->spmv_variant;
    g->spmv_variant = variant;
    if (variant != GT_SPMV_EDGE) {   // builds what this variant needs and the graph does not have yet (the wide build for GT_SPMV_PB_F32MSG)
        int st = gt_pb_build(g);
        gt_scratch_release();
        if (st != GT_OK) { g->spmv_variant = before; return st; }
    }
    return GT_OK;
}

int gt_graph_has_wide_build(const gt_graph *g) { return g && g->pb_wide ? 1 : 0; }
int gt_graph_info_get(const gt_graph *g, gt_graph_info *info) {
    GT_REQUIRE(g && info, GT_ERR_INVALID, "null argument");
    *info = g->info;
    return GT_OK;
}
int gt_graph_vertex_ids(const gt_graph *g, uint32_t *host_out, uint64_t count) {
    GT_REQUIRE(g && host_out, GT_ERR_INVALID, "null argument");
    GT_REQUIRE(count <= g->info.tile_height, GT_ERR_INVALID, "count exceeds tile_height");
    const gt_vidmap vm = gt_vidmap_of(g);
    const uint64_t base = (uint64_t)g->info.rank * g->info.tile_height;
    for (uint64_t i = 0; i < count; i++) { const uint32_t v = gt_vid_of(vm, base + i); host_out[i] = v < g->info.nrows ? v : 0xFFFFFFFFu; }
    return GT_OK;
}
int gt_graph_tile(const gt_graph *g, gt_tile_arrays *a) {
    GT_REQUIRE(g && a, GT_ERR_INVALID, "null argument");
    a->JA = g->JA; a->IA = g->IA; a->A = g->A; a->JC = g->JC; a->IR = g->IR; a->L2G = g->loc2glob;
    return GT_OK;
}
int gt_graph_tile_cf(gt_graph *g, gt_tile_cf_arrays *a) {
    GT_REQUIRE(g && a, GT_ERR_INVALID, "null argument");
    return gt_tcsc_cf_arrays(g, a);
}
int gt_graph_exchange_plan(const gt_graph *g, gt_exchange_plan *plan) {
    GT_REQUIRE(g && plan, GT_ERR_INVALID, "null argument");
    plan->nranks = g->info.nranks; plan->x_slices = g->info.x_slices;
    plan->send_offset = g->send_off.data(); plan->recv_offset = g->recv_off.data();
    plan->send_counts = g->send_counts.data(); plan->recv_counts = g->recv_counts.data();
    return GT_OK;
}

// the caller's x (compressed-column order) in the slot order the kernels read, or x itself under the identity layout
static int x_in_slot_order(const gt_graph *g, const void **x_dev, uint32_t w, hipStream_t s) {
    if (!(g->xslot && g->info.nnzcols)) return GT_OK;
    gt_graph *gm = const_cast<gt_graph *>(g);
    if (!gm->x_scratch) GT_HIP(hipMalloc(&gm->x_scratch, (uint64_t)g->x_len * 8));
    GT_HIP(hipMemsetAsync(gm->x_scratch, 0, (uint64_t)g->x_len * w, s));   // unused slots are never referenced by an entry
    if (w == 8) k_to_slots<uint64_t><<<grid_for(g->info.nnzcols), TPB, 0, s>>>((const uint64_t *)*x_dev, g->xslot, g->info.nnzcols, (uint64_t *)gm->x_scratch);
    else k_to_slots<uint32_t><<<grid_for(g->info.nnzcols), TPB, 0, s>>>((const uint32_t *)*x_dev, g->xslot, g->info.nnzcols, (uint32_t *)gm->x_scratch);
    GT_HIP(hipGetLastError());
    *x_dev = gm->x_scratch;
    return GT_OK;
}

int gt_spmv_cf(gt_graph *g, const void *x_dev, void *y_dev, int first_iteration, int running, int last_iteration, void *hip_stream) {
    GT_REQUIRE(g && x_dev && y_dev, GT_ERR_INVALID, "null argument");
    int st = gt_tcsc_cf_build(g);   // before x is staged: a failed build leaves nothing behind
    if (st != GT_OK) return st;
    st = x_in_slot_order(g, &x_dev, 8, (hipStream_t)hip_stream);
    if (st != GT_OK) return st;
    return gt_tcsc_cf_spmv(g, (const double *)x_dev, (double *)y_dev, first_iteration != 0, running != 0, last_iteration != 0, (hipStream_t)hip_stream);
}

int gt_spmv(const gt_graph *g, int semiring, const void *x_dev, void *y_dev, void *hip_stream) {
    GT_REQUIRE(g && x_dev && y_dev, GT_ERR_INVALID, "null argument");
    int st = x_in_slot_order(g, &x_dev, (semiring == GT_PLUS_F64) ? 8 : 4, (hipStream_t)hip_stream);   // hubs-first layout: the kernels read x by slot
    if (st != GT_OK) return st;
    // the caller's x and y are f64: a bare SpMV keeps f64 messages whatever the graph's variant says about PageRank programs
    return gt_launch_spmv(g, semiring, x_dev, y_dev, (hipStream_t)hip_stream, false, nullptr, 0, 0, 0xFFFFFFFFu, 0, nullptr, false, true);
}

// ---- programs
int gt_program_free(gt_program *p) {
    if (!p) return GT_OK;
    void *ptrs[] = {p->d_tail, p->bu_first, p->fl_v[0], p->fl_v[1], p->fl_rows, p->row_mark, p->d_fl, p->s0, p->s1, p->rank, p->C, p->x_own, p->y, p->d_active, p->rank_c, p->deg_c, p->C_c, p->xseg, p->send_own,
                    p->fr_col, p->fr_val, p->fr_off, p->fr_tmp, p->d_frontier};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (p->h_pinned) (void)hipHostFree(p->h_pinned);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->slice_in) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->slice_done) (void)hipEventDestroy(e);
    for (hipStream_t st : p->slice_streams) (void)hipStreamDestroy(st);
    for (hipEvent_t e : p->part_done) (void)hipEventDestroy(e);
    for (hipStream_t st : p->part_streams) (void)hipStreamDestroy(st);
    if (p->p2_go) (void)hipEventDestroy(p->p2_go);
    delete p;
    return GT_OK;
}

int gt_program_set_options(gt_program *p, const gt_program_options *opt) {
    GT_REQUIRE(p && opt && opt->size >= 8, GT_ERR_INVALID, "gt_program_set_options: null argument, or size is not set (gt_program_options_init)");
    gt_program_options o; gt_program_options_init(&o);
    memcpy(&o, opt, std::min<size_t>(opt->size, sizeof(o)));
    auto flag = [&](int32_t v, const char *name) { if (v >= 0) p->cfg.set(name, v ? "1" : "0"); };
    flag(o.frontier_lists, "GRAPHTAP_FRONTIER_LISTS"); flag(o.spmspv, "GRAPHTAP_SPMSPV"); flag(o.tail_kernel, "GRAPHTAP_TAIL_KERNEL");
    flag(o.bfs_bottom_up, "GRAPHTAP_BFS_BOTTOM_UP"); flag(o.cc_first, "GRAPHTAP_CC_FIRST"); flag(o.fuse_apply, "GRAPHTAP_FUSE_APPLY");
    flag(o.lean_state, "GRAPHTAP_PR_LEAN_STATE"); flag(o.hybrid, "GRAPHTAP_HYBRID");
    if (o.spmspv_fraction) p->cfg.set("GRAPHTAP_SPMSPV_FRACTION", std::to_string(o.spmspv_fraction));
    if (o.tail_list_max) p->cfg.set("GRAPHTAP_TAIL_LIST", std::to_string(o.tail_list_max));
    if (o.tail_entries_max) p->cfg.set("GRAPHTAP_TAIL_ENTRIES", std::to_string(o.tail_entries_max));
    if (o.timeout_s > 0) p->timeout_s = o.timeout_s;
    // (the frontier lists are sized by initialize(): a program that switches them on afterwards gets them at its next initialize())
    return GT_OK;
}

int gt_program_create(gt_program **out, gt_graph *g, const gt_program_params *prm) {
    GT_REQUIRE(out && g && prm, GT_ERR_INVALID, "null argument");
    GT_REQUIRE(prm->kind >= GT_DEG && prm->kind <= GT_CC, GT_ERR_INVALID, "unknown program kind %d", prm->kind);
    GT_REQUIRE(prm->order == GT_ROW || (prm->order == GT_COL && prm->kind == GT_DEG), GT_ERR_UNSUPPORTED,
               "_COL_ ordering is implemented for the Degree program only (the reference says the same for TCSC_CF, vp:1319-1322)");
    GT_REQUIRE(prm->kind != GT_SSSP || g->info.weighted, GT_ERR_INVALID,
               "SSSP needs a weighted graph (12-byte records; the reference builds sssp with -DHAS_WEIGHT)");
    *out = nullptr;
    gt_program *p = new gt_program();
    p->g = g; p->prm = *prm;
    p->stationary = (prm->kind == GT_DEG || prm->kind == GT_PR);  // apps/*.cpp
    switch (prm->kind) {
        case GT_DEG: p->semiring = GT_PLUS_U32; break;
        case GT_PR:
            p->semiring = GT_PLUS_F64; p->y_bytes = 8;
            // f32 messages (GT_SPMV_PB_F32MSG) are for fixed iteration counts; in converge mode the program switches to f64 ones
            // (gt_program_prepare), so the buffers hold 8 bytes per element either way
            p->f32_capable = (g->spmv_variant == GT_SPMV_PB_F32MSG);
            p->x_f32 = p->f32_capable;
            p->x_bytes = p->x_f32 ? 4 : 8; p->x_alloc_bytes = 8;
            break;
        case GT_SSSP: p->semiring = GT_MINPLUS_U32; break;
        default: p->semiring = GT_MIN_U32; break;
    }
    const uint32_t H = g->info.tile_height;
    p->x_elems = g->x_len;
    p->y_elems = (prm->order == GT_COL) ? (uint64_t)g->info.nranks * g->info.seg_stride : g->info.nnzrows;
    bool ok = hipMalloc((void **)&p->s0, (uint64_t)H * 4) == hipSuccess && hipMalloc((void **)&p->C, H) == hipSuccess &&
              hipMalloc(&p->x_own, std::max<uint64_t>(p->x_elems, 1) * p->x_alloc_bytes) == hipSuccess &&
              hipMalloc(&p->y, std::max<uint64_t>(p->y_elems, 1) * p->y_bytes) == hipSuccess &&
              hipMalloc((void **)&p->d_active, sizeof(unsigned long long)) == hipSuccess && hipHostMalloc(&p->h_pinned, 256, hipHostMallocDefault) == hipSuccess;
    if (ok && prm->kind == GT_BFS) ok = hipMalloc((void **)&p->s1, (uint64_t)H * 4) == hipSuccess;
    {   // frontier lists of the min programs: the vertices the last apply changed (on several ranks: the rank's own, local ids;
        // the C++ driver turns them into (index, value) pairs for the peers and runs the SpMSpV from the pairs it receives, dist.hip)
        const char *fe = gt_cfg(p, "GRAPHTAP_FRONTIER_LISTS");
        p->fl_enabled = !p->stationary && prm->order == GT_ROW && !(fe && atoi(fe) == 0);
        if (ok && p->fl_enabled) {
            p->fl_rows_cap = std::max<uint32_t>(g->info.nnzrows, 1);
            p->fl_cap = std::max<uint32_t>(std::min<uint32_t>(H, GT_FRONTIER_CAP), 1);   // a list never holds more than the segment's H vertices
            ok = hipMalloc((void **)&p->fl_v[0], (uint64_t)p->fl_cap * 4) == hipSuccess && hipMalloc((void **)&p->fl_v[1], (uint64_t)p->fl_cap * 4) == hipSuccess &&
                 hipMalloc((void **)&p->fl_rows, (uint64_t)p->fl_rows_cap * 4) == hipSuccess &&
                 hipMalloc((void **)&p->row_mark, ((uint64_t)g->info.nnzrows / 32 + 1) * 4) == hipSuccess && hipMalloc((void **)&p->d_fl, 4 * sizeof(unsigned int)) == hipSuccess &&
                 hipMalloc(&p->d_tail, 64) == hipSuccess;
            if (ok && prm->kind == GT_BFS && !g->flags.directed && !gt_has_exchange(g))   // symmetric graph, whole on one rank: bottom-up steps are possible (kernels.hip)
            {   // + one bit per row, written 4096 rows at a time (k_bu_collect)
                const uint64_t nrw = std::max<uint32_t>(g->info.nnzrows, 1), bit_words = ((nrw + 4095) / 4096) * 128;
                ok = hipMalloc((void **)&p->bu_rows, (6 * nrw + 3 * bit_words) * 4) == hipSuccess;   // (the quads first: 16-byte aligned)
                if (ok) { p->bu_first = p->bu_rows; p->bu_rows = p->bu_first + 4 * nrw; p->bu_long = p->bu_rows + nrw; p->bu_bits = p->bu_long + nrw; p->bu_reached = p->bu_bits + bit_words;
                          p->bu_next = p->bu_reached + bit_words; p->bu_words = (uint32_t)bit_words; }
                if (ok && gt_bu_first_neighbours(g, p->bu_first, 0) != GT_OK) ok = false;
            }
        }
    }
    if (ok && prm->kind == GT_PR) {
        const uint64_t nr = std::max<uint32_t>(g->info.nnzrows, 1);
        ok = hipMalloc((void **)&p->rank, (uint64_t)H * 8) == hipSuccess && hipMalloc((void **)&p->rank_c, nr * 8) == hipSuccess &&
             hipMalloc((void **)&p->deg_c, nr * 4) == hipSuccess && hipMalloc((void **)&p->C_c, nr) == hipSuccess;
    }
    if (ok && gt_has_exchange(g))
        ok = hipMalloc(&p->xseg, std::max<uint64_t>(g->info.nnzcols, 1) * p->x_alloc_bytes) == hipSuccess &&
             hipMalloc(&p->send_own, std::max<uint64_t>(g->send_elems, 1) * p->x_alloc_bytes) == hipSuccess;
    if (!ok) { gt_program_free(p); gt_set_error("out of device memory for program state"); return GT_ERR_HIP; }
    p->x = p->x_own; p->send = p->send_own;
    *out = p;
    return GT_OK;
}

int gt_program_set_stream(gt_program *p, void *hip_stream) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    p->stream = (hipStream_t)hip_stream;
    return GT_OK;
}

// PageRank: V-space -> compressed working set (after any initialize) and back (before V is read)
static int pr_pack_state(gt_program *p) {
    if (p->prm.kind != GT_PR) return GT_OK;
    const uint32_t nr = p->g->info.nnzrows;
    if (nr) k_pr_pack<<<grid_for(nr), TPB, 0, p->stream>>>(p->g->IR, nr, p->rank, p->s0, p->C, p->rank_c, p->deg_c, p->C_c);
    GT_HIP(hipGetLastError());
    p->v_stale = false; p->x_fresh = false; p->y_clean = false;
    return GT_OK;
}
static int pr_sync_state(gt_program *p) {
    if (p->prm.kind != GT_PR || !p->v_stale) return GT_OK;
    const uint32_t nr = p->g->info.nnzrows;
    if (nr) k_pr_unpack<<<grid_for(nr), TPB, 0, p->stream>>>(p->g->IR, nr, p->rank_c, p->C_c, p->rank, p->C);
    GT_HIP(hipGetLastError());
    p->v_stale = false;
    return GT_OK;
}

static int init_common(gt_program *p) {
    const gt_graph *g = p->g;
    const uint32_t H = g->info.tile_height, base = g->info.rank * H;
    hipStream_t s = p->stream;
    static std::atomic<uint64_t> epoch_counter{0};   // unique across programs: a freed program's address may be reused
    p->iteration = 0; p->converged = false; p->check_sticky = false; p->init_epoch = ++epoch_counter; p->rowless_reset = false; p->x_fresh = false;
    if (p->f32_capable && !p->x_f32) {   // back to the f32 messages a converge-mode run had left (gt_program_prepare)
        // buffers a driver installed hold (and are exchanged as) the OTHER width: they are taken back, as prepare() requires of
        // its caller -- the driver asks for the buffers again (graphtap_amd/vertex_program.py drops its tensors in initialize())
        p->x = p->x_own; p->send = p->send_own;
        p->x_f32 = true; p->x_bytes = 4;
    }
    p->last_active = (p->prm.kind == GT_BFS || p->prm.kind == GT_SSSP) ? 1 : ~0ull;   // the root alone is active (bfs.h:37-50, sssp.h:33-42)
    switch (p->prm.kind) {
        case GT_DEG:  // deg.h:31-34
            GT_HIP(hipMemsetAsync(p->s0, 0, (uint64_t)H * 4, s));
            GT_HIP(hipMemsetAsync(p->C, 1, H, s));
            break;
        case GT_PR:   // PR_State default rank = alpha (pr.h:15-16); base initializer returns `stationary`
            GT_HIP(hipMemsetAsync(p->s0, 0, (uint64_t)H * 4, s));
            k_fill<double><<<grid_for(H), TPB, 0, s>>>(p->rank, H, p->prm.alpha);
            GT_HIP(hipMemsetAsync(p->C, 1, H, s));
            break;
        default:
            k_init_min<<<grid_for(H), TPB, 0, s>>>(p->prm.kind, H, base, gt_vidmap_of(g), p->prm.root, p->s0, p->s1, p->C);
            p->bu_maps_valid = false;
            {   // the root's place in this rank's segment, if it lives here (one rank: the vertex id itself)
                const uint64_t ur = (g->perm_mask == 0xFFFFFFFFu) ? p->prm.root : (uint64_t)((p->prm.root * g->perm_a) & g->perm_mask);
                p->root_here = p->prm.kind != GT_CC && p->prm.root < g->info.nrows && ur >= base && ur < (uint64_t)base + H;
                p->root_local = p->root_here ? (uint32_t)(ur - base) : 0u;
            }
            if (p->prm.kind == GT_BFS && p->bu_first) { int st = gt_bu_maps_init(p, s); if (st != GT_OK) return st; }
            if (p->fl_enabled) {   // the first frontier: the root alone (bfs.h:37-50, sssp.h:33-42), every vertex for CC (cc.h:33-36: no list)
                GT_HIP(hipMemsetAsync(p->row_mark, 0, ((uint64_t)g->info.nnzrows / 32 + 1) * 4, s));
                GT_HIP(hipMemsetAsync(p->d_fl, 0, 4 * sizeof(unsigned int), s));
                p->fl_cur = 0; p->fl_prev_valid = true; p->fl_prev_n = 0; p->fl_rows_valid = false; p->list_iters = 0;
                p->bottom_up_iters = 0; p->x_deferred = false; p->x_stale = false;
                p->fl_cur_valid = p->prm.kind != GT_CC; p->fl_cur_n = 0;
                // the root's place in this rank's segment, if it lives here (one rank: the vertex id itself)
                const uint64_t uroot = (g->perm_mask == 0xFFFFFFFFu) ? p->prm.root : (uint64_t)((p->prm.root * g->perm_a) & g->perm_mask);
                const bool root_here = p->prm.root < g->info.nrows && uroot >= base && uroot < (uint64_t)base + H;
                const uint32_t root_local = (uint32_t)(uroot - base);
                p->bfs_settled = root_here ? 1 : 0;
                if (p->fl_cur_valid && root_here) {
                    const unsigned int one = 1;
                    GT_HIP(hipMemcpyAsync(p->fl_v[0], &root_local, 4, hipMemcpyHostToDevice, s));
                    GT_HIP(hipMemcpyAsync(p->d_fl, &one, sizeof(one), hipMemcpyHostToDevice, s));
                    GT_HIP(hipStreamSynchronize(s));   // `one` is on this stack frame
                    p->fl_cur_n = 1;
                }
            }
            break;
    }
    // messages: padding columns are never referenced; give them (and the exchange buffers) the semiring's neutral message
    auto neutral = [&](void *buf, uint64_t n) {
        if (!buf || !n) return;
        if (p->x_bytes == 8) k_fill<double><<<grid_for(n), TPB, 0, s>>>((double *)buf, n, 0.0);
        else if (p->x_f32) k_fill<float><<<grid_for(n), TPB, 0, s>>>((float *)buf, n, 0.0f);
        else k_fill<uint32_t><<<grid_for(n), TPB, 0, s>>>((uint32_t *)buf, n, p->stationary ? 0u : GT_INF);
    };
    neutral(p->x, p->x_elems);
    neutral(p->xseg, g->info.nnzcols);
    if (p->xseg) neutral(p->send, g->send_elems);
    // accumulators: init_nonstationary fills y with infinity() (vp:625-635); stationary y is zeroed per combine
    if (!p->stationary) k_fill<uint32_t><<<grid_for(p->y_elems), TPB, 0, s>>>((uint32_t *)p->y, p->y_elems, GT_INF);
    GT_HIP(hipGetLastError());
    if (!p->stationary && p->prm.order == GT_ROW) {   // frontiers up to the list cap: no allocation inside the iteration loop
        int st = gt_spmspv_reserve(p, (uint32_t)std::min<uint64_t>(p->fl_enabled ? H : p->x_elems, p->fl_enabled ? p->fl_cap : (1u << 20)));
        if (st != GT_OK) return st;
    }
    if (g->spmv_variant != GT_SPMV_EDGE && g->pb && p->prm.order == GT_ROW) {   // the value stream of the SpMV this program runs: not inside execute()
        int st = gt_pb_reserve_val(g, p->prm.kind == GT_PR ? 8u : 4u, s);   // PageRank: both message widths (converge mode runs f64 ones)
        if (st != GT_OK) return st;
        if (!p->stationary) { st = gt_pb_claim_val_min(g, p, p->init_epoch, s); if (st != GT_OK) return st; }   // BFS / SSSP / CC: the neutral fill of the stream, here rather than in the first pass
    }
    { int st = gt_kernels_preload(s); if (st != GT_OK) return st; }
    while (p->ev.size() < 64) { hipEvent_t a; GT_HIP(hipEventCreate(&a)); p->ev.push_back(a); }   // SpMV timing pairs of the first 32 iterations: not created inside execute()
    p->initialized = true;
    return pr_pack_state(p);
}

int gt_program_initialize(gt_program *p) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    return init_common(p);
}

int gt_program_initialize_from(gt_program *p, const gt_program *other) {
    GT_REQUIRE(p && other, GT_ERR_INVALID, "null argument");
    GT_REQUIRE(p->g->info.tile_height == other->g->info.tile_height && p->g->info.rank == other->g->info.rank, GT_ERR_INVALID,
               "initialize(other): the two programs cover different vertex segments");
    int st = init_common(p);
    if (st != GT_OK) return st;
    if (p->prm.kind == GT_PR) {
        GT_REQUIRE(other->prm.kind == GT_DEG || other->prm.kind == GT_PR, GT_ERR_INVALID, "PageRank initializes from a Degree program (apps/pr.cpp:47-48)");
        const uint32_t H = p->g->info.tile_height;
        k_pr_take_degree<<<grid_for(H), TPB, 0, p->stream>>>(H, p->g->IJ, other->s0, p->s0);
        GT_HIP(hipGetLastError());
        st = pr_pack_state(p);
        if (st != GT_OK) return st;
    }
    // non-stationary programs ignore `other` (vp:489-493)
    return GT_OK;
}

// What every driver calls first in execute(iters) (vp:408-413): initialize() if nobody did, the sticky converge-mode flag, and --
// PageRank under GT_SPMV_PB_F32MSG -- the message width of the run. With f32 messages the iteration at which the last row
// stops "changing" depends on rounding noise (a hub's rank of ~1e4 carries ~1e-4 of it, above the reference's absolute
// tolerance 1e-5, pr.h:13), i.e. on the order a layout adds in; the reference's count is deterministic. So converge mode runs
// f64 messages (exactly the GT_SPMV_PB arithmetic), fixed counts keep the f32 ones. Call BEFORE asking for the x / send buffers.
int gt_program_prepare(gt_program *p, uint32_t iters) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    if (!p->initialized) { int st = init_common(p); if (st != GT_OK) return st; }   // vp:410-411
    if (iters == 0) p->check_sticky = true;                                         // vp:412-413 (never reset by the reference)
    if (p->prm.kind == GT_PR && p->f32_capable) {
        const bool want_f32 = !p->check_sticky;
        if (want_f32 != p->x_f32) {
            GT_REQUIRE(p->x == p->x_own && p->send == p->send_own, GT_ERR_STATE, "the message width changes with the mode of the run: restore the program's own x / send buffers first");
            p->x_f32 = want_f32; p->x_bytes = want_f32 ? 4 : 8; p->x_fresh = false;   // messages are recomputed from the ranks in the new width
            hipStream_t s = p->stream;
            GT_HIP(hipMemsetAsync(p->x_own, 0, std::max<uint64_t>(p->x_elems, 1) * p->x_alloc_bytes, s));   // 0 is the neutral message in both widths
            if (p->xseg) {
                GT_HIP(hipMemsetAsync(p->xseg, 0, std::max<uint64_t>(p->g->info.nnzcols, 1) * p->x_alloc_bytes, s));
                GT_HIP(hipMemsetAsync(p->send_own, 0, std::max<uint64_t>(p->g->send_elems, 1) * p->x_alloc_bytes, s));
            }
        }
    }
    return GT_OK;
}

int gt_program_x(gt_program *p, void **dev_ptr, uint64_t *elems, uint32_t *elem_bytes) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    if (dev_ptr) *dev_ptr = p->x;
    if (elems) *elems = p->x_elems;
    if (elem_bytes) *elem_bytes = p->x_bytes;
    return GT_OK;
}
int gt_program_set_x(gt_program *p, void *dev_ptr) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    p->x = dev_ptr ? dev_ptr : p->x_own;
    return GT_OK;
}
int gt_program_send(gt_program *p, void **dev_ptr, uint64_t *elems, uint32_t *elem_bytes) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    if (dev_ptr) *dev_ptr = p->xseg ? p->send : nullptr;
    if (elems) *elems = p->xseg ? p->g->send_elems : 0;
    if (elem_bytes) *elem_bytes = p->x_bytes;
    return GT_OK;
}
int gt_program_set_send(gt_program *p, void *dev_ptr) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    GT_REQUIRE(p->xseg, GT_ERR_STATE, "single-rank programs have no send buffer");
    p->send = dev_ptr ? dev_ptr : p->send_own;
    return GT_OK;
}
int gt_program_y(gt_program *p, void **dev_ptr, uint64_t *elems, uint32_t *elem_bytes) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    if (dev_ptr) *dev_ptr = p->y;
    if (elems) *elems = p->y_elems;
    if (elem_bytes) *elem_bytes = p->y_bytes;
    return GT_OK;
}
int gt_program_iteration(const gt_program *p, uint32_t *iteration) {
    GT_REQUIRE(p && iteration, GT_ERR_INVALID, "null argument");
    *iteration = p->iteration;
    return GT_OK;
}

int gt_min_messenger(gt_program *p) {
    const gt_graph *g = p->g;
    hipStream_t s = p->stream;
    void *xm = p->xseg ? p->xseg : p->x;
    if (p->fl_enabled && p->fl_cur_valid && p->fl_prev_valid && !p->x_stale) {   // both frontiers are lists: x changes in their slots only
        int st = gt_frontier_messages(p, s); if (st != GT_OK) return st;
    } else {
        k_msg_min<<<grid_for(gt_x_owned(g)), TPB, 0, s>>>((uint32_t *)xm, gt_x_vertex(g), gt_x_owned(g), p->C, p->s0,
                                                           g->info.rank * g->info.tile_height, gt_vidmap_of(g), p->prm.kind);
        p->x_stale = false;
    }
    p->x_deferred = false;
    GT_HIP(hipGetLastError());
    return GT_OK;
}

int gt_program_scatter_gather(gt_program *p) {
    GT_REQUIRE(p && p->initialized, GT_ERR_STATE, "scatter_gather before initialize");
    const gt_graph *g = p->g;
    if (p->prm.order == GT_COL || g->info.nnzcols == 0) return GT_OK;  // Deg/_COL_: messages are the constant 1, folded into combine
    const uint32_t nc = gt_x_owned(g);             // slots of the owned columns' messages
    const uint32_t *xv = gt_x_vertex(g);           // slot -> local vertex
    hipStream_t s = p->stream;
    void *xm = p->xseg ? p->xseg : p->x;   // several ranks: the owned columns' messages, packed per destination below
    if (p->x_fresh) p->x_fresh = false;   // the fused PageRank apply already wrote them
    else switch (p->prm.kind) {
        case GT_DEG: k_msg_deg<<<grid_for(nc), TPB, 0, s>>>((uint32_t *)xm, nc); break;
        case GT_PR: {
            int st = pr_sync_state(p); if (st != GT_OK) return st;
            if (p->x_f32) k_msg_pr<float><<<grid_for(nc), TPB, 0, s>>>((float *)xm, xv, nc, p->s0, p->rank);
            else k_msg_pr<double><<<grid_for(nc), TPB, 0, s>>>((double *)xm, xv, nc, p->s0, p->rank);
            break;
        }
        default:
            p->x_deferred = gt_bfs_bottom_up_likely(p) || gt_cc_first_likely(p) || gt_list_spmspv_likely(p);   // the step needs no messages: combine writes them if it is declined
            if (!p->x_deferred) { int st = gt_min_messenger(p); if (st != GT_OK) return st; }
            break;
    }
    if (p->xseg && g->send_elems && !p->pack_deferred) {   // (deferred: the driver packs slice by slice, gt_program_pack_slice)
        if (p->x_bytes == 8) k_pack_send<uint64_t><<<grid_for(g->send_elems), TPB, 0, s>>>((const uint64_t *)p->xseg, g->send_idx, g->send_elems, (uint64_t *)p->send);
        else k_pack_send<uint32_t><<<grid_for(g->send_elems), TPB, 0, s>>>((const uint32_t *)p->xseg, g->send_idx, g->send_elems, (uint32_t *)p->send);
    }
    GT_HIP(hipGetLastError());
    return GT_OK;
}
// the per-destination packing of slice k alone: the C++ driver sends slice k while slice k+1 is still being packed
int gt_program_pack_slice_on(gt_program *p, uint32_t k, hipStream_t s) {
    const gt_graph *g = p->g;
    if (!p->xseg || k >= g->info.x_slices) return GT_OK;
    const uint64_t lo = g->send_off[k], n = g->send_off[k + 1] - lo;
    if (!n) return GT_OK;
    if (p->x_bytes == 8) k_pack_send<uint64_t><<<grid_for(n), TPB, 0, s>>>((const uint64_t *)p->xseg, g->send_idx + lo, n, (uint64_t *)p->send + lo);
    else k_pack_send<uint32_t><<<grid_for(n), TPB, 0, s>>>((const uint32_t *)p->xseg, g->send_idx + lo, n, (uint32_t *)p->send + lo);
    GT_HIP(hipGetLastError());
    return GT_OK;
}
int gt_program_pack_slice(gt_program *p, uint32_t k) { return gt_program_pack_slice_on(p, k, p->stream); }

// PageRank with apply armed to follow this combine (gt_program_execute, gt_program_fuse_apply): the epilogue phase 2 runs
// for the row bins one workgroup owns. Returns false when nothing is to be fused.
static bool fused_epilogue(gt_program *p, gt_pr_epilogue *epi) {
    const gt_graph *g = p->g;
    if (!(p->fuse_armed && p->prm.kind == GT_PR && g->spmv_variant != GT_SPMV_EDGE && g->pb != nullptr)) return false;
    const bool cf = (p->prm.compression == GT_TCSC_CF);
    *epi = gt_pr_epilogue{p->rank_c, p->deg_c, p->C_c, gt_row_slot(g), p->xseg ? p->xseg : p->x, p->x_f32 ? 1 : 0, p->prm.alpha, p->prm.tol,
                          cf ? 1 : 0, (p->fuse_iters != 0 && p->iteration + 1 == p->fuse_iters) ? 1 : 0, p->fuse_count ? p->d_active : nullptr, p->pr_state};
    return true;
}

// the next event of the program's SpMV timing pairs
static int gt_timing_event(gt_program *p, hipStream_t s, hipEvent_t *out) {
    if (p->ev_used + 1 > p->ev.size()) {
        // the pairs recorded so far are folded into a running sum and their events used again (a drain of the stream every
        // 32 SpMVs; no event is created inside the iteration loop). Only at a pair boundary: a sliced SpMV holds one open.
        if ((p->ev_used & 1) == 0) {
            { int st = gt_stream_wait_deadline(s, "the iteration loop (timing events)", p->timeout_s); if (st != GT_OK) return st; }
            for (size_t i = 0; i + 1 < p->ev_used; i += 2) { float ms = 0; GT_HIP(hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1])); p->ev_acc_ms += ms; p->ev_acc_pairs++; }
            p->ev_used = 0;
        } else { hipEvent_t a; GT_HIP(hipEventCreate(&a)); p->ev.push_back(a); }
    }
    *out = p->ev[p->ev_used++];
    return GT_OK;
}

// slices [lo, hi) of the K = x_slices steps of one SpMV; the accumulators are complete after hi == K
static int combine_impl(gt_program *p, bool timed, uint32_t lo, uint32_t hi) {
    const gt_graph *g = p->g;
    hipStream_t s = p->stream;
    const uint32_t K = g->info.x_slices;
    if (p->converged) return GT_OK;  // vp:1025, 1044: nothing visible happens once converged
    if (p->prm.order == GT_COL) {
        if (hi < K) return GT_OK;
        if (g->loc2glob) GT_HIP(hipMemsetAsync(p->y, 0, p->y_elems * p->y_bytes, s));
        k_col_counts<<<grid_for(g->ncols_total), TPB, 0, s>>>(g->JA, g->ncols_total, g->loc2glob, (uint32_t *)p->y);
        GT_HIP(hipGetLastError());
        return GT_OK;
    }
    if (lo == 0) {
        if (p->stationary && !p->y_clean) GT_HIP(hipMemsetAsync(p->y, 0, p->y_elems * p->y_bytes, s));  // K13, vp:1026-1032
        p->y_clean = false;
    }
    auto timing_event = [&](hipEvent_t *out) -> int { return gt_timing_event(p, s, out); };
    const bool sliced = K > 1 && !(lo == 0 && hi >= K);
    if (!sliced) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timed) {
            int st = timing_event(&e0); if (st != GT_OK) return st;
            st = timing_event(&e1); if (st != GT_OK) return st;
            GT_HIP(hipEventRecord(e0, s));
        }
        gt_pr_epilogue epi{};
        const bool fuse = hi >= K && fused_epilogue(p, &epi);
        if (fuse && p->fuse_count) GT_HIP(hipMemsetAsync(p->d_active, 0, sizeof(unsigned long long), s));
        bool sparse_done = false;
        p->fl_rows_valid = false;
        if (!p->stationary && lo == 0 && hi >= K) { int st = gt_spmspv_try(p, s, &sparse_done); if (st != GT_OK) return st; }
        // TCSC_CF computation filtering (compressed_column.hpp:671-708, vp:1264-1317): the entries of source rows matter on
        // the last iteration only (never in converge mode, SURVEY trap 5); their chunks stay out until then
        const bool cf_last = p->fuse_iters != 0 && p->iteration + 1 == p->fuse_iters;
        const bool skip_source = p->prm.kind == GT_PR && p->prm.compression == GT_TCSC_CF && p->cf_hint && !cf_last && gt_pb_source_entries(g) != 0 &&
                                 !getenv("GRAPHTAP_NO_CF_FILTER");
        if (skip_source) p->cf_filtered++;
        if (p->x_deferred) {   // the messenger scatter_gather() left out: needed unless a pass that reads no messages did the SpMV
            if (sparse_done) { p->x_stale = true; p->x_deferred = false; }
            else { int st = gt_min_messenger(p); if (st != GT_OK) return st; }
        }
        // the edge-parallel baseline runs a GT_TCSC_CF PageRank over the format's own pair lists (vp:1243-1317)
        const bool cf_lists = g->spmv_variant == GT_SPMV_EDGE && p->prm.kind == GT_PR && p->prm.compression == GT_TCSC_CF && !gt_has_exchange(g) &&
                              !getenv("GRAPHTAP_NO_CF_FILTER");
        int st;
        if (cf_lists) {
            const bool last = !p->cf_hint || cf_last;   // stepped without a hint: source rows every time (only the last apply reads them)
            if (!last) p->cf_filtered++;
            st = gt_tcsc_cf_spmv(const_cast<gt_graph *>(g), (const double *)p->x, (double *)p->y, p->iteration == 0, true, last, s);
        } else {
            static const bool act_report = getenv("GRAPHTAP_WINDOW_ACTIVITY") != nullptr;
            if (act_report && !sparse_done && !p->stationary) { int st2 = gt_pb_window_activity_report(g, p->x, s, p->iteration); if (st2 != GT_OK) return st2; }
            st = sparse_done ? GT_OK : gt_launch_spmv(g, p->semiring, p->x, p->y, s, p->x_f32, p, p->init_epoch, lo, hi, 0, fuse ? &epi : nullptr, skip_source, p->f32_capable && !p->x_f32);
        }
        if (st != GT_OK) return st;
        p->fused = fuse;
        if (timed) { GT_HIP(hipEventRecord(e1, s)); if (hi >= K) p->spmv_done++; }
        return GT_OK;
    }
    // Slice by slice (the pipelined multi-GPU loop): launched back to back on ONE stream the K phase-1 kernels of a
    // tile-row of 8 of R-MAT-26 cost +0.03-0.05 ms at K = 2 and +0.08 ms at K = 4 over one launch (0.37 ms per step):
    // a slice's ~650 workgroups fill the 512 resident slots 1.3 times and every launch drains before the next starts.
    // So each slice goes to a helper stream that waits only for what the caller had enqueued on `stream` when the slice
    // was issued (its exchange), and phase 2 joins them: +0.025 ms at K = 2, +0.05 ms at K = 4 (tools/bench_tilerow.py --sliced, rocprofv3 timeline). Timing: one event pair per SpMV, first slice issued ->
    // phase 2 finished (exchange waits that were not hidden are inside).
    if (p->slice_streams.empty()) {
        const uint32_t ns = K < 3 ? K : 3;   // HIP maps streams onto 4 hardware queues: `stream` + 3 helpers run truly side by side
        for (uint32_t i = 0; i < ns; i++) { hipStream_t t; GT_HIP(hipStreamCreateWithFlags(&t, hipStreamNonBlocking)); p->slice_streams.push_back(t); }
        for (uint32_t i = 0; i < K; i++) {
            hipEvent_t a, b;
            GT_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming)); GT_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            p->slice_in.push_back(a); p->slice_done.push_back(b);
        }
    }
    if (lo == 0) {
        int st = gt_launch_spmv(g, p->semiring, p->x, p->y, s, p->x_f32, p, p->init_epoch, 0, 0, GT_PB_PREPARE, nullptr, false, p->f32_capable && !p->x_f32);
        if (st != GT_OK) return st;
        if (timed) { hipEvent_t e0; st = timing_event(&e0); if (st != GT_OK) return st; GT_HIP(hipEventRecord(e0, s)); }
    }
    for (uint32_t k = lo; k < hi && k < K; k++) {
        hipStream_t hs = p->slice_streams[k % p->slice_streams.size()];
        GT_HIP(hipEventRecord(p->slice_in[k], s));
        GT_HIP(hipStreamWaitEvent(hs, p->slice_in[k], 0));
        int st = gt_launch_spmv(g, p->semiring, p->x, p->y, hs, p->x_f32, p, p->init_epoch, k, k + 1, GT_PB_PHASE1, nullptr, false, p->f32_capable && !p->x_f32);
        if (st != GT_OK) return st;
        GT_HIP(hipEventRecord(p->slice_done[k], hs));
    }
    if (hi >= K && p->p2_by_parts) return GT_OK;   // the driver runs phase 2 part by part (gt_program_phase2_part)
    if (hi >= K) {
        for (uint32_t k = 0; k < K; k++) GT_HIP(hipStreamWaitEvent(s, p->slice_done[k], 0));
        gt_pr_epilogue epi{};
        const bool fuse = fused_epilogue(p, &epi);
        if (fuse && p->fuse_count) GT_HIP(hipMemsetAsync(p->d_active, 0, sizeof(unsigned long long), s));
        int st = gt_launch_spmv(g, p->semiring, p->x, p->y, s, p->x_f32, p, p->init_epoch, K, K, GT_PB_PHASE2, fuse ? &epi : nullptr, false, p->f32_capable && !p->x_f32);
        if (st != GT_OK) return st;
        p->fused = fuse;
        if (timed) { hipEvent_t e1; st = timing_event(&e1); if (st != GT_OK) return st; GT_HIP(hipEventRecord(e1, s)); p->spmv_done++; }
    }
    return GT_OK;
}
int gt_program_combine(gt_program *p) {
    GT_REQUIRE(p && p->initialized, GT_ERR_STATE, "combine before initialize");
    return combine_impl(p, p->timing, 0, p->g->info.x_slices);
}
int gt_program_combine_slice(gt_program *p, uint32_t k) {
    GT_REQUIRE(p && p->initialized, GT_ERR_STATE, "combine before initialize");
    GT_REQUIRE(k < p->g->info.x_slices, GT_ERR_INVALID, "slice %u of %u", k, p->g->info.x_slices);
    return combine_impl(p, p->timing, k, k + 1);
}
// ---- phase 2 PART BY PART (the pipelined multi-rank loop of a fixed-count PageRank, dist.hip). With K slices the phase-2 work
// list has K parts (pb.hip, gt_pb::work_part): part k holds the row bins whose rows' columns travel in slice k of the NEXT
// iteration's exchange. After part k -- its fused applicator, and the apply kernel over the split bins of the part -- the
// messages of slice k are final: the driver packs and sends them while parts k+1.. still run, instead of waiting for the whole
// applicator (the reference broadcasts a segment only after its whole apply, vp:843-862).
// gt_program_parts_begin: after gt_program_fuse_apply, before the combine_slice calls of the iteration; false = not possible
// for this program (the caller takes the ordinary loop).
bool gt_program_parts_begin(gt_program *p) {
    const gt_graph *g = p->g;
    p->p2_by_parts = false;
    // GRAPHTAP_P2_PARTS: 1 = on (parts one after the other on the compute stream), 2 = on, the parts side by side on streams of
    // their own; unset / 0 = off. BUILT, MEASURED AND LEFT OFF (round 4, profiles/r04/ab_phase2_parts_*.txt): on a tile-row of 8
    // of R-MAT-26 phase 2 is ONE round of workgroups (207 row bins on 256 CUs), so its duration is one workgroup's streaming
    // time and every part repeats it: 0.303 -> 0.363 ms of compute per step with 2 parts, 0.332 -> 0.547 with 4 (side by side:
    // 0.444 / 0.854); on the whole graph (1 821 workgroups, RCCL at world size 1, K = 4) the parts gave 2.09 -> 2.06 ms per step,
    // but moving the packing kernel to the communication stream alone gave 2.09 -> 1.96 in the ordinary loop, against which the
    // parts LOSE (2.04). The early slices buy less than the extra launches and the emptier chip cost.
    const char *e = getenv("GRAPHTAP_P2_PARTS");
    const int mode = e ? atoi(e) : 0;
    if (mode <= 0 || p->converged || p->prm.kind != GT_PR || p->prm.order == GT_COL || !p->fuse_armed || p->fuse_count || !g->pb || g->spmv_variant == GT_SPMV_EDGE) return false;
    // (nothing rank-local may enter this decision -- every rank of a run must take the same loop: a tile-row without entries has
    // its K empty parts like any other, pb.hip; the tile height is the same on every rank)
    if (g->info.x_slices < 2 || gt_pb_parts(g) != g->info.x_slices) return false;
    p->p2_by_parts = true;
    return true;
}
// `concurrent`: part k runs on a stream of its own (priority descending with k), all parts side by side -- a tile-row of 8 has
// ~200 row bins for 256 CUs, so K launches one after the other would leave most of the chip idle in each; side by side the
// workgroups of part 0 are dispatched first and part 0 completes when they do (about half way through phase 2), without
// making phase 2 any longer. part_done[k] is recorded behind part k; the caller's stream joins them after the last part.
int gt_program_phase2_part(gt_program *p, uint32_t k, uint32_t num_iterations, int concurrent) {
    GT_REQUIRE(p && p->initialized && p->p2_by_parts, GT_ERR_STATE, "phase2_part without gt_program_parts_begin");
    const gt_graph *g = p->g;
    hipStream_t s = p->stream;
    const uint32_t K = g->info.x_slices, nr = g->info.nnzrows;
    GT_REQUIRE(k < K, GT_ERR_INVALID, "part %u of %u", k, K);
    const bool cf = p->prm.compression == GT_TCSC_CF;
    if (p->part_done.size() < K) {
        int least = 0, greatest = 0;
        GT_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));   // numerically lower = higher priority
        while (p->part_done.size() < K) {
            hipEvent_t e; GT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->part_done.push_back(e);
            hipStream_t t; const int pr = std::min(least, greatest + (int)p->part_streams.size());
            GT_HIP(hipStreamCreateWithPriority(&t, hipStreamNonBlocking, pr)); p->part_streams.push_back(t);
        }
        GT_HIP(hipEventCreateWithFlags(&p->p2_go, hipEventDisableTiming));
    }
    if (k == 0) {
        for (uint32_t i = 0; i < K; i++) GT_HIP(hipStreamWaitEvent(s, p->slice_done[i], 0));   // every phase-1 slice has written its value-stream slots
        if (p->iteration == 0 && !cf) k_clear_empty_rows<<<grid_for(g->info.tile_height), TPB, 0, s>>>(p->C, g->IJ, g->info.tile_height);   // what apply() does first (vp:1641-1650)
        if (concurrent) GT_HIP(hipEventRecord(p->p2_go, s));
    }
    hipStream_t ps = concurrent ? p->part_streams[k] : s;
    if (concurrent) GT_HIP(hipStreamWaitEvent(ps, p->p2_go, 0));
    gt_pr_epilogue epi{};
    GT_REQUIRE(fused_epilogue(p, &epi), GT_ERR_STATE, "phase2_part: the fused applicator is not armed");
    int st = gt_launch_spmv(g, p->semiring, p->x, p->y, ps, p->x_f32, p, p->init_epoch, K, K, GT_PB_PHASE2, &epi, false, p->f32_capable && !p->x_f32, k, k + 1);
    if (st != GT_OK) return st;
    uint32_t nlist = 0;
    const uint32_t *list = gt_pb_split_bins_part(g, k, &nlist);   // the bins of this part that phase 2 did not apply itself
    if (nlist) {
        const int last = (num_iterations != 0) && (p->iteration + 1 == num_iterations);
        void *xm = p->xseg ? p->xseg : p->x;
        const uint64_t nwork = (uint64_t)nlist << GT_PB_ROW_BIN_BITS;
        if (p->x_f32) k_pr_apply_msg<float><<<grid_for(nwork), TPB, 0, ps>>>((double *)p->y, gt_row_slot(g), nr, p->rank_c, p->deg_c, p->C_c, (float *)xm, p->prm.alpha, p->prm.tol, cf, last, nullptr, list, nlist, p->pr_state);
        else k_pr_apply_msg<double><<<grid_for(nwork), TPB, 0, ps>>>((double *)p->y, gt_row_slot(g), nr, p->rank_c, p->deg_c, p->C_c, (double *)xm, p->prm.alpha, p->prm.tol, cf, last, nullptr, list, nlist, p->pr_state);
    }
    GT_HIP(hipGetLastError());
    GT_HIP(hipEventRecord(p->part_done[k], ps));   // the messages of slice k are final
    if (k + 1 == K) {   // the iteration is complete: what combine's end and apply() leave behind (apply_launch, apply_finish)
        if (concurrent) for (uint32_t i = 0; i < K; i++) GT_HIP(hipStreamWaitEvent(s, p->part_done[i], 0));
        if (p->timing) { hipEvent_t e1; st = gt_timing_event(p, s, &e1); if (st != GT_OK) return st; GT_HIP(hipEventRecord(e1, s)); p->spmv_done++; }
        p->fused = false; p->fuse_armed = false; p->cf_hint = false; p->p2_by_parts = false;
        p->v_stale = true; p->x_fresh = true; p->y_clean = true; p->pr_state = 0; p->last_active = ~0ull;
        p->iteration++;   // vp:421
    }
    return GT_OK;
}

static bool fuse_enabled(const gt_program *p) {
    const char *e = gt_cfg(p, "GRAPHTAP_FUSE_APPLY");
    return !(e != nullptr && atoi(e) == 0);   // on by default
}
int gt_program_fuse_apply(gt_program *p, uint32_t num_iterations, int want_active) {
    GT_REQUIRE(p && p->initialized, GT_ERR_STATE, "fuse_apply before initialize");
    p->fuse_armed = fuse_enabled(p) && p->prm.kind == GT_PR && !p->converged;
    p->fuse_iters = num_iterations; p->fuse_count = want_active != 0; p->cf_hint = true;
    return GT_OK;
}
int gt_program_enable_timing(gt_program *p, int on) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    p->timing = on != 0;
    return GT_OK;
}
int gt_program_timing(gt_program *p, double *spmv_ms, uint32_t *launches, int reset) {
    GT_REQUIRE(p && spmv_ms && launches, GT_ERR_INVALID, "null argument");
    GT_HIP(hipStreamSynchronize(p->stream));
    *spmv_ms = 0; *launches = 0;
    *spmv_ms = p->ev_acc_ms;
    for (size_t i = 0; i + 1 < p->ev_used; i += 2) {
        float ms = 0;
        GT_HIP(hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]));
        *spmv_ms += ms;
    }
    *launches = p->spmv_done;   // complete SpMVs (a sliced SpMV contributes one event pair per slice to the sum)
    if (reset) { p->ev_used = 0; p->spmv_done = 0; p->ev_acc_ms = 0; p->ev_acc_pairs = 0; }
    return GT_OK;
}

// apply() in two halves. apply_launch enqueues the kernels (the active count stays on the device, the vertices that change are
// appended to the next frontier list); apply_finish does the host bookkeeping once somebody has read the count. gt_program_apply
// reads it at once; the multi-rank driver (dist.hip) reads it together with the all-reduced convergence word -- one host round
// trip per iteration.
static int apply_launch(gt_program *p, uint32_t num_iterations, bool want_active, bool deferred, bool *list_from_flags_out) {
    const gt_graph *g = p->g;
    hipStream_t s = p->stream;
    const uint32_t nr = g->info.nnzrows, H = g->info.tile_height;
    int dummy_active = 0; int *active = want_active ? &dummy_active : nullptr;   // (only tested against null below)
    unsigned long long *d_active = active ? p->d_active : nullptr;   // counted only when the caller wants it (converge mode)
    const bool fused = p->fused;   // phase 2 already applied the rows of its single-workgroup bins (and counted them)
    p->fused = false; p->fuse_armed = false; p->cf_hint = false;
    GT_REQUIRE(!fused || ((active != nullptr) == p->fuse_count && num_iterations == p->fuse_iters), GT_ERR_STATE,
               "apply() after a fused combine must use the arguments the fusion was armed with");
    if (d_active && !fused) GT_HIP(hipMemsetAsync(d_active, 0, sizeof(unsigned long long), s));
    const bool cf = (p->prm.kind == GT_PR && p->prm.compression == GT_TCSC_CF);
    bool list_from_flags = false;
    if (p->iteration == 0 && !cf) {
        if (p->prm.kind == GT_BFS || p->prm.kind == GT_SSSP) { if (p->root_here) k_clear_if_empty_row<<<1, 1, 0, s>>>(p->C, g->IJ, p->root_local); }
        else k_clear_empty_rows<<<grid_for(H), TPB, 0, s>>>(p->C, g->IJ, H);
    }
    switch (p->prm.kind) {
        case GT_DEG:
            if (p->prm.order == GT_COL) {
                if (g->info.nnzcols)
                    k_apply_deg_col<<<grid_for(g->info.nnzcols), TPB, 0, s>>>((const uint32_t *)p->y + (uint64_t)g->info.rank * g->info.seg_stride,
                                                                              g->JC, g->info.nnzcols, p->s0, p->C);
            } else if (nr) k_apply_deg_row<<<grid_for(nr), TPB, 0, s>>>((const uint32_t *)p->y, g->IR, nr, p->s0, p->C);
            break;
        case GT_PR: {
            int last = (num_iterations != 0) && (p->iteration + 1 == num_iterations);
            void *xm = p->xseg ? p->xseg : p->x;   // next iteration's messages of the owned columns
            uint32_t nlist = 0;
            // after a fused combine: only the bins phase 2 left -- of the build that SpMV ran on (4-byte messages: the wide one, if the graph has it)
            const uint32_t *list = fused ? gt_pb_split_bins(g, &nlist, gt_pb_uses_wide(g, p->semiring, g->spmv_variant == GT_SPMV_PB_F32MSG && p->x_f32)) : nullptr;
            const uint64_t nwork = list ? (uint64_t)nlist << GT_PB_ROW_BIN_BITS : nr;
            if (nwork && p->x_f32)
                k_pr_apply_msg<float><<<grid_for(nwork), TPB, 0, s>>>((double *)p->y, gt_row_slot(g), nr, p->rank_c, p->deg_c, p->C_c, (float *)xm,
                                                                      p->prm.alpha, p->prm.tol, cf, last, d_active, list, nlist, p->pr_state);
            else if (nwork)
                k_pr_apply_msg<double><<<grid_for(nwork), TPB, 0, s>>>((double *)p->y, gt_row_slot(g), nr, p->rank_c, p->deg_c, p->C_c, (double *)xm,
                                                                       p->prm.alpha, p->prm.tol, cf, last, d_active, list, nlist, p->pr_state);
            p->v_stale = true; p->x_fresh = true; p->y_clean = true;
            p->pr_state = 0;   // the loop that owns the iterations sets it before every combine; a caller stepping by hand gets the full applicator
            break;
        }
        default: {   // BFS, SSSP, CC
            const bool bfs = p->prm.kind == GT_BFS;
            const bool lists = p->fl_enabled && active != nullptr;   // the host learns the length of the next list from the active count
            uint32_t *next = lists ? p->fl_v[p->fl_cur ^ 1] : nullptr;
            unsigned int *next_n = lists ? p->d_fl + (p->fl_cur ^ 1) : nullptr;
            if (lists) GT_HIP(hipMemsetAsync(next_n, 0, sizeof(unsigned int), s));
            if (lists && p->fl_rows_valid) {
                // the SpMSpV (or the bottom-up step) of this iteration left the rows it lowered: no other row can change; the
                // vertices that were active -- the current list, if there is one -- are the only ones whose flag is set
                const unsigned gl = (unsigned)std::max<uint64_t>(std::min<uint64_t>(((uint64_t)p->fl_cur_n + TPB - 1) / TPB, 4096), 1);
                if (!p->fl_cur_valid) GT_HIP(hipMemsetAsync(p->C, 0, H, s));
                else if (p->fl_cur_n) k_list_clear_flags<<<gl, TPB, 0, s>>>(p->C, p->fl_v[p->fl_cur], p->d_fl + p->fl_cur);
                const unsigned ga = 1024;   // the length is on the device; the rounds are uniform per workgroup
                const bool maps = bfs && p->bu_maps_valid && !p->bu_step_emitted;   // (a bottom-up step wrote the next level itself)
                if (maps) GT_HIP(hipMemsetAsync(p->bu_bits, 0, (size_t)p->bu_words * 4, s));   // the rows this apply reaches = the next level
                if (bfs) k_apply_list<true><<<ga, TPB, 0, s>>>(p->fl_rows, p->d_fl + 2, (const uint32_t *)p->y, g->IR, p->s0, p->s1, p->C, p->iteration, d_active, next, next_n, p->fl_cap,
                                                               maps ? p->bu_bits : nullptr, maps ? p->bu_reached : nullptr);
                else k_apply_list<false><<<ga, TPB, 0, s>>>(p->fl_rows, p->d_fl + 2, (const uint32_t *)p->y, g->IR, p->s0, p->s1, p->C, p->iteration, d_active, next, next_n, p->fl_cap);
                p->list_iters++;
            } else {
                // the full apply writes the next iteration's messages itself: every vertex with a row gets its slot rewritten (so x is
                // exact afterwards even if it was stale); the vertices without a row had their say in iteration 0 and are reset once
                static const bool fuse_msg = !(getenv("GRAPHTAP_FUSE_MIN_MSG") && atoi(getenv("GRAPHTAP_FUSE_MIN_MSG")) == 0);
                // (not for BFS where bottom-up steps are possible: the iterations after its full applies read no messages, A/B:
                // apply 0.34 -> 0.45 ms on R-MAT-25 for a messenger that was going to be skipped anyway)
                const bool msg = fuse_msg && nr && !(bfs && p->bu_rows);
                uint32_t *xm = (uint32_t *)(p->xseg ? p->xseg : p->x);
                const uint32_t vb = g->info.rank * g->info.tile_height;
                if (msg && !p->rowless_reset) {
                    k_reset_rowless_messages<<<grid_for(gt_x_owned(g)), TPB, 0, s>>>(xm, gt_x_vertex(g), gt_x_owned(g), g->IJ);
                    p->rowless_reset = true;
                }
                if (nr && bfs && msg) k_apply_rows<true, true><<<grid_for(nr), TPB, 0, s>>>((const uint32_t *)p->y, g->IR, nr, p->s0, p->s1, p->C, p->iteration, d_active, xm, gt_row_slot(g), vb, gt_vidmap_of(g));
                else if (nr && bfs) k_apply_rows<true, false><<<grid_for(nr), TPB, 0, s>>>((const uint32_t *)p->y, g->IR, nr, p->s0, p->s1, p->C, p->iteration, d_active, nullptr, nullptr, 0, gt_vidmap_of(g),
                                                                                          (p->bu_maps_valid && !p->bu_step_emitted) ? p->bu_bits : nullptr,
                                                                                          (p->bu_maps_valid && !p->bu_step_emitted) ? p->bu_reached : nullptr);
                else if (nr && msg) k_apply_rows<false, true><<<grid_for(nr), TPB, 0, s>>>((const uint32_t *)p->y, g->IR, nr, p->s0, p->s1, p->C, p->iteration, d_active, xm, gt_row_slot(g), vb, gt_vidmap_of(g));
                else if (nr) k_apply_rows<false, false><<<grid_for(nr), TPB, 0, s>>>((const uint32_t *)p->y, g->IR, nr, p->s0, p->s1, p->C, p->iteration, d_active, nullptr, nullptr, 0, gt_vidmap_of(g));
                if (msg) { p->x_fresh = true; p->x_stale = false; }
                list_from_flags = lists;   // once the count is known (apply_finish) -- or at once when nobody is going to read it first
                if (lists && deferred && nr)   // (the kernel stops appending at the list's capacity; the count tells whether the list is whole)
                    k_list_from_flags<<<(unsigned)std::min<uint64_t>(((uint64_t)nr + 4095) / 4096, 4096), TPB, 0, s>>>(g->IR, nr, p->C, next, next_n, p->fl_cap);
            }
            p->fl_rows_valid = false;
            break;
        }
    }
    GT_HIP(hipGetLastError());
    p->bu_step_emitted = false;
    p->iteration++;  // vp:421
    *list_from_flags_out = list_from_flags && !deferred;
    return GT_OK;
}
// `h`: the active count of the apply just launched (ignored unless it was counted)
static int apply_finish(gt_program *p, bool counted, unsigned long long h, bool list_from_flags) {
    const gt_graph *g = p->g;
    hipStream_t s = p->stream;
    const uint32_t nr = g->info.nnzrows;
    if (counted) {
        p->last_active = h;
        if (p->prm.kind == GT_BFS) p->bfs_settled += h;
    } else p->last_active = ~0ull;
    if (p->fl_enabled && !p->stationary) {   // the list of the vertices this apply changed becomes the current frontier
        p->fl_prev_valid = p->fl_cur_valid; p->fl_prev_n = p->fl_cur_n;
        p->fl_cur ^= 1;
        p->fl_cur_valid = counted && p->last_active <= p->fl_cap && (!list_from_flags || gt_frontier_list_worth(p, p->last_active));   // (a list the row-list apply appended exists already)
        p->fl_cur_n = p->fl_cur_valid ? (uint32_t)p->last_active : 0;
        if (p->fl_cur_valid && list_from_flags && nr && p->fl_cur_n)   // a full apply that changed few: collect them
            k_list_from_flags<<<(unsigned)std::min<uint64_t>(((uint64_t)nr + 4095) / 4096, 4096), TPB, 0, s>>>(g->IR, nr, p->C, p->fl_v[p->fl_cur], p->d_fl + p->fl_cur, p->fl_cap);
        GT_HIP(hipGetLastError());
    }
    return GT_OK;
}

int gt_read_back(gt_program *p, void *dst, const void *src_dev, size_t bytes, hipStream_t s) {
    static const bool plain = getenv("GRAPHTAP_PLAIN_READBACK") != nullptr;   // A/B: pageable copy + hipStreamSynchronize
    if (plain || !p->h_pinned || bytes > 256) {
        GT_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, s));
        return gt_stream_wait_deadline(s, "the per-iteration read-back", p->timeout_s);
    }
    GT_HIP(hipMemcpyAsync(p->h_pinned, src_dev, bytes, hipMemcpyDeviceToHost, s));
    { int st = gt_stream_wait_deadline(s, "the per-iteration read-back", p->timeout_s); if (st != GT_OK) return st; }
    memcpy(dst, p->h_pinned, bytes);
    return GT_OK;
}

// Spins on a stream with a deadline (GRAPHTAP_TIMEOUT_S, else GRAPHTAP_DIST_TIMEOUT_S, default 300 s; read at every call so
// that a test can shorten it): a kernel that never finishes ends execute() with GT_ERR_TIMEOUT and a message -- never a
// hang with a core at 100 %, never a re-exec. The stream is left as it is (its work may still complete); the caller's handles
// stay valid for gt_program_free / gt_graph_free only after the device has drained.
double gt_wait_limit_s(void) {
    const char *e = getenv("GRAPHTAP_TIMEOUT_S");
    if (!e) e = getenv("GRAPHTAP_DIST_TIMEOUT_S");
    const double v = e ? atof(e) : 300.0;
    return v > 0 ? v : 300.0;
}
int gt_stream_wait_deadline(hipStream_t s, const char *what, double limit_s) {
    const double limit = limit_s > 0 ? limit_s : gt_wait_limit_s();
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return GT_OK;
        if (e != hipErrorNotReady) { gt_set_error("stream failed during %s: %s", what, hipGetErrorString(e)); return GT_ERR_HIP; }
        if ((spins & 255u) == 255u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            gt_set_error("%s did not complete within %g s (GRAPHTAP_TIMEOUT_S / gt_program_options.timeout_s): a kernel that does not finish, or a device that is gone", what, limit);
            return GT_ERR_TIMEOUT;
        }
    }
}

int gt_program_apply(gt_program *p, uint32_t num_iterations, uint64_t *active) {
    GT_REQUIRE(p && p->initialized, GT_ERR_STATE, "apply before initialize");
    // converged: apply_*() is skipped (vp:1616-1632) but the driver still counts the trip (vp:421), so a later
    // execute(n) terminates
    if (p->converged) { p->iteration++; p->fused = false; p->fuse_armed = false; if (active) *active = 0; return GT_OK; }
    bool lff = false;
    int st = apply_launch(p, num_iterations, active != nullptr, false, &lff);
    if (st != GT_OK) return st;
    unsigned long long h = 0;
    if (active) {
        st = gt_read_back(p, &h, p->d_active, sizeof(h), p->stream); if (st != GT_OK) return st;
        *active = h;
    }
    return apply_finish(p, active != nullptr, h, lff);
}
// the two halves for a driver that reads the count itself (device word: gt_program_active_word); converge mode only
int gt_program_apply_begin(gt_program *p, uint32_t num_iterations) {
    GT_REQUIRE(p && p->initialized && !p->converged, GT_ERR_STATE, "apply_begin: program not running");
    bool lff = false;
    return apply_launch(p, num_iterations, true, true, &lff);
}
int gt_program_apply_end(gt_program *p, uint64_t active_local) { return apply_finish(p, true, active_local, false); }

int gt_program_finish_converged(gt_program *p) {
    GT_REQUIRE(p && p->initialized, GT_ERR_STATE, "finish before initialize");
    p->converged = true;
    if (p->prm.kind == GT_PR && p->prm.compression == GT_TCSC_CF && p->g->info.nnzrows) {
        k_pr_cf_tail_c<<<grid_for(p->g->info.nnzrows), TPB, 0, p->stream>>>(p->g->R2C, p->g->info.nnzrows, p->rank_c, p->prm.alpha);
        GT_HIP(hipGetLastError());
        p->v_stale = true;
    }
    return GT_OK;
}

int gt_program_execute(gt_program *p, uint32_t iters, gt_exec_stats *stats) {
    GT_REQUIRE(p, GT_ERR_INVALID, "null argument");
    GT_REQUIRE(!gt_has_exchange(p->g), GT_ERR_STATE,
               "gt_program_execute runs single-rank graphs; multi-rank runs drive scatter_gather/combine/apply with an exchange of x between them");
    { int st = gt_program_prepare(p, iters); if (st != GT_OK) return st; }   // vp:410-413 + the message width of this run
    const bool check = p->check_sticky;
    hipStream_t s = p->stream;
    p->ev_used = 0; p->spmv_done = 0; p->spmspv_iters = 0; p->cf_filtered = 0; p->list_iters = 0; p->tail_iters = 0; p->spmspv_allocs = 0; p->ev_acc_ms = 0; p->ev_acc_pairs = 0;
    { int st = gt_stream_wait_deadline(s, "the work queued before execute()", p->timeout_s); if (st != GT_OK) return st; }
    const uint32_t val_allocs0 = gt_pb_val_allocs(p->g);
    const size_t ev0 = p->ev.size();
    auto t0 = std::chrono::steady_clock::now();
    // GRAPHTAP_TIMING=1: drain the stream after every phase so that the three phase timers are device times
    // (the reference's -DTIMING build, vp:640-684, 1018-1054, 1611-1637); off by default: phases overlap host work.
    const bool phase_timing = stats != nullptr && getenv("GRAPHTAP_TIMING") != nullptr;
    const bool fuse_apply = fuse_enabled(p);
    double t_sg = 0, t_cb = 0, t_ap = 0, q_sg = 0, q_cb = 0, q_ap = 0;
    uint32_t samples = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](std::chrono::steady_clock::time_point &t, double &acc, double &sq) -> int {
        if (phase_timing) { int st = gt_stream_wait_deadline(s, "a phase of the iteration loop (GRAPHTAP_TIMING)", p->timeout_s); if (st != GT_OK) return st; }
        auto t2 = now(); const double ms = std::chrono::duration<double, std::milli>(t2 - t).count(); acc += ms; sq += ms * ms; t = t2;
        return GT_OK;
    };
    for (;;) {
        auto tp = now();
        int st = gt_program_scatter_gather(p); if (st != GT_OK) return st;
        st = lap(tp, t_sg, q_sg); if (st != GT_OK) return st;
        // PageRank: apply follows combine at once, so phase 2 may apply the rows whose sums it completes (pb.hip)
        p->fuse_armed = fuse_apply && p->prm.kind == GT_PR && !p->converged; p->fuse_iters = iters; p->fuse_count = check; p->cf_hint = true;
        p->pr_state = gt_pr_state_mode(p, iters, check);
        st = combine_impl(p, stats != nullptr, 0, p->g->info.x_slices); if (st != GT_OK) return st;
        st = lap(tp, t_cb, q_cb); if (st != GT_OK) return st;
        uint64_t active = 0;
        st = gt_program_apply(p, iters, check ? &active : nullptr); if (st != GT_OK) return st;
        st = lap(tp, t_ap, q_ap); if (st != GT_OK) return st;
        samples++;
        if (check) {
            if (active == 0) { st = gt_program_finish_converged(p); if (st != GT_OK) return st; break; }
            if (!p->stationary && !phase_timing) {   // a short list: the rest of the run -- or as much of it as stays short -- in one launch
                bool conv = false; uint32_t ran = 0;
                st = gt_tail_try(p, s, &conv, &ran); if (st != GT_OK) return st;
                if (ran) p->bu_maps_valid = false;   // iterations that ran inside the tail kernel did not keep the row bitmaps
                if (conv) { st = gt_program_finish_converged(p); if (st != GT_OK) return st; break; }
            }
        } else if (p->iteration >= iters) break;
    }
    { int st = gt_stream_wait_deadline(s, "the iteration loop of execute()", p->timeout_s); if (st != GT_OK) return st; }   // every wait of execute() has a deadline
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->iterations = p->iteration; stats->converged = p->converged;
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
        stats->scatter_gather_ms = t_sg; stats->combine_ms = t_cb; stats->apply_ms = t_ap;
        stats->scatter_gather_sq = q_sg; stats->combine_sq = q_cb; stats->apply_sq = q_ap; stats->phase_samples = samples;
        stats->fused_apply_rows = (fuse_apply && p->prm.kind == GT_PR && p->g->spmv_variant != GT_SPMV_EDGE) ? gt_pb_rows_single(p->g, gt_pb_uses_wide(p->g, p->semiring, p->g->spmv_variant == GT_SPMV_PB_F32MSG && p->x_f32)) : 0;
        stats->spmspv_iterations = p->spmspv_iters; stats->cf_filtered_iterations = p->cf_filtered; stats->list_iterations = p->list_iters;
        stats->allocs_in_execute = (gt_pb_val_allocs(p->g) - val_allocs0) + p->spmspv_allocs + (uint32_t)(p->ev.size() - ev0);
        stats->spmv_ms = p->ev_acc_ms; stats->spmv_launches = p->ev_acc_pairs;
        for (size_t i = 0; i + 1 < p->ev_used; i += 2) {
            float ms = 0;
            GT_HIP(hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]));
            stats->spmv_ms += ms; stats->spmv_launches++;
        }
    }
    return GT_OK;
}

int gt_program_copy_state(gt_program *p, int field, void *host_out, uint64_t count) {
    GT_REQUIRE(p && host_out, GT_ERR_INVALID, "null argument");
    GT_REQUIRE(count <= p->g->info.tile_height, GT_ERR_INVALID, "count exceeds tile_height");
    const void *src = nullptr; uint32_t w = 4;
    const int k = p->prm.kind;
    switch (field) {
        case GT_F_DEGREE: if (k == GT_DEG || k == GT_PR) src = p->s0; break;
        case GT_F_RANK: if (k == GT_PR) { src = p->rank; w = 8; } break;
        case GT_F_PARENT: if (k == GT_BFS) src = p->s0; break;
        case GT_F_HOPS: if (k == GT_BFS) src = p->s1; break;
        case GT_F_DISTANCE: if (k == GT_SSSP) src = p->s0; break;
        case GT_F_LABEL: if (k == GT_CC) src = p->s0; break;
        case GT_F_ACTIVE: src = p->C; w = 1; break;
        default: break;
    }
    GT_REQUIRE(src, GT_ERR_INVALID, "field %d does not exist for program kind %d", field, k);
    { int st = pr_sync_state(p); if (st != GT_OK) return st; }
    GT_HIP(hipStreamSynchronize(p->stream));
    GT_HIP(hipMemcpy(host_out, src, count * w, hipMemcpyDeviceToHost));
    return GT_OK;
}

int gt_program_checksum(gt_program *p, uint64_t *value_sum, uint64_t *reachable) {
    GT_REQUIRE(p && value_sum && reachable, GT_ERR_INVALID, "null argument");
    const gt_graph_info &i = p->g->info;
    const uint32_t H = i.tile_height;
    const uint64_t base = (uint64_t)i.rank * H;
    const gt_vidmap vm = gt_vidmap_of(p->g);
    uint64_t s = 0, c = 0;
    { int st = pr_sync_state(p); if (st != GT_OK) return st; }
    GT_HIP(hipStreamSynchronize(p->stream));
    if (p->prm.kind == GT_PR) {  // get_state() = rank (pr.h:17), infinity() = 0 (vp:40)
        std::vector<double> h(H);
        GT_HIP(hipMemcpy(h.data(), p->rank, (uint64_t)H * 8, hipMemcpyDeviceToHost));
        for (uint32_t v = 0; v < H; v++)
            if (h[v] != 0.0 && gt_vid_of(vm, base + v) < i.nrows) { s = (uint64_t)((double)s + h[v]); c++; }
    } else {
        const uint32_t inf = (p->prm.kind == GT_DEG) ? 0u : GT_INF;
        const uint32_t *src = (p->prm.kind == GT_BFS) ? p->s1 : p->s0;  // BFS get_state() = hops (bfs.h:27)
        std::vector<uint32_t> h(H);
        GT_HIP(hipMemcpy(h.data(), src, (uint64_t)H * 4, hipMemcpyDeviceToHost));
        for (uint32_t v = 0; v < H; v++)
            if (h[v] != inf && gt_vid_of(vm, base + v) < i.nrows) { s += h[v]; c++; }
    }
    *value_sum = s; *reachable = c;
    return GT_OK;
}

}  // extern "C"
