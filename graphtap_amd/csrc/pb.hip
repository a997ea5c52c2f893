// pb.hip -- "propagation blocking" SpMV for the owned tile-row: the production kernel pair.
//
// Same result as the edge-parallel kernel in kernels.hip (and as the reference's column-major
// loop, src/vp/vertex_program.hpp:1162-1173 / 1490-1503), organised so that every HBM access is a
// stream and every random access hits LDS:
//
//   phase 1  "scatter"   one workgroup per CHUNK of the column-major entry stream (a window of
//            W = 8192 consecutive compressed columns, split every CH entries): the window's messages
//            x[col0 .. col0+W) are staged in LDS (coalesced load), then for every entry, in
//            (chunk, row-bin) order, VAL[k] = x[col] (+ w) is written at the entry's slot k of the
//            row-bin-major value stream. Reads 2 B/entry (window-local column + head flag),
//            writes F B/entry, in runs of ~one (chunk, bin) segment.
//   phase 2  "gather"    one workgroup per ROW BIN (R = 16384 consecutive compressed rows; heavy
//            bins are split by entry count): the bin's partial accumulators live in LDS (R x F =
//            128 KiB for f64), the bin's slice of VAL and of the static bin-local row ids LROW is
//            streamed once and combined with LDS atomics (ds_add_f64 / ds_min_u32), then the
//            accumulators are merged into y (plain RMW when the bin has one workgroup, device
//            atomics when it was split).
//
// Static per-graph data (built once on the device by gt_pb_build with rocPRIM sorts/scans):
//   LCOL[v]  u16  v-order = entries sorted by (chunk, bin, col, row); low 13 bits: col - col0,
//                 bit 15: first entry of a (chunk, bin) segment
//   LROW[k]  u16  k-order = segments sorted by (bin, chunk); row & (R-1)
//   WT[v]    u32  weights in v-order (min-plus only)
//   KSTART[s], G[g]: where segment s starts in k-order; for every group of 64 entries of the
//                 v-order the k-slot of its first entry and of its first segment head.
// HBM traffic per entry per SpMV: 2 + F (phase 1) + F + 2 (phase 2) + ~0.2 -> 20.2 B for f64,
// 12.2 B for u32, against the 4.6 / 4.4 B of the algorithmic minimum.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gt_internal.h"

namespace {

constexpr int RB = 14;                 // log2 rows per bin
constexpr uint32_t R = 1u << RB;       // 16384 rows: 128 KiB of f64 accumulators in LDS
constexpr uint32_t W = 8192;           // columns per window: 64 KiB of f64 messages in LDS
constexpr uint32_t CH_DEFAULT = 1u << 18;
constexpr uint32_t EPW = 1u << 18;     // entries per phase-2 workgroup
constexpr int P1_THREADS = 1024;
constexpr int P2_THREADS = 1024;
constexpr uint16_t HEAD = 0x8000;
constexpr int TPB = 256;

inline unsigned grid_for(uint64_t n) {
    uint64_t b = (n + TPB - 1) / TPB;
    if (b < 1) b = 1;
    if (b > 256u * 32u) b = 256u * 32u;
    return (unsigned)b;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(uint64_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
    template <class T> T *as() { return (T *)p; }
};

// ------------------------------------------------------------------ build kernels
__global__ void k_win_counts(const uint32_t *__restrict__ JA, uint32_t ncols, uint32_t nwin, uint32_t ch, uint32_t *__restrict__ nsub) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nwin; q += gridDim.x * blockDim.x) {
        uint64_t c0 = (uint64_t)q * W, c1 = c0 + W;
        uint32_t e0 = JA[c0 < ncols ? c0 : ncols], e1 = JA[c1 < ncols ? c1 : ncols];
        nsub[q] = (e1 - e0 + ch - 1) / ch;
    }
}
__global__ void k_fill_chunks(const uint32_t *__restrict__ JA, uint32_t ncols, uint32_t nwin, uint32_t ch,
                              const uint32_t *__restrict__ cbase, uint32_t *__restrict__ cv0, uint32_t *__restrict__ cv1,
                              uint32_t *__restrict__ ccol0) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nwin; q += gridDim.x * blockDim.x) {
        uint64_t c0 = (uint64_t)q * W, c1 = c0 + W;
        uint32_t e0 = JA[c0 < ncols ? c0 : ncols], e1 = JA[c1 < ncols ? c1 : ncols];
        uint32_t n = (e1 - e0 + ch - 1) / ch, base = cbase[q];
        for (uint32_t s = 0; s < n; s++) {
            uint64_t a = (uint64_t)e0 + (uint64_t)s * ch, b = a + ch;
            cv0[base + s] = (uint32_t)a; cv1[base + s] = (uint32_t)(b < e1 ? b : e1); ccol0[base + s] = (uint32_t)c0;
        }
    }
}
// sort key of every entry: (chunk << binbits) | row bin
__global__ void k_keys(const uint32_t *__restrict__ cv0, const uint32_t *__restrict__ cv1, const uint32_t *__restrict__ IA,
                       int binbits, uint32_t *__restrict__ key, uint32_t *__restrict__ idx) {
    const uint32_t c = blockIdx.x;
    for (uint64_t e = (uint64_t)cv0[c] + threadIdx.x; e < cv1[c]; e += blockDim.x) {
        key[e] = (c << binbits) | (IA[e] >> RB);
        idx[e] = (uint32_t)e;
    }
}
__global__ void k_chunk_row_keys(const uint32_t *__restrict__ cv0, const uint32_t *__restrict__ cv1, const uint32_t *__restrict__ IA,
                                 uint64_t *__restrict__ key) {
    const uint32_t c = blockIdx.x;
    for (uint64_t e = (uint64_t)cv0[c] + threadIdx.x; e < cv1[c]; e += blockDim.x) key[e] = ((uint64_t)c << 32) | IA[e];
}
__global__ void k_count_unique64(const uint64_t *__restrict__ key, uint64_t n, unsigned long long *__restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        c += (i == 0 || key[i] != key[i - 1]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
__global__ void k_iota(uint32_t *__restrict__ p, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}
__global__ void k_heads(const uint32_t *__restrict__ key, uint64_t n, uint32_t *__restrict__ head) {
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x)
        head[v] = (v == 0 || key[v] != key[v - 1]) ? 1u : 0u;
}
// sid[v] = inclusive scan of head; segment s = sid-1
__global__ void k_segments(const uint32_t *__restrict__ key, const uint32_t *__restrict__ sid, uint64_t n, uint32_t binmask,
                           uint32_t *__restrict__ vstart, uint32_t *__restrict__ segbin) {
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        if (v == 0 || key[v] != key[v - 1]) { uint32_t s = sid[v] - 1; vstart[s] = (uint32_t)v; segbin[s] = key[v] & binmask; }
    }
}
__global__ void k_seg_len_sorted(const uint32_t *__restrict__ order, const uint32_t *__restrict__ vstart, uint32_t nseg, uint32_t nnz,
                                 uint32_t *__restrict__ len_sorted) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nseg; t += gridDim.x * blockDim.x) {
        uint32_t s = order[t];
        len_sorted[t] = (s + 1 < nseg ? vstart[s + 1] : nnz) - vstart[s];
    }
}
__global__ void k_kstart(const uint32_t *__restrict__ order, const uint32_t *__restrict__ kscan, uint32_t nseg, uint32_t *__restrict__ kstart) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nseg; t += gridDim.x * blockDim.x) kstart[order[t]] = kscan[t];
}
__global__ void k_bin_offsets(const uint32_t *__restrict__ bins_sorted, const uint32_t *__restrict__ kscan, uint32_t nseg, uint32_t nnz,
                              uint32_t nbins, uint32_t *__restrict__ bin_off) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= nbins; b += gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nseg;
        while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (bins_sorted[mid] < b) lo = mid + 1; else hi = mid; }
        bin_off[b] = lo < nseg ? kscan[lo] : nnz;
    }
}
__global__ void k_static_streams(const uint32_t *__restrict__ key, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ sid,
                                 uint64_t n, int binbits, const uint32_t *__restrict__ ccol0, const uint32_t *__restrict__ IA,
                                 const uint32_t *__restrict__ JI, const uint32_t *__restrict__ A, const uint32_t *__restrict__ vstart,
                                 const uint32_t *__restrict__ kstart, uint16_t *__restrict__ LCOL, uint16_t *__restrict__ LROW,
                                 uint32_t *__restrict__ WT) {
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t e = idx[v], c = key[v] >> binbits, s = sid[v] - 1;
        bool head = (v == 0 || key[v] != key[v - 1]);
        LCOL[v] = (uint16_t)((JI[e] - ccol0[c]) | (head ? HEAD : 0));
        uint32_t k = kstart[s] + ((uint32_t)v - vstart[s]);
        LROW[k] = (uint16_t)(IA[e] & (R - 1));
        if (WT) WT[v] = A[e];
    }
}
// Group table: for every 64 consecutive entries of the v-order, where lane 0 lands in the k-order
// (k0), where the first six segment heads among lanes 1..63 land (k[0..5]) and the segment of lane 0
// (s, for the rare groups with seven or more heads). One 32-byte scalar load per group in phase 1,
// so that no load of phase 1 depends on another load.
struct GroupRec { uint32_t k0, k[6], s; };   // 32 bytes: one s_load_dwordx8 per group
__global__ void k_group_table(const uint32_t *__restrict__ key, const uint32_t *__restrict__ sid, uint64_t n, uint32_t nseg,
                              const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ kstart, GroupRec *__restrict__ G) {
    const uint64_t ngroups = (n + 63) / 64;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = g * 64;
        const uint32_t s0 = sid[v] - 1;
        GroupRec r;
        r.s = s0; r.k0 = kstart[s0] + ((uint32_t)v - vstart[s0]);
        // segments are numbered in v-order: the i-th head among lanes 1..63 opens segment s0 + i
        for (int i = 0; i < 6; i++) r.k[i] = (s0 + 1 + i < nseg) ? kstart[s0 + 1 + i] : 0;
        G[g] = r;
    }
}

// ------------------------------------------------------------------ phase 1
// T  = type of x / y / the LDS accumulators (double or uint32_t)
// TV = type of the value stream VAL and of the LDS message window: T, or float for the
//      "f32 messages" PageRank variant (messages rounded to f32, sums still accumulated in f64)
template <class T, class TV> struct Msg;
template <> struct Msg<double, double> { static __device__ __forceinline__ double val(double x, uint32_t) { return x; } };
template <> struct Msg<double, float> { static __device__ __forceinline__ float val(float x, uint32_t) { return x; } };
template <> struct Msg<uint32_t, uint32_t> { static __device__ __forceinline__ uint32_t val(uint32_t x, uint32_t w) { return x == GT_INF ? GT_INF : x + w; } };

template <class T, class TV, class TX, bool WEIGHTED>
__global__ void __launch_bounds__(P1_THREADS) k_pb_scatter(const uint32_t *__restrict__ cv0, const uint32_t *__restrict__ cv1,
                                                           const uint32_t *__restrict__ ccol0, uint32_t ncols, uint32_t nnz,
                                                           const uint16_t *__restrict__ LCOL, const uint32_t *__restrict__ WT,
                                                           const uint32_t *__restrict__ KSTART, const GroupRec *__restrict__ G,
                                                           const TX *__restrict__ x, TV *__restrict__ VAL) {
    __shared__ TV xwin[W];
    const uint32_t c = blockIdx.x;
    const uint32_t v0 = cv0[c], v1 = cv1[c], col0 = ccol0[c];
    const uint32_t wn = (ncols - col0 < W) ? ncols - col0 : W;
    {   // stage the window: all loads of a lane in flight together
        constexpr int PER = W / P1_THREADS;
        TX t[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) { const uint32_t j = threadIdx.x + i * P1_THREADS; t[i] = (j < wn) ? x[col0 + j] : TX(0); }
#pragma unroll
        for (int i = 0; i < PER; i++) xwin[threadIdx.x + i * P1_THREADS] = (TV)t[i];
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr uint32_t NW = P1_THREADS / 64;
    constexpr int U = 8;   // 64-entry groups in flight per wave
    const uint64_t lane_le = (lane == 63) ? ~0ull : ((2ull << lane) - 1);  // lanes 0..lane
    const uint32_t gend = (uint32_t)(((uint64_t)v1 + 63) >> 6);
    const uint32_t *__restrict__ Gw = reinterpret_cast<const uint32_t *>(G);
    // Groups of 64 entries are aligned to 64 in v-space; a group at a chunk border is visited by both
    // chunks, each storing only its own lanes. Software pipeline: the loads of trip t+1 are issued
    // BEFORE the stores of trip t, so a wave does not wait on its own store acknowledgements (vmcnt
    // retires in order) to see its next inputs.
    uint16_t lc[U], nlc[U]; uint32_t gw[U], ngw[U], w[U], nw[U];
    auto issue_loads = [&](uint32_t g0, uint16_t (&olc)[U], uint32_t (&ogw)[U], uint32_t (&ow)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t g = (g0 + u < gend) ? g0 + u : gend - 1;
            const uint64_t v = (uint64_t)g * 64 + lane;
            olc[u] = (v < nnz) ? LCOL[v] : (uint16_t)0;                // 2 B/lane
            ogw[u] = Gw[(uint64_t)g * 8 + (lane & 7)];                  // lane i holds dword i & 7 of the 32-byte group record
            if constexpr (WEIGHTED) ow[u] = (v >= v0 && v < v1) ? WT[v] : 0u; else ow[u] = 0;
        }
    };
    uint32_t g0 = (v0 >> 6) + wave * U;
    if (g0 < gend) issue_loads(g0, lc, gw, w);
    while (g0 < gend) {
        const uint32_t gn = g0 + NW * U;
        if (gn < gend) issue_loads(gn, nlc, ngw, nw);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t g = g0 + u;
            if (g >= gend) break;
            const uint64_t v = (uint64_t)g * 64 + lane;
            const uint64_t heads = __ballot((lc[u] & HEAD) != 0) & lane_le & ~1ull;  // heads in lanes 1..lane
            const uint32_t cnt = __popcll(heads);
            const uint32_t hpos = heads ? 63 - __clzll(heads) : 0;
            // dword 0 = k of lane 0; dword i (1..6) = k of the i-th head; dword 7 = segment of lane 0
            const uint32_t ks = __shfl(gw[u], cnt < 7 ? cnt : 7);
            const bool mine = (v >= v0 && v < v1);
            const bool rare = (cnt >= 7);   // seven or more runs inside 64 entries: < 1 % of the entries
            const TV val = Msg<T, TV>::val(xwin[lc[u] & (W - 1)], w[u]);
            if (__ballot(rare) != 0) {      // wave-uniform branch: the dependent load and its wait stay in here
                if (rare && mine) VAL[KSTART[ks + cnt] + (lane - hpos)] = val;
            }
            if (!rare && mine) VAL[ks + (lane - hpos)] = val;
        }
        g0 = gn;
#pragma unroll
        for (int u = 0; u < U; u++) { lc[u] = nlc[u]; gw[u] = ngw[u]; w[u] = nw[u]; }
    }
}

// ------------------------------------------------------------------ phase 2
struct BinWork { uint32_t bin, k0, k1, single; };

template <class T, bool IS_MIN> __device__ __forceinline__ void lds_combine(T *acc, uint32_t r, T a) {
    if constexpr (IS_MIN) { if (a != GT_INF) atomicMin(&acc[r], a); }
    else if constexpr (sizeof(T) == 8) unsafeAtomicAdd(&acc[r], a);   // ds_add_f64
    else atomicAdd(&acc[r], a);
}

template <class T, class TV, bool IS_MIN>
__global__ void __launch_bounds__(P2_THREADS) k_pb_gather(const BinWork *__restrict__ work, const uint16_t *__restrict__ LROW,
                                                          const TV *__restrict__ VAL, uint32_t nrows, T *__restrict__ y) {
    __shared__ T acc[R];
    const BinWork wk = work[blockIdx.x];
    const T neutral = IS_MIN ? (T)GT_INF : (T)0;
    for (uint32_t i = threadIdx.x; i < R; i += P2_THREADS) acc[i] = neutral;
    __syncthreads();
    // Main loop: 4 consecutive entries per lane per load (16-byte VAL loads for f32/u32 streams, 2 x 16 B for
    // f64; 8-byte LROW loads), two such quads in flight per lane; scalar head and tail around the 4-aligned body.
    const uint64_t k0 = wk.k0, k1 = wk.k1;
    const uint64_t ka = (k0 + 3) & ~3ull, kb = k1 & ~3ull;   // aligned body [ka, kb)
    if (ka >= kb) {
        for (uint64_t k = k0 + threadIdx.x; k < k1; k += P2_THREADS) lds_combine<T, IS_MIN>(acc, LROW[k], (T)VAL[k]);
    } else {
        if (k0 + threadIdx.x < ka) lds_combine<T, IS_MIN>(acc, LROW[k0 + threadIdx.x], (T)VAL[k0 + threadIdx.x]);
        if (kb + threadIdx.x < k1) lds_combine<T, IS_MIN>(acc, LROW[kb + threadIdx.x], (T)VAL[kb + threadIdx.x]);
        struct alignas(8) R4 { uint16_t r[4]; };
        struct alignas(4 * sizeof(TV) > 16 ? 16 : 4 * sizeof(TV)) V4 { TV a[4]; };
        const R4 *__restrict__ LR4 = reinterpret_cast<const R4 *>(LROW);
        const V4 *__restrict__ VA4 = reinterpret_cast<const V4 *>(VAL);
        const uint64_t qa = ka >> 2, qb = kb >> 2;
        uint64_t q = qa + threadIdx.x;
        for (; q + P2_THREADS < qb; q += 2ull * P2_THREADS) {
            const R4 r0 = LR4[q], r1 = LR4[q + P2_THREADS];
            const V4 a0 = VA4[q], a1 = VA4[q + P2_THREADS];
#pragma unroll
            for (int j = 0; j < 4; j++) lds_combine<T, IS_MIN>(acc, r0.r[j], (T)a0.a[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) lds_combine<T, IS_MIN>(acc, r1.r[j], (T)a1.a[j]);
        }
        if (q < qb) {
            const R4 r0 = LR4[q]; const V4 a0 = VA4[q];
#pragma unroll
            for (int j = 0; j < 4; j++) lds_combine<T, IS_MIN>(acc, r0.r[j], (T)a0.a[j]);
        }
    }
    __syncthreads();
    const uint32_t row0 = wk.bin << RB;
    const uint32_t rn = (nrows - row0 < R) ? nrows - row0 : R;
    for (uint32_t i = threadIdx.x; i < rn; i += P2_THREADS) {
        T a = acc[i];
        if (a == neutral) continue;
        if (wk.single) {
            if constexpr (IS_MIN) { if (a < y[row0 + i]) y[row0 + i] = a; }
            else y[row0 + i] += a;
        } else {
            if constexpr (IS_MIN) atomicMin(&y[row0 + i], a);
            else if constexpr (sizeof(T) == 8) unsafeAtomicAdd(&y[row0 + i], a);
            else atomicAdd(&y[row0 + i], a);
        }
    }
}

}  // namespace

struct gt_pb {
    uint32_t nbins = 0, nchunks = 0, nwork = 0, nnz = 0;
    uint32_t *cv0 = nullptr, *cv1 = nullptr, *ccol0 = nullptr;
    uint16_t *LCOL = nullptr, *LROW = nullptr;
    uint32_t *WT = nullptr, *KSTART = nullptr;
    void *G = nullptr;  // GroupRec per 64 entries
    BinWork *work = nullptr;
    void *VAL = nullptr;       // value stream scratch, 8 B/entry once an f64 SpMV ran, else 4 B/entry
    uint32_t val_bytes = 0;
};

void gt_pb_free(gt_pb *pb) {
    if (!pb) return;
    void *ptrs[] = {pb->cv0, pb->cv1, pb->ccol0, pb->LCOL, pb->LROW, pb->WT, pb->KSTART, pb->G, pb->work, pb->VAL};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete pb;
}

#define PB_HIP(call)                                                                                \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            gt_set_error("pb build: %s failed: %s (line %d)", #call, hipGetErrorString(e_), __LINE__); \
            gt_pb_free(pb);                                                                         \
            return GT_ERR_HIP;                                                                      \
        }                                                                                           \
    } while (0)
#define PB_ALLOC(buf, bytes)                                                                        \
    do {                                                                                            \
        if ((buf).alloc(bytes)) { gt_set_error("pb build: out of device memory (%llu bytes)", (unsigned long long)(bytes)); gt_pb_free(pb); return GT_ERR_HIP; } \
    } while (0)
#define PB_MALLOC(ptr, bytes) PB_HIP(hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 1))

int gt_pb_build(gt_graph *g) {
    const uint32_t nnz = (uint32_t)g->info.nnz_local, ncols = g->ncols_total, nr = g->info.nnzrows;
    gt_pb *pb = new gt_pb();
    pb->nnz = nnz;
    pb->nbins = std::max<uint32_t>(1, (nr + R - 1) / R);
    g->pb = nullptr;
    if (nnz == 0) { g->pb = pb; return GT_OK; }
    hipStream_t s = 0;
    int binbits = 1;
    while ((1u << binbits) < pb->nbins) binbits++;
    const uint32_t nwin = (ncols + W - 1) / W;
    uint32_t ch = CH_DEFAULT;
    DevBuf nsub, cbase, tmp;
    PB_ALLOC(nsub, (uint64_t)(nwin + 1) * 4); PB_ALLOC(cbase, (uint64_t)(nwin + 1) * 4);
    uint32_t nchunks = 0;
    for (;;) {  // chunk ids must fit above the bin bits of a 32-bit sort key
        PB_HIP(hipMemsetAsync(nsub.p, 0, (uint64_t)(nwin + 1) * 4, s));
        k_win_counts<<<grid_for(nwin), TPB, 0, s>>>(g->JA, ncols, nwin, ch, nsub.as<uint32_t>());
        size_t tb = 0;
        PB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, nsub.as<uint32_t>(), cbase.as<uint32_t>(), nwin + 1, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceScan::ExclusiveSum(st.p, tb, nsub.as<uint32_t>(), cbase.as<uint32_t>(), nwin + 1, s));
        PB_HIP(hipMemcpyAsync(&nchunks, cbase.as<uint32_t>() + nwin, 4, hipMemcpyDeviceToHost, s));
        PB_HIP(hipStreamSynchronize(s));
        if ((uint64_t)nchunks < (1ull << (32 - binbits))) break;
        ch *= 2;
    }
    int chunkbits = 1;
    while ((1ull << chunkbits) < nchunks) chunkbits++;
    pb->nchunks = nchunks;
    PB_MALLOC(pb->cv0, (uint64_t)nchunks * 4); PB_MALLOC(pb->cv1, (uint64_t)nchunks * 4); PB_MALLOC(pb->ccol0, (uint64_t)nchunks * 4);
    k_fill_chunks<<<grid_for(nwin), TPB, 0, s>>>(g->JA, ncols, nwin, ch, cbase.as<uint32_t>(), pb->cv0, pb->cv1, pb->ccol0);

    // v-order: entries sorted by (chunk, bin); the radix sort is stable, so (col,row) order survives inside a segment
    DevBuf key, key2, idx, idx2;
    PB_ALLOC(key, (uint64_t)nnz * 4); PB_ALLOC(key2, (uint64_t)nnz * 4); PB_ALLOC(idx, (uint64_t)nnz * 4); PB_ALLOC(idx2, (uint64_t)nnz * 4);
    k_keys<<<nchunks, TPB, 0, s>>>(pb->cv0, pb->cv1, g->IA, binbits, key.as<uint32_t>(), idx.as<uint32_t>());
    hipcub::DoubleBuffer<uint32_t> dk(key.as<uint32_t>(), key2.as<uint32_t>()), di(idx.as<uint32_t>(), idx2.as<uint32_t>());
    {
        size_t tb = 0;
        PB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, di, nnz, 0, binbits + chunkbits, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceRadixSort::SortPairs(st.p, tb, dk, di, nnz, 0, binbits + chunkbits, s));
        PB_HIP(hipStreamSynchronize(s));
    }
    uint32_t *skey = dk.Current(), *sidx = di.Current(), *scratch_a = dk.Alternate(), *scratch_b = di.Alternate();
    // segments = runs of equal key
    uint32_t *head = scratch_a, *sid = scratch_b;
    k_heads<<<grid_for(nnz), TPB, 0, s>>>(skey, nnz, head);
    {
        size_t tb = 0;
        PB_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, head, sid, nnz, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceScan::InclusiveSum(st.p, tb, head, sid, nnz, s));
    }
    uint32_t nseg = 0;
    PB_HIP(hipMemcpyAsync(&nseg, sid + (nnz - 1), 4, hipMemcpyDeviceToHost, s));
    PB_HIP(hipStreamSynchronize(s));
    DevBuf vstart, segbin, segbin2, order, order2, lens, kscan;
    PB_ALLOC(vstart, (uint64_t)nseg * 4); PB_ALLOC(segbin, (uint64_t)nseg * 4); PB_ALLOC(segbin2, (uint64_t)nseg * 4);
    PB_ALLOC(order, (uint64_t)nseg * 4); PB_ALLOC(order2, (uint64_t)nseg * 4); PB_ALLOC(lens, (uint64_t)nseg * 4); PB_ALLOC(kscan, (uint64_t)nseg * 4);
    k_segments<<<grid_for(nnz), TPB, 0, s>>>(skey, sid, nnz, (1u << binbits) - 1, vstart.as<uint32_t>(), segbin.as<uint32_t>());
    // k-order: segments by (bin, chunk) -- stable sort of the (chunk, bin)-ordered segment list by bin
    {
        k_iota<<<grid_for(nseg), TPB, 0, s>>>(order2.as<uint32_t>(), nseg);
        size_t tb = 0;
        PB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, segbin.as<const uint32_t>(), segbin2.as<uint32_t>(),
                                                  order2.as<const uint32_t>(), order.as<uint32_t>(), nseg, 0, binbits, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceRadixSort::SortPairs(st.p, tb, segbin.as<const uint32_t>(), segbin2.as<uint32_t>(),
                                                  order2.as<const uint32_t>(), order.as<uint32_t>(), nseg, 0, binbits, s));
    }
    k_seg_len_sorted<<<grid_for(nseg), TPB, 0, s>>>(order.as<uint32_t>(), vstart.as<uint32_t>(), nseg, nnz, lens.as<uint32_t>());
    {
        size_t tb = 0;
        PB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, lens.as<uint32_t>(), kscan.as<uint32_t>(), nseg, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceScan::ExclusiveSum(st.p, tb, lens.as<uint32_t>(), kscan.as<uint32_t>(), nseg, s));
    }
    if (getenv("GRAPHTAP_PB_STATS")) {  // how many (chunk, row) pairs are distinct? (what pre-aggregation in phase 1 would leave)
        DevBuf k64, k64b; PB_ALLOC(k64, (uint64_t)nnz * 8); PB_ALLOC(k64b, (uint64_t)nnz * 8);
        k_chunk_row_keys<<<nchunks, TPB, 0, s>>>(pb->cv0, pb->cv1, g->IA, k64.as<uint64_t>());
        hipcub::DoubleBuffer<uint64_t> d64(k64.as<uint64_t>(), k64b.as<uint64_t>());
        size_t tb = 0;
        PB_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, d64, nnz, 0, 32 + chunkbits, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceRadixSort::SortKeys(st.p, tb, d64, nnz, 0, 32 + chunkbits, s));
        DevBuf cntb; PB_ALLOC(cntb, 8); PB_HIP(hipMemsetAsync(cntb.p, 0, 8, s));
        k_count_unique64<<<grid_for(nnz), TPB, 0, s>>>(d64.Current(), nnz, cntb.as<unsigned long long>());
        unsigned long long uq = 0; PB_HIP(hipMemcpy(&uq, cntb.p, 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "[pb] distinct (chunk,row) pairs: %llu of %u entries -> pre-aggregation factor %.3f\n", uq, nnz, (double)nnz / uq);
    }
    if (getenv("GRAPHTAP_PB_STATS")) {  // entry-weighted histogram of (chunk, bin) run lengths
        std::vector<uint32_t> hl(nseg);
        PB_HIP(hipMemcpy(hl.data(), lens.p, (uint64_t)nseg * 4, hipMemcpyDeviceToHost));
        uint64_t hist[33] = {0}, cnt[33] = {0};
        for (uint32_t l : hl) { int b = 0; while ((1u << (b + 1)) <= l) b++; hist[b] += l; cnt[b]++; }
        fprintf(stderr, "[pb] nnz=%u nbins=%u nchunks=%u nseg=%u mean run=%.1f\n", nnz, pb->nbins, nchunks, nseg, (double)nnz / nseg);
        for (int b = 0; b < 33; b++) if (cnt[b]) fprintf(stderr, "[pb] run length [%u,%u): %10llu runs, %5.2f%% of entries\n", 1u << b, 1u << (b + 1), (unsigned long long)cnt[b], 100.0 * hist[b] / nnz);
    }
    PB_MALLOC(pb->KSTART, (uint64_t)(nseg + 64) * 4);
    PB_HIP(hipMemsetAsync(pb->KSTART, 0, (uint64_t)(nseg + 64) * 4, s));
    k_kstart<<<grid_for(nseg), TPB, 0, s>>>(order.as<uint32_t>(), kscan.as<uint32_t>(), nseg, pb->KSTART);
    DevBuf binoff; PB_ALLOC(binoff, (uint64_t)(pb->nbins + 1) * 4);
    k_bin_offsets<<<grid_for(pb->nbins + 1), TPB, 0, s>>>(segbin2.as<uint32_t>(), kscan.as<uint32_t>(), nseg, nnz, pb->nbins, binoff.as<uint32_t>());

    const uint64_t ngroups = ((uint64_t)nnz + 63) / 64;
    PB_MALLOC(pb->LCOL, (uint64_t)nnz * 2); PB_MALLOC(pb->LROW, (uint64_t)nnz * 2);
    PB_MALLOC(pb->G, ngroups * sizeof(GroupRec));
    if (g->A) PB_MALLOC(pb->WT, (uint64_t)nnz * 4);
    k_static_streams<<<grid_for(nnz), TPB, 0, s>>>(skey, sidx, sid, nnz, binbits, pb->ccol0, g->IA, g->JI, g->A, vstart.as<uint32_t>(),
                                                   pb->KSTART, pb->LCOL, pb->LROW, pb->WT);
    k_group_table<<<grid_for(ngroups), TPB, 0, s>>>(skey, sid, nnz, nseg, vstart.as<uint32_t>(), pb->KSTART, (GroupRec *)pb->G);
    // phase-2 work list (host: nbins is small)
    std::vector<uint32_t> hoff(pb->nbins + 1);
    PB_HIP(hipMemcpyAsync(hoff.data(), binoff.p, (uint64_t)(pb->nbins + 1) * 4, hipMemcpyDeviceToHost, s));
    PB_HIP(hipStreamSynchronize(s));
    PB_HIP(hipGetLastError());
    std::vector<BinWork> work;
    for (uint32_t b = 0; b < pb->nbins; b++) {
        uint64_t n = hoff[b + 1] - hoff[b];
        if (!n) continue;
        uint32_t parts = (uint32_t)((n + EPW - 1) / EPW);
        for (uint32_t i = 0; i < parts; i++) {
            uint64_t a = hoff[b] + (uint64_t)i * EPW, e = std::min<uint64_t>(a + EPW, hoff[b + 1]);
            work.push_back(BinWork{b, (uint32_t)a, (uint32_t)e, parts == 1 ? 1u : 0u});
        }
    }
    // heaviest work first does not matter here (all parts but the last of a bin are EPW long)
    pb->nwork = (uint32_t)work.size();
    PB_MALLOC(pb->work, work.size() * sizeof(BinWork));
    PB_HIP(hipMemcpy(pb->work, work.data(), work.size() * sizeof(BinWork), hipMemcpyHostToDevice));
    g->pb = pb;
    return GT_OK;
}

template <class T, class TV, class TX, bool WEIGHTED, bool IS_MIN>
static int pb_run(const gt_graph *g, gt_pb *pb, const TX *x, T *y, hipStream_t s) {
    k_pb_scatter<T, TV, TX, WEIGHTED><<<pb->nchunks, P1_THREADS, 0, s>>>(pb->cv0, pb->cv1, pb->ccol0, g->ncols_total, pb->nnz, pb->LCOL, pb->WT,
                                                                     pb->KSTART, (const GroupRec *)pb->G, x, (TV *)pb->VAL);
    k_pb_gather<T, TV, IS_MIN><<<pb->nwork, P2_THREADS, 0, s>>>(pb->work, pb->LROW, (const TV *)pb->VAL, g->info.nnzrows, y);
    GT_HIP(hipGetLastError());
    return GT_OK;
}

int gt_pb_spmv(const gt_graph *g, int semiring, const void *x, void *y, hipStream_t s, bool f32_messages, bool x_is_f32) {
    gt_pb *pb = g->pb;
    GT_REQUIRE(pb, GT_ERR_STATE, "propagation-blocking structures were not built for this graph");
    if (pb->nnz == 0) return GT_OK;
    const uint32_t need = (semiring == GT_PLUS_F64 && !f32_messages) ? 8 : 4;
    if (pb->val_bytes < need) {  // one value stream per graph: SpMVs of one graph must not overlap in time
        if (pb->VAL) GT_HIP(hipFree(pb->VAL));
        pb->VAL = nullptr; pb->val_bytes = 0;
        GT_HIP(hipMalloc(&pb->VAL, (uint64_t)pb->nnz * need));
        pb->val_bytes = need;
    }
    switch (semiring) {
        case GT_PLUS_F64:
            GT_REQUIRE(!x_is_f32 || f32_messages, GT_ERR_STATE, "f32 message vector with an f64-message SpMV variant");
            if (f32_messages && x_is_f32) return pb_run<double, float, float, false, false>(g, pb, (const float *)x, (double *)y, s);
            if (f32_messages) return pb_run<double, float, double, false, false>(g, pb, (const double *)x, (double *)y, s);
            return pb_run<double, double, double, false, false>(g, pb, (const double *)x, (double *)y, s);
        case GT_PLUS_U32: return pb_run<uint32_t, uint32_t, uint32_t, false, false>(g, pb, (const uint32_t *)x, (uint32_t *)y, s);
        case GT_MIN_U32: return pb_run<uint32_t, uint32_t, uint32_t, false, true>(g, pb, (const uint32_t *)x, (uint32_t *)y, s);
        case GT_MINPLUS_U32:
            GT_REQUIRE(pb->WT, GT_ERR_INVALID, "min-plus SpMV needs a weighted graph");
            return pb_run<uint32_t, uint32_t, uint32_t, true, true>(g, pb, (const uint32_t *)x, (uint32_t *)y, s);
        default: gt_set_error("unknown semiring %d", semiring); return GT_ERR_INVALID;
    }
}
