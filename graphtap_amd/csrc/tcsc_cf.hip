// tcsc_cf.hip -- the tile in TCSC_CF form, built in HBM from the TCSC arrays, and the SpMV over its pair lists.
//
// Replaces TCSC_CF_BASE (src/ds/compressed_column.hpp:419-470; populate :603-1120) and the TCSC_CF branch of
// Vertex_Program::spmv_stationary (src/vp/vertex_program.hpp:1243-1317):
//   * IA (and A): inside every column the entries of SOURCE rows (rows whose vertex has no column) sit in the column's tail.
//     The reference gets there with a sequence of swaps (:671-708) whose result is: with the column's source entries in
//     ascending position s_0 < s_1 < ... and its regular entries in DESCENDING position g_0 > g_1 > ..., the entries s_p and
//     g_p trade places for every p with g_p > s_p (a prefix of the p's). That closed form is what k_cf_swap applies, one thread
//     per source entry -- the order of the regular rows the reference ends up with is reproduced exactly
//     (tests/golden/tcsc_cf.npz holds the reference's arrays).
//   * four pair lists [begin, end) into IA with the compressed column of each pair: regular rows of regular columns, regular
//     rows of sink columns, source rows of regular columns, source rows of sink columns -- including the reference's two
//     peculiarities of the last list (its length is the number of source ENTRIES of sink columns, :1040-1060, the unused pairs
//     stay zero; a pair starts at JA[j] + n rather than JA[j+1] - n, :1094), which no program can observe (a sink column's
//     message is 0 in PageRank, the only program run on TCSC_CF).
// Built on first use (gt_graph_tile_cf, or a PageRank under GT_TCSC_CF on the edge-parallel baseline variant); the
// propagation-blocking variants keep the same split inside their own streams (pb.hip, row classes).
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "gt_internal.h"

namespace {

constexpr int TPB = 256;
constexpr uint32_t ITEM = 4096;   // entries per work item of the pair-list SpMV (a hub column's pair is millions of entries long)

inline unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + TPB - 1) / TPB, 1u << 20); }

struct IsSource {
    const uint32_t *R2C;
    __host__ __device__ uint32_t operator()(uint32_t row) const { return R2C[row] == 0xFFFFFFFFu ? 1u : 0u; }
};

// S[i] = source entries before entry i (S is an inclusive scan shifted by one: S[0] = 0, S[nnz] = all of them)
__global__ void k_cf_regular_positions(const uint32_t *__restrict__ JI, const uint32_t *__restrict__ JA, const uint32_t *__restrict__ S,
                                       const uint32_t *__restrict__ IA, const uint32_t *__restrict__ R2C, uint64_t nnz,
                                       uint32_t *__restrict__ G) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x) {
        if (R2C[IA[i]] == 0xFFFFFFFFu) continue;
        const uint32_t j = JI[i], b = JA[j], e = JA[j + 1];
        const uint32_t n = S[e] - S[b], nreg = (e - b) - n;
        const uint32_t before = ((uint32_t)i - b) - (S[i] - S[b]);   // regular entries of the column ahead of this one
        const uint32_t t = nreg - 1 - before;                        // its rank from the column's end
        if (t < n) G[b + t] = (uint32_t)i;                           // only the first n can have a partner
    }
}

__global__ void k_cf_swap(const uint32_t *__restrict__ JI, const uint32_t *__restrict__ JA, const uint32_t *__restrict__ S,
                          const uint32_t *__restrict__ IA, const uint32_t *__restrict__ A, const uint32_t *__restrict__ R2C,
                          const uint32_t *__restrict__ G, uint64_t nnz, uint32_t *__restrict__ IAcf, uint32_t *__restrict__ Acf) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nnz; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = IA[i];
        if (R2C[r] != 0xFFFFFFFFu) continue;
        const uint32_t j = JI[i], b = JA[j], e = JA[j + 1];
        const uint32_t n = S[e] - S[b], nreg = (e - b) - n;
        const uint32_t p = S[i] - S[b];
        if (p >= nreg) continue;
        const uint32_t g = G[b + p];
        if (g <= (uint32_t)i) continue;
        IAcf[i] = IA[g]; IAcf[g] = r;
        if (A) { Acf[i] = A[g]; Acf[g] = A[i]; }
    }
}

// per compressed column: the pair of its regular rows, and what it contributes to the four lists
__global__ void k_cf_columns(const uint32_t *__restrict__ JA, const uint32_t *__restrict__ S, const uint32_t *__restrict__ JC,
                             const uint8_t *__restrict__ IJ, uint32_t nc, uint32_t *__restrict__ nnz_pairs,
                             uint32_t *__restrict__ f0, uint32_t *__restrict__ f1, uint32_t *__restrict__ f2, uint32_t *__restrict__ f3,
                             uint32_t *__restrict__ c3) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        const uint32_t b = JA[j], e = JA[j + 1], m = e - b, n = S[e] - S[b];
        nnz_pairs[2 * (uint64_t)j] = b; nnz_pairs[2 * (uint64_t)j + 1] = e - n;
        const bool regular_col = (IJ[JC[j]] & 1u) != 0, has = m != 0;
        f0[j] = has && regular_col && m != n;
        f1[j] = has && !regular_col && m != n;
        f2[j] = has && regular_col && n != 0;
        f3[j] = has && !regular_col && n != 0;
        c3[j] = (has && !regular_col) ? n : 0u;
    }
}

__global__ void k_cf_fill_lists(const uint32_t *__restrict__ JA, const uint32_t *__restrict__ S, uint32_t nc,
                                const uint32_t *__restrict__ f0, const uint32_t *__restrict__ f1, const uint32_t *__restrict__ f2,
                                const uint32_t *__restrict__ f3, const uint32_t *__restrict__ o0, const uint32_t *__restrict__ o1,
                                const uint32_t *__restrict__ o2, const uint32_t *__restrict__ o3,
                                uint32_t *__restrict__ ja0, uint32_t *__restrict__ jc0, uint32_t *__restrict__ ja1, uint32_t *__restrict__ jc1,
                                uint32_t *__restrict__ ja2, uint32_t *__restrict__ jc2, uint32_t *__restrict__ ja3, uint32_t *__restrict__ jc3) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nc; j += gridDim.x * blockDim.x) {
        const uint32_t b = JA[j], e = JA[j + 1], n = S[e] - S[b];
        if (f0[j]) { const uint32_t o = o0[j]; ja0[2 * (uint64_t)o] = b; ja0[2 * (uint64_t)o + 1] = e - n; jc0[o] = j; }
        if (f1[j]) { const uint32_t o = o1[j]; ja1[2 * (uint64_t)o] = b; ja1[2 * (uint64_t)o + 1] = e - n; jc1[o] = j; }
        if (f2[j]) { const uint32_t o = o2[j]; ja2[2 * (uint64_t)o] = e - n; ja2[2 * (uint64_t)o + 1] = e; jc2[o] = j; }
        if (f3[j]) { const uint32_t o = o3[j]; ja3[2 * (uint64_t)o] = b + n; ja3[2 * (uint64_t)o + 1] = e; jc3[o] = j; }   // :1094
    }
}

// work items of the SpMV: pair q is cut into ceil(len / ITEM) items
__global__ void k_cf_item_counts(const uint32_t *__restrict__ ja, uint32_t np, uint32_t *__restrict__ cnt) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < np; q += gridDim.x * blockDim.x)
        cnt[q] = (ja[2 * (uint64_t)q + 1] - ja[2 * (uint64_t)q] + ITEM - 1) / ITEM;
}
__global__ void k_cf_items(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ off, uint32_t np, uint2 *__restrict__ items) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < np; q += gridDim.x * blockDim.x)
        for (uint32_t k = 0, o = off[q]; k < cnt[q]; k++) items[o + k] = make_uint2(q, k);
}

// y[IA[i]] += x[column of the pair] over the pairs of one list: one wave per item
__global__ void __launch_bounds__(TPB) k_cf_spmv(const uint2 *__restrict__ items, uint32_t nitems, const uint32_t *__restrict__ ja,
                                                 const uint32_t *__restrict__ jc, const uint32_t *__restrict__ IAcf,
                                                 const uint32_t *__restrict__ xslot, const double *__restrict__ x, double *__restrict__ y) {
    const uint32_t lane = threadIdx.x & 63u, wpb = TPB / 64;
    for (uint32_t it = blockIdx.x * wpb + (threadIdx.x >> 6); it < nitems; it += gridDim.x * wpb) {
        const uint2 w = items[it];
        const uint32_t lo = ja[2 * (uint64_t)w.x] + w.y * ITEM, hi = min(ja[2 * (uint64_t)w.x + 1], lo + ITEM);
        const uint32_t l = jc[w.x];
        const double xl = x[xslot ? xslot[l] : l];
        for (uint32_t i = lo + lane; i < hi; i += 64) unsafeAtomicAdd(&y[IAcf[i]], xl);
    }
}

}  // namespace

struct gt_tcsc_cf {
    uint32_t *IA = nullptr, *A = nullptr, *nnz_pairs = nullptr;
    uint32_t NC[4] = {0, 0, 0, 0};
    uint32_t *JA[4] = {nullptr, nullptr, nullptr, nullptr}, *JC[4] = {nullptr, nullptr, nullptr, nullptr};
    uint2 *items[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t nitems[4] = {0, 0, 0, 0};
};

void gt_tcsc_cf_free(gt_tcsc_cf *c) {
    if (!c) return;
    void *ptrs[] = {c->IA, c->A, c->nnz_pairs};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (int k = 0; k < 4; k++) { if (c->JA[k]) (void)hipFree(c->JA[k]); if (c->JC[k]) (void)hipFree(c->JC[k]); if (c->items[k]) (void)hipFree(c->items[k]); }
    delete c;
}

namespace {
struct Scratch {   // freed on every exit path
    std::vector<void *> p;
    ~Scratch() { for (void *q : p) (void)hipFree(q); }
    template <class T> int get(T **out, uint64_t elems) {
        void *q = nullptr;
        if (hipMalloc(&q, std::max<uint64_t>(elems, 1) * sizeof(T)) != hipSuccess) { gt_set_error("TCSC_CF build: out of device memory"); return GT_ERR_HIP; }
        p.push_back(q); *out = (T *)q; return GT_OK;
    }
};
template <class In>
int exclusive_sum(Scratch &sc, In in, uint32_t *out, uint64_t n, hipStream_t s) {
    size_t tb = 0;
    GT_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, n, s));
    char *tmp; int st = sc.get(&tmp, tb); if (st != GT_OK) return st;
    GT_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, in, out, n, s));
    return GT_OK;
}
}  // namespace

int gt_tcsc_cf_build(gt_graph *g) {
    if (g->cf) return GT_OK;
    GT_REQUIRE(!gt_has_exchange(g), GT_ERR_UNSUPPORTED, "the TCSC_CF pair lists exist for a single tile (one rank): the propagation-blocking "
               "variants split source rows inside their own streams on any number of ranks");
    const uint64_t nnz = g->info.nnz_local;
    const uint32_t nc = g->info.nnzcols;
    hipStream_t s = nullptr;
    gt_tcsc_cf *c = new gt_tcsc_cf();
    struct Guard { gt_tcsc_cf *c; ~Guard() { gt_tcsc_cf_free(c); } } guard{c};
    auto keep = [&](uint32_t **out, uint64_t elems) -> int {
        if (hipMalloc((void **)out, std::max<uint64_t>(elems, 1) * 4) != hipSuccess) { gt_set_error("TCSC_CF build: out of device memory"); return GT_ERR_HIP; }
        return GT_OK;
    };
    int st;
    if ((st = keep(&c->IA, nnz)) != GT_OK || (g->A && (st = keep(&c->A, nnz)) != GT_OK) || (st = keep(&c->nnz_pairs, 2 * (uint64_t)nc)) != GT_OK) return st;
    GT_HIP(hipMemcpyAsync(c->IA, g->IA, nnz * 4, hipMemcpyDeviceToDevice, s));
    if (g->A) GT_HIP(hipMemcpyAsync(c->A, g->A, nnz * 4, hipMemcpyDeviceToDevice, s));
    Scratch sc;
    uint32_t *S, *G;
    if ((st = sc.get(&S, nnz + 1)) != GT_OK || (st = sc.get(&G, nnz)) != GT_OK) return st;
    // entries of source rows ahead of every entry
    hipcub::TransformInputIterator<uint32_t, IsSource, const uint32_t *> flags(g->IA, IsSource{g->R2C});
    GT_HIP(hipMemsetAsync(S, 0, 4, s));
    if (nnz) {
        size_t tb = 0;
        GT_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, flags, S + 1, nnz, s));
        char *tmp; if ((st = sc.get(&tmp, tb)) != GT_OK) return st;
        GT_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tb, flags, S + 1, nnz, s));
        k_cf_regular_positions<<<grid_for(nnz), TPB, 0, s>>>(g->JI, g->JA, S, g->IA, g->R2C, nnz, G);
        k_cf_swap<<<grid_for(nnz), TPB, 0, s>>>(g->JI, g->JA, S, g->IA, g->A, g->R2C, G, nnz, c->IA, c->A);
    }
    // the four lists
    uint32_t *f[4], *o[4], *c3, *o3c;
    for (int k = 0; k < 4; k++) if ((st = sc.get(&f[k], (uint64_t)nc + 1)) != GT_OK || (st = sc.get(&o[k], (uint64_t)nc + 1)) != GT_OK) return st;
    if ((st = sc.get(&c3, (uint64_t)nc + 1)) != GT_OK || (st = sc.get(&o3c, (uint64_t)nc + 1)) != GT_OK) return st;
    for (int k = 0; k < 4; k++) GT_HIP(hipMemsetAsync(f[k] + nc, 0, 4, s));
    GT_HIP(hipMemsetAsync(c3 + nc, 0, 4, s));
    if (nc) k_cf_columns<<<grid_for(nc), TPB, 0, s>>>(g->JA, S, g->JC, g->IJ, nc, c->nnz_pairs, f[0], f[1], f[2], f[3], c3);
    for (int k = 0; k < 4; k++) if ((st = exclusive_sum(sc, f[k], o[k], (uint64_t)nc + 1, s)) != GT_OK) return st;   // o[k][nc] = pairs of list k
    if ((st = exclusive_sum(sc, c3, o3c, (uint64_t)nc + 1, s)) != GT_OK) return st;
    uint32_t filled[4], src_entries_sink = 0;
    for (int k = 0; k < 4; k++) GT_HIP(hipMemcpyAsync(&filled[k], o[k] + nc, 4, hipMemcpyDeviceToHost, s));
    GT_HIP(hipMemcpyAsync(&src_entries_sink, o3c + nc, 4, hipMemcpyDeviceToHost, s));
    GT_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < 4; k++) {
        c->NC[k] = (k == 3) ? src_entries_sink : filled[k];   // list 3 is as long as its columns have source ENTRIES (:1040-1062)
        if ((st = keep(&c->JA[k], 2 * (uint64_t)c->NC[k])) != GT_OK || (st = keep(&c->JC[k], c->NC[k])) != GT_OK) return st;
        GT_HIP(hipMemsetAsync(c->JA[k], 0, std::max<uint64_t>(2 * (uint64_t)c->NC[k], 1) * 4, s));
        GT_HIP(hipMemsetAsync(c->JC[k], 0, std::max<uint64_t>(c->NC[k], 1) * 4, s));
    }
    if (nc) k_cf_fill_lists<<<grid_for(nc), TPB, 0, s>>>(g->JA, S, nc, f[0], f[1], f[2], f[3], o[0], o[1], o[2], o[3],
                                                        c->JA[0], c->JC[0], c->JA[1], c->JC[1], c->JA[2], c->JC[2], c->JA[3], c->JC[3]);
    // work items of the SpMV over each list (zero-filled pairs of list 3 are empty ranges: no items)
    for (int k = 0; k < 4; k++) {
        const uint32_t np = c->NC[k];
        if (!np) continue;
        uint32_t *cnt, *off;
        if ((st = sc.get(&cnt, (uint64_t)np + 1)) != GT_OK || (st = sc.get(&off, (uint64_t)np + 1)) != GT_OK) return st;
        GT_HIP(hipMemsetAsync(cnt + np, 0, 4, s));
        k_cf_item_counts<<<grid_for(np), TPB, 0, s>>>(c->JA[k], np, cnt);
        if ((st = exclusive_sum(sc, cnt, off, (uint64_t)np + 1, s)) != GT_OK) return st;
        GT_HIP(hipMemcpyAsync(&c->nitems[k], off + np, 4, hipMemcpyDeviceToHost, s));
        GT_HIP(hipStreamSynchronize(s));
        if (!c->nitems[k]) continue;
        if (hipMalloc((void **)&c->items[k], (uint64_t)c->nitems[k] * sizeof(uint2)) != hipSuccess) { gt_set_error("TCSC_CF build: out of device memory"); return GT_ERR_HIP; }
        k_cf_items<<<grid_for(np), TPB, 0, s>>>(cnt, off, np, c->items[k]);
    }
    GT_HIP(hipStreamSynchronize(s));
    GT_HIP(hipGetLastError());
    guard.c = nullptr;
    g->cf = c;
    return GT_OK;
}

int gt_tcsc_cf_arrays(gt_graph *g, gt_tile_cf_arrays *a) {
    int st = gt_tcsc_cf_build(g);
    if (st != GT_OK) return st;
    const gt_tcsc_cf *c = g->cf;
    a->IA = c->IA; a->A = c->A; a->JA_REG_R_NNZ_C = c->nnz_pairs;
    for (int k = 0; k < 4; k++) { a->NC[k] = c->NC[k]; a->JA[k] = c->JA[k]; a->JC[k] = c->JC[k]; }
    return GT_OK;
}

// x in slot order (g->xslot), y per compressed row; accumulates into y
int gt_tcsc_cf_spmv(gt_graph *g, const double *x, double *y, bool first, bool running, bool last, hipStream_t s) {
    int st = gt_tcsc_cf_build(g);
    if (st != GT_OK) return st;
    const gt_tcsc_cf *c = g->cf;
    // vp:1246-1262 (iteration 0: regular rows of sink columns), :1264-1281 (regular rows of regular columns),
    // :1282-1313 (last iteration: source rows; the sink-column list only when the regular-column one is not empty, :1298)
    const bool use[4] = {running, first, last, last && c->NC[GT_CF_SRC_R_REG_C] != 0};
    const int order[4] = {GT_CF_REG_R_SNK_C, GT_CF_REG_R_REG_C, GT_CF_SRC_R_REG_C, GT_CF_SRC_R_SNK_C};
    for (int o = 0; o < 4; o++) {
        const int k = order[o];
        if (!use[k] || !c->nitems[k]) continue;
        const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)c->nitems[k] + TPB / 64 - 1) / (TPB / 64), 256u * 32u);
        k_cf_spmv<<<blocks, TPB, 0, s>>>(c->items[k], c->nitems[k], c->JA[k], c->JC[k], c->IA, g->xslot, x, y);
    }
    GT_HIP(hipGetLastError());
    return GT_OK;
}
