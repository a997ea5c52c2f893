// dist.hip -- the iteration loop over SEVERAL ranks (one process per GPU), host code in C++ calling RCCL directly.
//
// Replaces the communication the reference runs inside Vertex_Program::execute (src/vp/vertex_program.hpp:408-441) from its
// C++ mains (src/apps/pr.cpp:15-60):
//   MPI_Ibcast of every x segment down its column group, vp:843-862 / 970-1013  ->  K grouped ncclSend/ncclRecv rounds of the
//       NEEDED columns only (gt_graph_exchange_plan), on a communication stream, slice k+1 in flight while the SpMV's
//       phase 1 of slice k runs (gt_program_combine_slice, helper streams of engine.hip);
//   Isend/Irecv of partial y to the row-group leader, vp:1083-1111              ->  gone: tile-rows make y complete locally;
//                                                                                  Degree in _COL_ order: one ncclAllReduce;
//   MPI_Allreduce of the convergence count, vp:1918, and of the checksums, vp:1940,1956 -> ncclAllReduce of 8-byte words.
// graphtap_amd/dist.py is the same loop over torch.distributed (tests, bench.py); both drive the phase-level C ABI.
//
// Transports: RCCL (the product; librccl is resolved with dlopen so that a process that already carries a copy -- PyTorch
// ships one -- keeps using that one, and a box without it still loads this library), and LOOPBACK: the p ranks of one process
// on one GPU, one host thread each, exchanging with device copies -- the p-rank rehearsal a one-GPU box allows (RCCL refuses
// two ranks on one device), used by tests/.
#include <dlfcn.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <algorithm>
#include <mutex>
#include <vector>

#include "gt_internal.h"

namespace {

// ---- RCCL entry points, resolved at first use
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;   // optional (diagnostics)
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                // optional: tears the communicator down after a deadline passed
};
Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);   // the copy this process already uses, if any
            if (r.lib) break;
        }
        for (const char *name : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!r.lib) return;
#define GT_SYM(field, sym) r.field = (decltype(r.field))dlsym(r.lib, sym)
        GT_SYM(GetUniqueId, "ncclGetUniqueId"); GT_SYM(CommInitRank, "ncclCommInitRank"); GT_SYM(CommDestroy, "ncclCommDestroy");
        GT_SYM(GroupStart, "ncclGroupStart"); GT_SYM(GroupEnd, "ncclGroupEnd"); GT_SYM(Send, "ncclSend"); GT_SYM(Recv, "ncclRecv");
        GT_SYM(AllReduce, "ncclAllReduce"); GT_SYM(GetErrorString, "ncclGetErrorString"); GT_SYM(CommCount, "ncclCommCount"); GT_SYM(CommAbort, "ncclCommAbort");
#undef GT_SYM
        if (!(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllReduce && r.GetErrorString)) r.lib = nullptr;
    });
    return r.lib ? &r : nullptr;
}
#define GT_LOOP_BARRIER(c)                                                                                      \
    do { if (!(c).barrier()) { gt_set_error("a peer rank of this loopback group failed"); return GT_ERR_STATE; } } while (0)
#define GT_NCCL(call)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) { gt_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, rccl()->GetErrorString(r_)); return GT_ERR_HIP; } \
    } while (0)

__global__ void k_dist_preload() {}
__global__ void k_add_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// ---- sparse frontier exchange (min programs): blocks of the send buffer with few active (!= infinity()) messages travel as
// (index in block, value) pairs -- the reference's sparse broadcast, vp:970-1013, with its 0 = dense / k = k-1 entries count
// header (vp:766-773) replaced by a count exchange
struct BlockTab { uint32_t start, len; };   // a (slice, peer) block of the send buffer / of x
__device__ __forceinline__ uint32_t block_of(const BlockTab *__restrict__ tab, uint32_t nb, uint32_t i) {
    uint32_t lo = 0, hi = nb;   // last block whose start <= i (blocks are in ascending order, empty ones included)
    while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (tab[mid].start <= i) lo = mid; else hi = mid; }
    return lo;
}
__global__ void k_block_active(const uint32_t *__restrict__ send, uint32_t n, const BlockTab *__restrict__ tab, uint32_t nb, uint32_t *__restrict__ counts) {
    const uint32_t n64 = (n + 63) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += gridDim.x * blockDim.x) {
        const bool act = i < n && send[i] != GT_INF;
        const uint32_t b = i < n ? block_of(tab, nb, i) : 0xFFFFFFFFu;
        const uint32_t b0 = __builtin_amdgcn_readfirstlane(b);
        if (__all(b == b0 || i >= n)) {   // a wave rarely straddles blocks
            const uint64_t m = __ballot(act);
            if ((threadIdx.x & 63) == 0 && m && b0 != 0xFFFFFFFFu) atomicAdd(&counts[b0], (uint32_t)__popcll((unsigned long long)m));
        } else if (act) atomicAdd(&counts[b], 1u);
    }
}
// A slot of block b's pair list for every lane that wants one: ONE atomic per distinct block inside the wave (there are only K*P
// cursors: an atomic per pair was 2 ms for a 0.5 M-vertex list and 7.9 ms for a compaction of millions). All lanes of the wave
// must call; `want` = this lane takes a slot.
__device__ __forceinline__ uint32_t block_slot(uint32_t *__restrict__ cursor, uint32_t b, bool want) {
    uint32_t slot = 0;
    uint64_t todo = __ballot(want);
    const uint32_t lane = threadIdx.x & 63u;
    while (todo) {
        const uint32_t leader = (uint32_t)__ffsll((unsigned long long)todo) - 1u;
        const uint32_t b0 = __shfl(b, (int)leader);
        const uint64_t same = __ballot(want && b == b0) & todo;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&cursor[b0], (uint32_t)__popcll((unsigned long long)same));
        base = __shfl(base, (int)leader);
        if (want && b == b0) slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0));
        todo &= ~same;
    }
    return slot;
}
// pairs of block b go to pairs[tab[b].start ...] (a sparse block holds at most len / 2 of them)
__global__ void k_block_compact(const uint32_t *__restrict__ send, uint32_t n, const BlockTab *__restrict__ tab, uint32_t nb, const uint8_t *__restrict__ sparse,
                                uint32_t *__restrict__ cursor, uint2 *__restrict__ pairs) {
    const uint32_t n64 = (n + 63) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += gridDim.x * blockDim.x) {
        const uint32_t v = i < n ? send[i] : GT_INF;
        uint32_t b = 0; bool want = false;
        if (v != GT_INF) { b = block_of(tab, nb, i); want = sparse[b] != 0; }
        const uint32_t slot = block_slot(cursor, b, want);
        if (want) pairs[tab[b].start + slot] = uint2{i - tab[b].start, v};
    }
}
__global__ void k_scatter_pairs(const uint2 *__restrict__ pairs, uint32_t n, uint32_t *__restrict__ xblock) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const uint2 pv = pairs[i]; xblock[pv.x] = pv.y; }
}

// one launch for all the blocks of x that arrive as pairs: they start from infinity() (blockIdx.y = block)
__global__ void k_fill_blocks(uint32_t *__restrict__ x, const BlockTab *__restrict__ rtab, const uint8_t *__restrict__ sel, uint32_t v) {
    const uint32_t b = blockIdx.y;
    if (!sel[b]) return;
    const BlockTab t = rtab[b];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < t.len; i += gridDim.x * blockDim.x) x[t.start + i] = v;
}
// ---- frontier lists on several ranks (the reference's (xi, xv) pairs at any np, vp:711-784, 970-1013, 1475-1489)
// send positions of every owned column: inverse of send_idx as a CSR (built once per graph)
__global__ void k_send_keys(const uint32_t *__restrict__ send_idx, uint32_t n, uint32_t *__restrict__ key, uint32_t *__restrict__ pos) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { key[i] = send_idx[i]; pos[i] = i; }
}
__global__ void k_send_ptr(const uint32_t *__restrict__ key_sorted, uint32_t n, uint32_t ncols, uint32_t *__restrict__ ptr) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c <= ncols; c += gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = n;   // first i with key_sorted[i] >= c
        while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (key_sorted[mid] < c) lo = mid + 1; else hi = mid; }
        ptr[c] = lo;
    }
}
// the vertices the last apply changed -> (index in block, message) pairs in every block of the send buffer that carries their
// column. A list longer than `cap` (or none: n_dev's top bit) raises *flag instead: the rank then sends dense blocks.
__global__ void k_pairs_from_list(const uint32_t *__restrict__ list, const unsigned int *__restrict__ n_dev, uint32_t cap, const uint8_t *__restrict__ IJ,
                                  const uint32_t *__restrict__ JV, const uint32_t *__restrict__ send_ptr, const uint32_t *__restrict__ send_pos,
                                  const uint32_t *__restrict__ s0, uint32_t vid_base, gt_vidmap vm, int kind, const BlockTab *__restrict__ stab, uint32_t nb,
                                  uint32_t *__restrict__ cursor, uint2 *__restrict__ pairs, uint32_t *__restrict__ flag) {
    const uint32_t n = *n_dev;
    if (n > cap) { if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 1u; return; }
    const uint32_t n64 = (n + 63) & ~63u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += gridDim.x * blockDim.x) {
        uint32_t j = 0, j1 = 0, m = 0;
        if (i < n) {
            const uint32_t v = list[i];
            if (IJ[v] & 2u) {   // (no column: the vertex sends nothing)
                const uint32_t c = JV[v];
                m = (kind == GT_BFS) ? gt_vid_of(vm, (uint64_t)vid_base + v) : s0[v];   // bfs.h:52-54, sssp.h:44-46, cc.h:38-40
                j = send_ptr[c]; j1 = send_ptr[c + 1];
            }
        }
        while (__any(j < j1)) {   // wave-uniform rounds: the t-th destination of every lane's column
            const bool want = j < j1;
            uint32_t pos = 0, b = 0;
            if (want) { pos = send_pos[j]; b = block_of(stab, nb, pos); }
            const uint32_t slot = block_slot(cursor, b, want);
            if (want) pairs[stab[b].start + slot] = uint2{pos - stab[b].start, m};
            j++;
        }
    }
}
// the words of the per-iteration all-reduce: [0] active vertices, [1] ranks that need a second (dense) round, then for every
// (source rank, block) the number of active messages, bit 32 set when the block travels as pairs
__global__ void k_words(unsigned long long *__restrict__ w, uint32_t nwords, const unsigned long long *__restrict__ d_active, const uint32_t *__restrict__ flag,
                        const uint32_t *__restrict__ counts, const uint8_t *__restrict__ pair_form, uint32_t me, uint32_t nb) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += gridDim.x * blockDim.x) {
        unsigned long long v = 0;
        if (i == 0) v = d_active ? *d_active : 0ull;
        else if (i == 1) v = (flag && *flag) ? 1ull : 0ull;
        else if (i >= 2 + me * nb && i < 2 + (me + 1) * nb) {
            const uint32_t b = i - 2 - me * nb;
            const bool pf = pair_form ? pair_form[b] != 0 : !(flag && *flag);   // list mode: every block as pairs
            v = (unsigned long long)counts[b] | (pf ? (1ull << 32) : 0ull);
        }
        w[i] = v;
    }
}
// the received pairs of all blocks -> one frontier: local column, message, entry count (ftab[b] = {first pair's output slot, pairs})
struct FrontTab { uint32_t out0, n; };
__global__ void __launch_bounds__(256) k_pairs_frontier(const uint2 *__restrict__ pairs, const BlockTab *__restrict__ rtab, const FrontTab *__restrict__ ftab, uint32_t nb, uint32_t total,
                                 const uint32_t *__restrict__ JA, uint32_t *__restrict__ col, uint32_t *__restrict__ val, uint32_t *__restrict__ deg,
                                 unsigned long long *__restrict__ entries) {
    unsigned long long e = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nb;   // last block whose out0 <= i among those that hold pairs (empty ones share their successor's out0)
        while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (ftab[mid].out0 <= i) lo = mid; else hi = mid; }
        while (ftab[lo].n == 0 || i - ftab[lo].out0 >= ftab[lo].n) lo--;   // step back over empty blocks with the same out0
        const uint2 pv = pairs[rtab[lo].start + (i - ftab[lo].out0)];
        const uint32_t c = rtab[lo].start + pv.x, dg = JA[c + 1] - JA[c];
        col[i] = c; val[i] = pv.y; deg[i] = dg; e += dg;
    }
    if (entries) {   // one atomic per workgroup
        __shared__ unsigned long long part[4];
        for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
        __syncthreads();
        if (threadIdx.x == 0) { e = part[0] + part[1] + part[2] + part[3]; if (e) atomicAdd(entries, e); }
    }
}

// ---- collectives of the distributed build (Matrix::distribute, mat/matrix.hpp:693-810): both transports
__global__ void k_max_u8(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) if (src[i] > dst[i]) dst[i] = src[i];
}
// ---- loopback: what the p ranks of one process share
struct LoopPeer { const char *send = nullptr; const uint32_t *y = nullptr; const gt_graph *g = nullptr; uint32_t x_bytes = 0; uint64_t word = 0;
                  const uint2 *pairs = nullptr; const unsigned long long *words = nullptr; const uint32_t *counts = nullptr; const void *coll = nullptr; /* host [K*P] active messages per send block, or null = all dense */ };
struct LoopCtx {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<LoopPeer> peer;
    bool aborted = false;   // a rank left gt_dist_execute with an error: nobody waits for it any more
    bool barrier() {   // false: a peer failed -- or never came (deadline, GRAPHTAP_TIMEOUT_S) -- and the caller gives up too
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const uint64_t gen = generation;
        if (++arrived == n) { arrived = 0; generation++; cv.notify_all(); }
        else if (!cv.wait_for(lk, std::chrono::duration<double>(gt_wait_limit_s()), [&] { return generation != gen || aborted; })) {
            aborted = true; arrived = 0; generation++; cv.notify_all();   // ranks that disagree on the shape of the loop must not wait for each other for ever
        }
        return !aborted;
    }
    void abort() { std::unique_lock<std::mutex> lk(mu); aborted = true; arrived = 0; generation++; cv.notify_all(); }
};

}  // namespace

struct gt_dist {
    int rank = 0, nranks = 1;
    // RCCL
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    bool dead = false;              // a deadline passed: the communicator was aborted, every later collective fails at once
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr;
    std::vector<hipEvent_t> ev_slice, ev_pack;
    uint64_t *d_word = nullptr;     // device staging of the small all-reduces
    // loopback
    std::shared_ptr<LoopCtx> loop;
    uint32_t *tmp = nullptr; uint64_t tmp_elems = 0;
    // sparse frontier exchange: block tables of the current graph, counts, pair buffers
    const gt_graph *sp_graph = nullptr;
    BlockTab *d_stab = nullptr;                      // [K*P] send blocks
    uint32_t *d_cnt = nullptr, *d_cur = nullptr;     // [K*P] active messages / compaction cursors per send block
    uint8_t *d_smode = nullptr;                      // [K*P] 1 = the send block travels as pairs
    uint32_t *d_cx = nullptr;                        // [2][P*K] count exchange staging (RCCL)
    uint2 *pairs_send = nullptr, *pairs_recv = nullptr;
    uint64_t pairs_send_cap = 0, pairs_recv_cap = 0;
    std::vector<BlockTab> stab, rtab;                // host copies: send blocks, blocks of x
    std::vector<uint32_t> cnt_send, cnt_recv;        // [K*P] this iteration: what I send / what each peer sends me
    bool sparse_now = false;                         // this iteration's exchange uses the counts above
    bool lists_protocol = false;                     // the converge-mode loop of the min programs (counts ride the convergence all-reduce)
    std::vector<uint8_t> form_send, form_recv;       // [K*P] 1 = the block travels as (index, value) pairs this iteration
    // frontier lists on several ranks: send positions of every owned column (CSR), tables and words of the per-iteration all-reduce
    uint32_t *send_ptr = nullptr, *send_pos = nullptr;
    BlockTab *d_rtab = nullptr;                      // [K*P] blocks of x
    FrontTab *d_ftab = nullptr;                      // [K*P] where the pairs of each block go in the frontier
    std::vector<FrontTab> h_ftab;
    uint8_t *d_rsel = nullptr;                       // [K*P] blocks of x to pre-fill with infinity()
    uint32_t *d_flag = nullptr;                      // 1 = this rank's list did not fit: it needs the dense round
    unsigned long long *d_words = nullptr; uint32_t nwords = 0;   // [2 + P*K*P]
    std::vector<unsigned long long> h_words;
    uint32_t list_send_cap = 0;                      // longest list that is sure to leave every block below half full
    uint64_t list_iters = 0, pair_spmspv_iters = 0, round_trips = 0;   // diagnostics of the last execute
    // per-iteration timing of the last execute (HIP events on the program's stream / the communication stream)
    std::vector<hipEvent_t> tev;                     // per iteration: start, messages packed, SpMV done, apply done, first slice in, last slice in
    size_t tev_used = 0;
    int rccl_ranks = 0;
    uint64_t plan_checked = 0;                       // serial of the graph whose exchange plan was verified against the peers'
    bool timing = false; uint32_t t_iter = 0;        // events are recorded for the first GT_DIST_TIMED_ITERS iterations of a timed execute
    std::vector<uint8_t> t_mode;                     // per timed iteration: 0 dense blocks, 1 pairs scattered into x, 2 SpMSpV from the pairs
    std::vector<double> iter_ms;                     // [iterations][GT_DIST_TIME_FIELDS], filled when the execute ends
    uint64_t bytes_sent = 0, bytes_dense = 0, exchanges = 0;   // since the last gt_dist_exchange_stats reset (sent: to other ranks and to itself)
};

namespace {

// ---- per-iteration timing (HIP events; gt_dist_iteration_times)
constexpr uint32_t GT_DIST_TIMED_ITERS = 64, GT_DIST_TIME_FIELDS = 7, GT_DIST_TICKS = 6;
enum { T_START = 0, T_SEND_READY = 1, T_SPMV_DONE = 2, T_APPLY_DONE = 3, T_FIRST_SLICE = 4, T_LAST_SLICE = 5 };
int tick_at(gt_dist *d, uint32_t it, int which, hipStream_t s) {
    if (!d->timing || it >= GT_DIST_TIMED_ITERS) return GT_OK;
    const size_t i = (size_t)it * GT_DIST_TICKS + which;
    while (d->tev.size() <= i) { hipEvent_t e; GT_HIP(hipEventCreate(&e)); d->tev.push_back(e); }
    GT_HIP(hipEventRecord(d->tev[i], s));
    return GT_OK;
}
int tick(gt_dist *d, int which, hipStream_t s) { return tick_at(d, d->t_iter, which, s); }
// after the run: [pack, first slice in, last slice in, SpMV (send ready -> accumulators complete, exchange waits inside), apply,
// rest of the iteration (next messages, all-reduce, host round trip), mode]; -1 = not measured
int collect_times(gt_dist *d, uint32_t iterations, bool comm_ticks) {
    d->iter_ms.clear();
    if (!d->timing) return GT_OK;
    const uint32_t n = std::min(iterations, std::min(d->t_iter, GT_DIST_TIMED_ITERS));
    auto ms = [&](uint32_t it, int a, uint32_t it2, int b) -> double {
        float t = 0;
        return hipEventElapsedTime(&t, d->tev[(size_t)it * GT_DIST_TICKS + a], d->tev[(size_t)it2 * GT_DIST_TICKS + b]) == hipSuccess ? (double)t : -1.0;
    };
    for (uint32_t it = 0; it < n; it++) {
        d->iter_ms.push_back(ms(it, T_START, it, T_SEND_READY));
        d->iter_ms.push_back(comm_ticks ? ms(it, T_SEND_READY, it, T_FIRST_SLICE) : -1.0);
        d->iter_ms.push_back(comm_ticks ? ms(it, T_SEND_READY, it, T_LAST_SLICE) : -1.0);
        d->iter_ms.push_back(ms(it, T_SEND_READY, it, T_SPMV_DONE));
        d->iter_ms.push_back(ms(it, T_SPMV_DONE, it, T_APPLY_DONE));
        d->iter_ms.push_back(it + 1 < n ? ms(it, T_APPLY_DONE, it + 1, T_START) : -1.0);
        d->iter_ms.push_back(it < d->t_mode.size() ? (double)d->t_mode[it] : 0.0);
    }
    (void)hipGetLastError();
    return GT_OK;
}

int sync_deadline(gt_dist *d, hipStream_t s, const char *what);
int dist_all_reduce_words(gt_dist *d, uint64_t *v, uint32_t count, hipStream_t s) {
    if (d->loop) {
        LoopCtx &c = *d->loop;
        for (uint32_t i = 0; i < count; i++) {     // one word at a time: this is test plumbing
            c.peer[d->rank].word = v[i];
            GT_LOOP_BARRIER(c);
            uint64_t sum = 0;
            for (int r = 0; r < c.n; r++) sum += c.peer[r].word;
            GT_LOOP_BARRIER(c);
            v[i] = sum;
        }
        return GT_OK;
    }
    GT_REQUIRE(count <= 64, GT_ERR_INVALID, "at most 64 words per all-reduce");
    GT_HIP(hipMemcpyAsync(d->d_word, v, (size_t)count * 8, hipMemcpyHostToDevice, s));
    GT_NCCL(rccl()->AllReduce(d->d_word, d->d_word, count, ncclUint64, ncclSum, d->comm, s));
    GT_HIP(hipMemcpyAsync(v, d->d_word, (size_t)count * 8, hipMemcpyDeviceToHost, s));
    return sync_deadline(d, s, "an all-reduce of 8-byte words");
}

static inline bool block_sparse(uint32_t count, uint32_t len) { return len != 0 && 2ull * count < len; }   // pairs cost 8 B, a dense message 4

// waits for a stream with a deadline: an exchange that never completes (a peer died, a plan went wrong) must end the run with a
// message, not hang it (GRAPHTAP_DIST_TIMEOUT_S, default 300)
int sync_deadline(gt_dist *d, hipStream_t s, const char *what) {
    if (d->dead) { gt_set_error("rank %d: the communicator was aborted after an earlier deadline; no further collective runs on it", d->rank); return GT_ERR_TIMEOUT; }
    const double limit = gt_wait_limit_s();   // GRAPHTAP_TIMEOUT_S, else GRAPHTAP_DIST_TIMEOUT_S, default 300 s
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return GT_OK;
        if (e != hipErrorNotReady) { gt_set_error("rank %d: %s failed: %s", d->rank, what, hipGetErrorString(e)); return GT_ERR_HIP; }
        if ((spins & 1023u) == 1023u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
            // The RCCL kernel this stream waits for would also block every hipFree / hipDeviceSynchronize of the error path
            // (gt_graph_free, the scratch pool): the communicator is aborted first, which ends its kernels, so that the error
            // can be reported at all.
            if (!d->loop && d->comm && d->own_comm && rccl() && rccl()->CommAbort) { (void)rccl()->CommAbort(d->comm); d->comm = nullptr; }   // (a communicator handed in by the caller is the caller's to abort)
            d->dead = true;
            gt_set_error("rank %d of %d: %s did not complete within %g s (a peer gone, or an exchange that does not match its plan); the communicator was aborted",
                         d->rank, d->nranks, what, limit);
            return GT_ERR_TIMEOUT;
        }
    }
}

// Once per (communicator, graph): every rank's send counts against every receiver's recv counts. A mismatch (graphs built from
// different edge lists or flags, a different GRAPHTAP_X_SLICES on some rank) would make the grouped send/recv rounds hang or
// scribble; it is reported instead.
int plan_verify(gt_dist *d, const gt_graph *g) {
    if (d->plan_checked == g->serial) return GT_OK;
    const uint32_t K = g->info.x_slices, P = g->info.nranks, me = g->info.rank;
    {   // first the shape (a fixed-size collective): the ranks must agree on K and P before a K x P x P matrix can be summed
        uint64_t w[3] = {K, (uint64_t)K * K, 1};
        int st = gt_dist_all_reduce_sum_u64_host(d, w, 3); if (st != GT_OK) return st;
        GT_REQUIRE(w[2] == P && w[0] == (uint64_t)K * P && w[1] == (uint64_t)K * K * P, GT_ERR_STATE,
                   "exchange plan: the ranks disagree on the number of slices or of ranks (this rank: %u slices, %u ranks; %llu ranks answered, their slices sum to %llu): "
                   "the same GRAPHTAP_X_SLICES and nranks everywhere?", K, P, (unsigned long long)w[2], (unsigned long long)w[0]);
    }
    std::vector<uint64_t> m((size_t)2 * P * K * P, 0);   // [0]: what r sends q in slice k at [(r*K + k)*P + q]; [1]: what q expects from r, same index
    const size_t half = (size_t)P * K * P;
    for (uint32_t k = 0; k < K; k++) for (uint32_t q = 0; q < P; q++) {
        m[((size_t)me * K + k) * P + q] = g->send_counts[(size_t)k * P + q];
        m[half + ((size_t)q * K + k) * P + me] = g->recv_counts[(size_t)k * P + q];
    }
    { int st = gt_dist_all_reduce_sum_u64_host(d, m.data(), (uint32_t)m.size()); if (st != GT_OK) return st; }
    for (size_t i = 0; i < half; i++)
        GT_REQUIRE(m[i] == m[half + i], GT_ERR_STATE, "exchange plan mismatch: rank %zu sends %llu elements to rank %zu in slice %zu, which expects %llu "
                   "(were the graphs built from the same edge list, flags and GRAPHTAP_X_SLICES?)", i / ((size_t)K * P), (unsigned long long)m[i], i % P, (i / P) % K,
                   (unsigned long long)m[half + i]);
    d->plan_checked = g->serial;
    return GT_OK;
}

// block tables of the graph an execute runs on (host + device), the CSR of send positions, the all-reduce words
int tables_prepare(gt_dist *d, gt_program *p) {
    const gt_graph *g = p->g;
    if (d->sp_graph == g) return GT_OK;
    const uint32_t K = g->info.x_slices, P = g->info.nranks, NB = K * P;
    d->stab.assign(NB, BlockTab{0, 0}); d->rtab.assign(NB, BlockTab{0, 0});
    for (uint32_t k = 0; k < K; k++) {
        uint64_t so = g->send_off[k], ro = g->recv_off[k];
        for (uint32_t q = 0; q < P; q++) {
            d->stab[k * P + q] = BlockTab{(uint32_t)so, g->send_counts[(size_t)k * P + q]};
            d->rtab[k * P + q] = BlockTab{(uint32_t)ro, g->recv_counts[(size_t)k * P + q]};
            so += g->send_counts[(size_t)k * P + q]; ro += g->recv_counts[(size_t)k * P + q];
        }
    }
    for (void **q : {(void **)&d->d_stab, (void **)&d->d_cnt, (void **)&d->d_cur, (void **)&d->d_smode, (void **)&d->d_cx, (void **)&d->d_rtab, (void **)&d->d_ftab,
                     (void **)&d->d_rsel, (void **)&d->d_flag, (void **)&d->d_words, (void **)&d->send_ptr, (void **)&d->send_pos})
        if (*q) { (void)hipFree(*q); *q = nullptr; }
    GT_HIP(hipMalloc((void **)&d->d_stab, NB * sizeof(BlockTab))); GT_HIP(hipMalloc((void **)&d->d_cnt, NB * 4)); GT_HIP(hipMalloc((void **)&d->d_cur, NB * 4));
    GT_HIP(hipMalloc((void **)&d->d_smode, NB)); GT_HIP(hipMalloc((void **)&d->d_cx, 2ull * NB * 4));
    GT_HIP(hipMalloc((void **)&d->d_rtab, NB * sizeof(BlockTab))); GT_HIP(hipMalloc((void **)&d->d_ftab, NB * sizeof(FrontTab)));
    GT_HIP(hipMalloc((void **)&d->d_rsel, NB)); GT_HIP(hipMalloc((void **)&d->d_flag, 4));
    d->nwords = 2 + P * NB;
    GT_HIP(hipMalloc((void **)&d->d_words, (size_t)d->nwords * 8)); d->h_words.assign(d->nwords, 0);
    GT_HIP(hipMemcpy(d->d_stab, d->stab.data(), NB * sizeof(BlockTab), hipMemcpyHostToDevice));
    GT_HIP(hipMemcpy(d->d_rtab, d->rtab.data(), NB * sizeof(BlockTab), hipMemcpyHostToDevice));
    if (d->pairs_send_cap < std::max<uint64_t>(g->send_elems, 1)) { if (d->pairs_send) GT_HIP(hipFree(d->pairs_send)); d->pairs_send = nullptr; GT_HIP(hipMalloc((void **)&d->pairs_send, std::max<uint64_t>(g->send_elems, 1) * 8)); d->pairs_send_cap = std::max<uint64_t>(g->send_elems, 1); }
    if (d->pairs_recv_cap < std::max<uint32_t>(g->ncols_total, 1)) { if (d->pairs_recv) GT_HIP(hipFree(d->pairs_recv)); d->pairs_recv = nullptr; GT_HIP(hipMalloc((void **)&d->pairs_recv, (uint64_t)std::max<uint32_t>(g->ncols_total, 1) * 8)); d->pairs_recv_cap = std::max<uint32_t>(g->ncols_total, 1); }
    d->cnt_send.assign(NB, 0); d->cnt_recv.assign(NB, 0); d->form_send.assign(NB, 0); d->form_recv.assign(NB, 0);
    // send positions per owned column (min programs: the list -> pairs kernel walks them)
    const uint32_t n = (uint32_t)g->send_elems, nc = g->info.nnzcols;
    GT_HIP(hipMalloc((void **)&d->send_ptr, ((uint64_t)nc + 2) * 4)); GT_HIP(hipMalloc((void **)&d->send_pos, std::max<uint64_t>(n, 1) * 4));
    if (n && !p->stationary) {
        uint32_t *k1 = nullptr, *k2 = nullptr, *v1 = nullptr; void *tmp = nullptr;
        GT_HIP(hipMalloc((void **)&k1, (uint64_t)n * 4)); GT_HIP(hipMalloc((void **)&k2, (uint64_t)n * 4)); GT_HIP(hipMalloc((void **)&v1, (uint64_t)n * 4));
        const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)n + 255) / 256, 4096);
        k_send_keys<<<grid, 256>>>(g->send_idx, n, k1, v1);
        size_t tb = 0;
        int bits = 1; while ((1ull << bits) < (uint64_t)nc + 1) bits++;
        GT_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k1, k2, v1, d->send_pos, n, 0, bits, (hipStream_t)0));
        GT_HIP(hipMalloc(&tmp, tb ? tb : 1));
        GT_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k1, k2, v1, d->send_pos, n, 0, bits, (hipStream_t)0));
        k_send_ptr<<<(unsigned)std::min<uint64_t>(((uint64_t)nc + 256) / 256, 4096), 256>>>(k2, n, nc, d->send_ptr);
        GT_HIP(hipDeviceSynchronize());
        (void)hipFree(k1); (void)hipFree(k2); (void)hipFree(v1); (void)hipFree(tmp);
    } else GT_HIP(hipMemset(d->send_ptr, 0, ((uint64_t)nc + 2) * 4));
    // a list of at most this many vertices leaves every block it touches below half full: pairs are then never more bytes than the dense block
    uint32_t minlen = 0xFFFFFFFFu;
    for (uint32_t b = 0; b < NB; b++) if (d->stab[b].len) minlen = std::min(minlen, d->stab[b].len);
    d->list_send_cap = minlen == 0xFFFFFFFFu ? 0u : (minlen - 1) / 2;
    const char *e = getenv("GRAPHTAP_DIST_LISTS");   // "1": lists of any length travel as pairs, "0": never
    if (e && atoi(e) == 1) d->list_send_cap = 0x7FFFFFFFu;
    if (e && atoi(e) == 0) d->list_send_cap = 0;
    d->sp_graph = g;
    return GT_OK;
}

// the blocks of x that arrive as pairs start from infinity(): only their active messages arrive (one launch for all of them)
int fill_pair_blocks(gt_dist *d, gt_program *p, hipStream_t s) {
    const uint32_t NB = (uint32_t)d->rtab.size();
    uint32_t maxlen = 0;
    for (uint32_t b = 0; b < NB; b++) if (d->form_recv[b]) maxlen = std::max(maxlen, d->rtab[b].len);
    if (!maxlen) return GT_OK;
    GT_HIP(hipMemcpyAsync(d->d_rsel, d->form_recv.data(), NB, hipMemcpyHostToDevice, s));
    k_fill_blocks<<<dim3(std::min<uint32_t>((maxlen + 255) / 256, 1024), NB), 256, 0, s>>>((uint32_t *)p->x, d->d_rtab, d->d_rsel, GT_INF);
    GT_HIP(hipGetLastError());
    return GT_OK;
}

// min programs: counts the active messages of every send block, tells every peer, compacts the sparse blocks into pairs
int sparse_prepare(gt_dist *d, gt_program *p, hipStream_t s) {
    d->sparse_now = false;
    const gt_graph *g = p->g;
    const char *env = getenv("GRAPHTAP_SPARSE_EXCHANGE");
    if (p->stationary || p->x_bytes != 4 || (env && atoi(env) == 0)) return GT_OK;   // (the same decision on every rank)
    const uint32_t K = g->info.x_slices, P = g->info.nranks, NB = K * P;
    { int st = tables_prepare(d, p); if (st != GT_OK) return st; }
    const uint32_t n = (uint32_t)g->send_elems;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)n + 255) / 256, 4096));
    GT_HIP(hipMemsetAsync(d->d_cnt, 0, NB * 4, s));
    k_block_active<<<grid, 256, 0, s>>>((const uint32_t *)p->send, n, d->d_stab, NB, d->d_cnt);
    GT_HIP(hipMemcpyAsync(d->cnt_send.data(), d->d_cnt, NB * 4, hipMemcpyDeviceToHost, s));
    GT_HIP(hipStreamSynchronize(s));
    // every peer learns how many active messages each of its blocks from me holds (and so which form they take)
    if (d->loop) {
        LoopCtx &c = *d->loop;
        c.peer[d->rank].counts = d->cnt_send.data();
        GT_LOOP_BARRIER(c);
        for (uint32_t k = 0; k < K; k++) for (uint32_t q = 0; q < P; q++) d->cnt_recv[k * P + q] = c.peer[q].counts[k * P + d->rank];
        GT_LOOP_BARRIER(c);
    } else {
        std::vector<uint32_t> out(NB), in(NB);   // [peer][slice]
        for (uint32_t k = 0; k < K; k++) for (uint32_t q = 0; q < P; q++) out[q * K + k] = d->cnt_send[k * P + q];
        GT_HIP(hipMemcpyAsync(d->d_cx, out.data(), NB * 4, hipMemcpyHostToDevice, s));
        GT_NCCL(rccl()->GroupStart());
        for (uint32_t q = 0; q < P; q++) {
            GT_NCCL(rccl()->Send(d->d_cx + q * K, K, ncclUint32, (int)q, d->comm, s));
            GT_NCCL(rccl()->Recv(d->d_cx + NB + q * K, K, ncclUint32, (int)q, d->comm, s));
        }
        GT_NCCL(rccl()->GroupEnd());
        GT_HIP(hipMemcpyAsync(in.data(), d->d_cx + NB, NB * 4, hipMemcpyDeviceToHost, s));
        GT_HIP(hipStreamSynchronize(s));
        for (uint32_t k = 0; k < K; k++) for (uint32_t q = 0; q < P; q++) d->cnt_recv[k * P + q] = in[q * K + k];
    }
    bool any = false;
    for (uint32_t b = 0; b < NB; b++) {
        d->form_send[b] = block_sparse(d->cnt_send[b], d->stab[b].len); any |= d->form_send[b] != 0;
        d->form_recv[b] = block_sparse(d->cnt_recv[b], d->rtab[b].len);
    }
    if (any) {
        GT_HIP(hipMemcpyAsync(d->d_smode, d->form_send.data(), NB, hipMemcpyHostToDevice, s));
        GT_HIP(hipMemsetAsync(d->d_cur, 0, NB * 4, s));
        k_block_compact<<<grid, 256, 0, s>>>((const uint32_t *)p->send, n, d->d_stab, NB, d->d_smode, d->d_cur, d->pairs_send);
    }
    { int st = fill_pair_blocks(d, p, s); if (st != GT_OK) return st; }
    d->sparse_now = true;
    return GT_OK;
}

// the K slices of one iteration's exchange are ISSUED here (all of them); consume(k) then makes `s` wait for slice k
int exchange_issue(gt_dist *d, gt_program *p, hipStream_t s, bool one_round = false) {
    const gt_graph *g = p->g;
    const uint32_t K = g->info.x_slices, P = g->info.nranks, w = p->x_bytes;
    if (!d->lists_protocol) { int st = sparse_prepare(d, p, s); if (st != GT_OK) return st; }   // (the list protocol has counts and forms already)
    const bool sp = d->sparse_now;
    d->exchanges++;
    if (d->loop) {
        LoopCtx &c = *d->loop;
        c.peer[d->rank].pairs = d->pairs_send;
        GT_HIP(hipStreamSynchronize(s));                      // my send buffer is packed
        GT_LOOP_BARRIER(c);
        for (uint32_t k = 0; k < K; k++) {
            uint64_t dst = g->recv_off[k];
            for (uint32_t src = 0; src < P; src++) {
                const gt_graph *gs = c.peer[src].g;
                uint64_t off = gs->send_off[k];
                for (uint32_t q = 0; q < (uint32_t)d->rank; q++) off += gs->send_counts[(size_t)k * P + q];
                const uint32_t n = g->recv_counts[(size_t)k * P + src];
                GT_REQUIRE(n == gs->send_counts[(size_t)k * P + d->rank], GT_ERR_STATE, "exchange plan mismatch between ranks %u and %d (slice %u)", src, d->rank, k);
                const uint32_t cnt = sp ? d->cnt_recv[k * P + src] : 0;
                if (sp && d->form_recv[k * P + src]) {
                    if (cnt) GT_HIP(hipMemcpyAsync(d->pairs_recv + dst, c.peer[src].pairs + off, (uint64_t)cnt * 8, hipMemcpyDeviceToDevice, s));
                } else if (n) GT_HIP(hipMemcpyAsync((char *)p->x + dst * w, c.peer[src].send + off * w, (uint64_t)n * w, hipMemcpyDeviceToDevice, s));
                dst += n;
            }
        }
        for (uint32_t b = 0; b < K * P; b++) {   // what this rank SENDS (the mirror image of the copies above)
            const uint32_t len = g->send_counts[b], cnt = sp ? d->cnt_send[b] : 0;
            d->bytes_dense += (uint64_t)len * w; d->bytes_sent += (sp && d->form_send[b]) ? (uint64_t)cnt * 8 : (uint64_t)len * w;
        }
        GT_HIP(hipStreamSynchronize(s));
        GT_LOOP_BARRIER(c);                                          // every rank has read every send buffer
        return GT_OK;
    }
    const ncclDataType_t ty = (w == 8) ? ncclUint64 : ncclUint32;
    GT_HIP(hipEventRecord(d->ev_ready, s));
    GT_HIP(hipStreamWaitEvent(d->comm_stream, d->ev_ready, 0));   // sends read what scatter_gather packed; receives overwrite an x nobody reads any more
    while (d->ev_slice.size() < K) { hipEvent_t e; GT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); d->ev_slice.push_back(e); }
    while (d->ev_pack.size() < K) { hipEvent_t e; GT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); d->ev_pack.push_back(e); }
    // slice by slice, ON THE COMMUNICATION STREAM (round 4; on the compute stream until round 3): the sends of slice k follow ITS
    // packing, and the compute stream -- which has nothing to do with the send buffer -- goes straight on to wait for slice 0
    // (k_pack_send of a tile-row of 8 is 0.05 ms, 0.026 of it used to sit in front of the first send AND of phase 1)
    // one_round: every slice in ONE grouped round (a list iteration consumes all of them at once, and its messages are a few
    // pairs: K rounds of ~25 us each were most of its exchange)
    if (one_round) GT_NCCL(rccl()->GroupStart());
    for (uint32_t k = 0; k < K; k++) {
        uint64_t so = g->send_off[k], ro = g->recv_off[k];
        if (p->pack_deferred) { int st = gt_program_pack_slice_on(p, k, d->comm_stream); if (st != GT_OK) return st; }
        if (!one_round) GT_NCCL(rccl()->GroupStart());
        for (uint32_t q = 0; q < P; q++) {
            const uint32_t ns = g->send_counts[(size_t)k * P + q], nr = g->recv_counts[(size_t)k * P + q];
            const uint32_t cs = sp ? d->cnt_send[k * P + q] : 0, cr = sp ? d->cnt_recv[k * P + q] : 0;
            d->bytes_dense += (uint64_t)ns * w;
            if (sp && d->form_send[k * P + q]) { if (cs) GT_NCCL(rccl()->Send(d->pairs_send + so, 2ull * cs, ncclUint32, (int)q, d->comm, d->comm_stream)); d->bytes_sent += 8ull * cs; }
            else if (ns) { GT_NCCL(rccl()->Send((const char *)p->send + so * w, ns, ty, (int)q, d->comm, d->comm_stream)); d->bytes_sent += (uint64_t)ns * w; }
            if (sp && d->form_recv[k * P + q]) { if (cr) GT_NCCL(rccl()->Recv(d->pairs_recv + ro, 2ull * cr, ncclUint32, (int)q, d->comm, d->comm_stream)); }
            else if (nr) GT_NCCL(rccl()->Recv((char *)p->x + ro * w, nr, ty, (int)q, d->comm, d->comm_stream));
            so += ns; ro += nr;
        }
        if (one_round) continue;
        GT_NCCL(rccl()->GroupEnd());
        GT_HIP(hipEventRecord(d->ev_slice[k], d->comm_stream));
        if (k == 0) { int st = tick(d, T_FIRST_SLICE, d->comm_stream); if (st != GT_OK) return st; }
        if (k + 1 == K) { int st = tick(d, T_LAST_SLICE, d->comm_stream); if (st != GT_OK) return st; }
    }
    if (one_round) {
        GT_NCCL(rccl()->GroupEnd());
        for (uint32_t k = 0; k < K; k++) GT_HIP(hipEventRecord(d->ev_slice[k], d->comm_stream));
        int st = tick(d, T_FIRST_SLICE, d->comm_stream); if (st != GT_OK) return st;
        st = tick(d, T_LAST_SLICE, d->comm_stream); if (st != GT_OK) return st;
    }
    return GT_OK;
}
// ONE slice of a dense exchange (stationary programs), issued as soon as ITS messages are final: the pipelined PageRank loop
// calls it after part k of phase 2 (gt_program_phase2_part), so that slice k is packed and travels while the later parts -- and
// then the receiver's phase 1 of the earlier slices -- run. `it` = the iteration that will consume the slice (for the
// diagnostics). RCCL: the packing kernel itself runs on the communication stream, off the compute stream's critical path.
int exchange_issue_slice(gt_dist *d, gt_program *p, hipStream_t s, uint32_t k, uint32_t it, hipEvent_t ready = nullptr) {
    const gt_graph *g = p->g;
    const uint32_t K = g->info.x_slices, P = g->info.nranks, w = p->x_bytes;
    if (k == 0) d->exchanges++;
    uint64_t so = g->send_off[k], ro = g->recv_off[k];
    for (uint32_t q = 0; q < P; q++) { const uint64_t b = (uint64_t)g->send_counts[(size_t)k * P + q] * w; d->bytes_dense += b; d->bytes_sent += b; }
    if (d->loop) {
        LoopCtx &c = *d->loop;
        int st = gt_program_pack_slice(p, k); if (st != GT_OK) return st;
        GT_HIP(hipStreamSynchronize(s));                      // slice k of my send buffer is packed
        GT_LOOP_BARRIER(c);
        uint64_t dst = ro;
        for (uint32_t src = 0; src < P; src++) {
            const gt_graph *gs = c.peer[src].g;
            uint64_t off = gs->send_off[k];
            for (uint32_t q = 0; q < (uint32_t)d->rank; q++) off += gs->send_counts[(size_t)k * P + q];
            const uint32_t n = g->recv_counts[(size_t)k * P + src];
            GT_REQUIRE(n == gs->send_counts[(size_t)k * P + d->rank], GT_ERR_STATE, "exchange plan mismatch between ranks %u and %d (slice %u)", src, d->rank, k);
            if (n) GT_HIP(hipMemcpyAsync((char *)p->x + dst * w, c.peer[src].send + off * w, (uint64_t)n * w, hipMemcpyDeviceToDevice, s));
            dst += n;
        }
        GT_HIP(hipStreamSynchronize(s));
        GT_LOOP_BARRIER(c);                                   // every rank has read slice k of every send buffer
        return GT_OK;
    }
    const ncclDataType_t ty = (w == 8) ? ncclUint64 : ncclUint32;
    while (d->ev_slice.size() < K) { hipEvent_t e; GT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); d->ev_slice.push_back(e); }
    while (d->ev_pack.size() < K) { hipEvent_t e; GT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); d->ev_pack.push_back(e); }
    // `ready`: the messages of slice k are final and nobody reads slice k of x any more (the event behind part k of phase 2, or -- the
    // first exchange of a run -- the point the compute stream has reached)
    if (!ready) { GT_HIP(hipEventRecord(d->ev_pack[k], s)); ready = d->ev_pack[k]; }
    GT_HIP(hipStreamWaitEvent(d->comm_stream, ready, 0));
    { int st = gt_program_pack_slice_on(p, k, d->comm_stream); if (st != GT_OK) return st; }
    GT_NCCL(rccl()->GroupStart());
    for (uint32_t q = 0; q < P; q++) {
        const uint32_t ns = g->send_counts[(size_t)k * P + q], nr = g->recv_counts[(size_t)k * P + q];
        if (ns) GT_NCCL(rccl()->Send((const char *)p->send + so * w, ns, ty, (int)q, d->comm, d->comm_stream));
        if (nr) GT_NCCL(rccl()->Recv((char *)p->x + ro * w, nr, ty, (int)q, d->comm, d->comm_stream));
        so += ns; ro += nr;
    }
    GT_NCCL(rccl()->GroupEnd());
    GT_HIP(hipEventRecord(d->ev_slice[k], d->comm_stream));
    if (k == 0) { int st = tick_at(d, it, T_FIRST_SLICE, d->comm_stream); if (st != GT_OK) return st; }
    if (k + 1 == K) { int st = tick_at(d, it, T_LAST_SLICE, d->comm_stream); if (st != GT_OK) return st; }
    return GT_OK;
}
int exchange_consume(gt_dist *d, gt_program *p, uint32_t k, hipStream_t s) {
    if (!d->loop) GT_HIP(hipStreamWaitEvent(s, d->ev_slice[k], 0));
    if (d->sparse_now) {   // the pairs of slice k's sparse blocks land in x
        const uint32_t P = p->g->info.nranks;
        for (uint32_t q = 0; q < P; q++) {
            const BlockTab t = d->rtab[k * P + q];
            const uint32_t cnt = d->cnt_recv[k * P + q];
            if (d->form_recv[k * P + q] && cnt) k_scatter_pairs<<<(cnt + 255) / 256, 256, 0, s>>>(d->pairs_recv + t.start, cnt, (uint32_t *)p->x + t.start);
        }
        GT_HIP(hipGetLastError());
    }
    return GT_OK;
}

// ---- the converge-mode loop of the min programs (BFS / SSSP / CC): frontier lists on every rank
// After apply() every rank turns the list of the vertices it just changed into (index, value) pairs per destination block
// (k_pairs_from_list; a list that is too long raises a flag instead), and ONE all-reduce carries the convergence count, the
// flags and every block's pair count: one host round trip per iteration while the frontiers are lists (the reference's count
// header, vp:766-773, rides its broadcasts the same way). A rank whose list did not fit packs dense messages and a second
// all-reduce tells its counts (the iterations with large frontiers, whose compute dwarfs a round trip). A rank that received
// nothing but pairs, few of them, runs the SpMSpV straight from the pairs (local column = block start + index) and applies
// the rows it lowered; otherwise pairs are scattered into x and the streaming pass runs as before.
int all_reduce_vec(gt_dist *d, hipStream_t s, const unsigned long long *extra_dev, unsigned long long *extra_host, const uint32_t *flag_dev, uint32_t *flag_host) {
    d->round_trips++;
    if (extra_dev) GT_HIP(hipMemcpyAsync(extra_host, extra_dev, 8, hipMemcpyDeviceToHost, s));   // this rank's own active count, before the sum
    if (flag_dev) GT_HIP(hipMemcpyAsync(flag_host, flag_dev, 4, hipMemcpyDeviceToHost, s));
    if (d->loop) {
        LoopCtx &c = *d->loop;
        std::vector<unsigned long long> mine(d->nwords);
        GT_HIP(hipMemcpyAsync(mine.data(), d->d_words, (size_t)d->nwords * 8, hipMemcpyDeviceToHost, s));
        GT_HIP(hipStreamSynchronize(s));
        c.peer[d->rank].words = mine.data();
        GT_LOOP_BARRIER(c);
        for (uint32_t i = 0; i < d->nwords; i++) { unsigned long long t = 0; for (int r = 0; r < c.n; r++) t += c.peer[r].words[i]; d->h_words[i] = t; }
        GT_LOOP_BARRIER(c);
        return GT_OK;
    }
    GT_NCCL(rccl()->AllReduce(d->d_words, d->d_words, d->nwords, ncclUint64, ncclSum, d->comm, s));
    GT_HIP(hipMemcpyAsync(d->h_words.data(), d->d_words, (size_t)d->nwords * 8, hipMemcpyDeviceToHost, s));
    return sync_deadline(d, s, "the per-iteration all-reduce (convergence word + pair counts)");
}
// counts and forms of what I send / receive, out of the all-reduced words
void take_counts(gt_dist *d, uint32_t NB) {
    const uint32_t P = (uint32_t)d->nranks, K = NB / P, me = (uint32_t)d->rank;
    for (uint32_t b = 0; b < NB; b++) {
        const unsigned long long w = d->h_words[2 + (size_t)me * NB + b];
        d->cnt_send[b] = (uint32_t)w; d->form_send[b] = (uint8_t)((w >> 32) & 1u);
    }
    for (uint32_t k = 0; k < K; k++) for (uint32_t q = 0; q < P; q++) {   // block (k, source q) of x = block (k, me) of rank q's send buffer
        const unsigned long long w = d->h_words[2 + (size_t)q * NB + k * P + me];
        d->cnt_recv[k * P + q] = (uint32_t)w; d->form_recv[k * P + q] = (uint8_t)((w >> 32) & 1u);
    }
}
// this rank's messages for the coming iteration: pairs from its list if the list is short enough (decided on the device), and
// the words of the all-reduce. `d_active` = the device word of the apply just launched (null before the first iteration).
int lists_prepare_send(gt_dist *d, gt_program *p, hipStream_t s, const unsigned long long *d_active) {
    const gt_graph *g = p->g;
    const uint32_t NB = (uint32_t)d->stab.size();
    GT_HIP(hipMemsetAsync(d->d_cur, 0, NB * 4, s));
    GT_HIP(hipMemsetAsync(d->d_flag, 0, 4, s));
    // the list the apply just launched is writing (deferred apply: fl_cur is flipped by apply_end afterwards), or the initial one
    const int li = d_active ? (p->fl_cur ^ 1) : p->fl_cur;
    const bool may = p->fl_enabled && (d_active != nullptr || p->fl_cur_valid);
    if (may) {
        const uint32_t cap = std::min(d->list_send_cap, p->fl_cap);
        k_pairs_from_list<<<1024, 256, 0, s>>>(p->fl_v[li], p->d_fl + li, cap, g->IJ, g->JV, d->send_ptr, d->send_pos, p->s0, g->info.rank * g->info.tile_height,
                                               gt_vidmap_of(g), p->prm.kind, d->d_stab, NB, d->d_cur, d->pairs_send, d->d_flag);
    } else {
        static const uint32_t one = 1;
        GT_HIP(hipMemcpyAsync(d->d_flag, &one, 4, hipMemcpyHostToDevice, s));
    }
    k_words<<<(d->nwords + 255) / 256, 256, 0, s>>>(d->d_words, d->nwords, d_active, d->d_flag, d->d_cur, nullptr, (uint32_t)d->rank, NB);
    GT_HIP(hipGetLastError());
    return GT_OK;
}
// the dense round of a rank whose list did not fit: messages of all owned columns, packed; per-block counts; the blocks that are
// sparse by the byte rule compacted into pairs; words for the second all-reduce
int dense_prepare_send(gt_dist *d, gt_program *p, hipStream_t s, bool i_need_dense) {
    const gt_graph *g = p->g;
    const uint32_t NB = (uint32_t)d->stab.size(), n = (uint32_t)g->send_elems;
    if (i_need_dense) {
        int st = gt_program_scatter_gather(p); if (st != GT_OK) return st;
        const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(((uint64_t)n + 255) / 256, 4096));
        GT_HIP(hipMemsetAsync(d->d_cnt, 0, NB * 4, s));
        k_block_active<<<grid, 256, 0, s>>>((const uint32_t *)p->send, n, d->d_stab, NB, d->d_cnt);
        // forms by the byte rule, on the host: this rank's own counts are needed for it (a local read; the peers learn them from the all-reduce)
        GT_HIP(hipMemcpyAsync(d->cnt_send.data(), d->d_cnt, NB * 4, hipMemcpyDeviceToHost, s));
        GT_HIP(hipStreamSynchronize(s));
        bool any = false;
        for (uint32_t b = 0; b < NB; b++) { d->form_send[b] = block_sparse(d->cnt_send[b], d->stab[b].len); any |= d->form_send[b] != 0; }
        GT_HIP(hipMemcpyAsync(d->d_smode, d->form_send.data(), NB, hipMemcpyHostToDevice, s));
        if (any) {
            GT_HIP(hipMemsetAsync(d->d_cur, 0, NB * 4, s));
            k_block_compact<<<grid, 256, 0, s>>>((const uint32_t *)p->send, n, d->d_stab, NB, d->d_smode, d->d_cur, d->pairs_send);
        }
        k_words<<<(d->nwords + 255) / 256, 256, 0, s>>>(d->d_words, d->nwords, nullptr, nullptr, d->d_cnt, d->d_smode, (uint32_t)d->rank, NB);
    } else {   // my pairs stand: the same counts again (every block in pair form)
        k_words<<<(d->nwords + 255) / 256, 256, 0, s>>>(d->d_words, d->nwords, nullptr, nullptr, d->d_cur, nullptr, (uint32_t)d->rank, NB);
    }
    GT_HIP(hipGetLastError());
    return GT_OK;
}
// the SpMV of a rank that received nothing but (few) pairs: the frontier is the pairs themselves
// `exact`: the entries of the frontier's columns are counted first and the SpMSpV runs only up to nnz / 32 of them (one more host
// read -- only for frontiers of more than a few thousand columns, whose compute dwarfs it); *done = false leaves the pairs to the
// caller, who scatters them into x for the streaming pass.
int combine_from_pairs(gt_dist *d, gt_program *p, hipStream_t s, uint32_t total, bool exact, uint64_t max_entries, bool *done) {
    const gt_graph *g = p->g;
    const uint32_t NB = (uint32_t)d->rtab.size(), K = g->info.x_slices;
    *done = true;
    if (!d->loop) for (uint32_t k = 0; k < K; k++) GT_HIP(hipStreamWaitEvent(s, d->ev_slice[k], 0));
    GT_HIP(hipMemsetAsync(p->d_fl + 2, 0, sizeof(unsigned int), s));
    p->fl_rows_valid = true;
    if (total == 0) return GT_OK;   // nothing is active for this tile-row: y keeps its running minima
    d->h_ftab.resize(NB);   // (a member: the copy below may read it after this function returned; it is rewritten after the next round trip)
    uint32_t o = 0;
    for (uint32_t b = 0; b < NB; b++) { d->h_ftab[b] = FrontTab{o, d->cnt_recv[b]}; o += d->cnt_recv[b]; }
    GT_HIP(hipMemcpyAsync(d->d_ftab, d->h_ftab.data(), NB * sizeof(FrontTab), hipMemcpyHostToDevice, s));
    if (p->fr_cap < total + 1) { int st = gt_spmspv_reserve(p, total); if (st != GT_OK) return st; }
    if (exact) GT_HIP(hipMemsetAsync(p->d_frontier + 1, 0, sizeof(unsigned long long), s));
    k_pairs_frontier<<<(unsigned)std::min<uint64_t>(((uint64_t)total + 255) / 256, 4096), 256, 0, s>>>(d->pairs_recv, d->d_rtab, d->d_ftab, NB, total, g->JA,
                                                                                                      p->fr_col, p->fr_val, p->fr_off, exact ? p->d_frontier + 1 : nullptr);
    GT_HIP(hipGetLastError());
    if (exact) {
        unsigned long long entries = 0;
        GT_HIP(hipMemcpyAsync(&entries, p->d_frontier + 1, sizeof(entries), hipMemcpyDeviceToHost, s));
        int st = sync_deadline(d, s, "the exchange of a list iteration"); if (st != GT_OK) return st;
        d->round_trips++;
        if (entries > max_entries) { *done = false; p->fl_rows_valid = false; return GT_OK; }
    }
    d->pair_spmspv_iters++;
    return gt_spmspv_run_frontier(p, total, s);
}

int lists_execute(gt_dist *d, gt_program *p, gt_exec_stats *stats, std::chrono::steady_clock::time_point t0) {
    const gt_graph *g = p->g;
    const uint32_t K = g->info.x_slices, P = g->info.nranks, NB = K * P;
    hipStream_t s = p->stream;
    { int st = tables_prepare(d, p); if (st != GT_OK) return st; }
    if (d->loop) d->loop->peer[d->rank].pairs = d->pairs_send;
    d->lists_protocol = true; d->sparse_now = true;
    d->list_iters = d->pair_spmspv_iters = d->round_trips = 0;
    const char *senv = getenv("GRAPHTAP_SPMSPV");
    const int spm = senv ? atoi(senv) : -1;   // 0: never the SpMSpV from pairs, 1: whenever everything arrived as pairs, unset: by size
    static const uint64_t frac = getenv("GRAPHTAP_SPMSPV_FRACTION") ? (uint64_t)atoll(getenv("GRAPHTAP_SPMSPV_FRACTION")) : 32;
    // the messages of iteration 0
    int st = lists_prepare_send(d, p, s, nullptr); if (st != GT_OK) return st;
    unsigned long long local_active = 0; uint32_t my_flag = 0;
    st = all_reduce_vec(d, s, nullptr, nullptr, d->d_flag, &my_flag); if (st != GT_OK) return st;
    for (;;) {
        st = tick(d, T_START, s); if (st != GT_OK) return st;
        if (d->h_words[1] != 0) {   // somebody's list did not fit: the dense round
            st = dense_prepare_send(d, p, s, my_flag != 0); if (st != GT_OK) return st;
            st = all_reduce_vec(d, s, nullptr, nullptr, nullptr, nullptr); if (st != GT_OK) return st;
        } else d->list_iters++;
        if (my_flag == 0) { p->x_stale = true; p->x_fresh = false; }   // no messenger ran for this iteration: the next dense one rewrites every message
        take_counts(d, NB);
        uint64_t total = 0; bool all_pairs = true;
        for (uint32_t b = 0; b < NB; b++) { if (d->form_recv[b]) total += d->cnt_recv[b]; else if (d->rtab[b].len) all_pairs = false; }
        // everything arrived as pairs: the SpMSpV straight from them -- at once for a few thousand columns, after counting their
        // entries otherwise (a frontier of hubs is the streaming pass's business: the single-rank rule, nnz / 32 entries)
        bool from_pairs = all_pairs && spm != 0 && total < 0x7FFFFFFFull && (spm == 1 || total * 8 <= g->info.nnz_local / frac);
        if (!from_pairs) { st = fill_pair_blocks(d, p, s); if (st != GT_OK) return st; }
        st = tick(d, T_SEND_READY, s); if (st != GT_OK) return st;
        // ONE grouped round for all K slices when every block of every rank travels as pairs (or is empty). The shape of the
        // exchange must be the same on every rank -- a rank with all its sends in one group against peers with K groups can
        // leave a peer's later group waiting behind an earlier one: a wait cycle at P > 2 -- so it is derived from the
        // all-reduced words, which every rank holds identically, never from what this rank alone received. (What a rank DOES
        // with its pairs -- the SpMSpV straight from them, or scattering them into x -- stays its own choice.)
        bool one_round = true;
        for (size_t i = 2; i < d->h_words.size() && one_round; i++) one_round = ((d->h_words[i] >> 32) & 1ull) != 0 || (uint32_t)d->h_words[i] == 0;
        st = exchange_issue(d, p, s, one_round); if (st != GT_OK) return st;
        if (from_pairs) {
            bool done = false;
            static const uint64_t exact_from = getenv("GRAPHTAP_DIST_EXACT_FROM") ? (uint64_t)atoll(getenv("GRAPHTAP_DIST_EXACT_FROM")) : 4096;   // pairs above which the entries are counted first
            static const uint64_t max_entries_env = getenv("GRAPHTAP_DIST_MAX_ENTRIES") ? (uint64_t)atoll(getenv("GRAPHTAP_DIST_MAX_ENTRIES")) : ~0ull;   // (tests: force the fallback)
            st = combine_from_pairs(d, p, s, (uint32_t)total, spm != 1 && total > exact_from, std::min<uint64_t>(g->info.nnz_local / frac, max_entries_env), &done); if (st != GT_OK) return st;
            if (!done) { from_pairs = false; st = fill_pair_blocks(d, p, s); if (st != GT_OK) return st; }
        }
        if (d->timing && d->t_iter < GT_DIST_TIMED_ITERS) { d->t_mode.resize(d->t_iter + 1); d->t_mode[d->t_iter] = from_pairs ? 2 : (my_flag == 0 ? 1 : 0); }
        if (!from_pairs) {
            for (uint32_t k = 0; k < K; k++) {
                st = exchange_consume(d, p, k, s); if (st != GT_OK) return st;
                st = (K > 1) ? gt_program_combine_slice(p, k) : gt_program_combine(p); if (st != GT_OK) return st;
            }
        }
        st = tick(d, T_SPMV_DONE, s); if (st != GT_OK) return st;
        st = gt_program_apply_begin(p, 0); if (st != GT_OK) return st;
        st = tick(d, T_APPLY_DONE, s); if (st != GT_OK) return st;
        d->t_iter++;
        st = lists_prepare_send(d, p, s, p->d_active); if (st != GT_OK) return st;
        st = all_reduce_vec(d, s, p->d_active, &local_active, d->d_flag, &my_flag); if (st != GT_OK) return st;   // the host round trip of the iteration
        st = gt_program_apply_end(p, local_active); if (st != GT_OK) return st;
        p->last_active = d->h_words[0];
        if (d->h_words[0] == 0) { st = gt_program_finish_converged(p); if (st != GT_OK) return st; break; }   // has_converged, vp:1918
    }
    d->lists_protocol = false; d->sparse_now = false;
    return GT_OK;
}

// ---- the pipelined loop of a fixed-count PageRank (stationary, K > 1 slices, fused applicator)
// Per iteration: phase 1 of slice k as soon as slice k has landed (as before); then phase 2 in K parts, and after part k -- whose
// fused applicator writes the messages of the columns that travel in slice k -- slice k of the NEXT iteration's exchange is packed
// (on the communication stream) and sent, while the compute stream goes on with part k+1. The exchange no longer waits for the
// whole applicator (dist.hip of round 3: ev_ready after apply), the packing leaves the compute stream altogether, and the first
// slice has the rest of phase 2 to travel in. A peer posts its receive for slice k only after ITS part k, i.e. after all of its
// phase 1 has read the previous x: nothing is overwritten early. (Replaces the bcast of every segment after its whole apply,
// vp:843-862, 1083-1111.)
int pipelined_execute(gt_dist *d, gt_program *p, uint32_t iters) {
    const gt_graph *g = p->g;
    const uint32_t K = g->info.x_slices;
    hipStream_t s = p->stream;
    // the K parts of phase 2 run one after the other on the compute stream; GRAPHTAP_P2_PARTS=2: side by side on streams of their own
    const bool concurrent = !d->loop && getenv("GRAPHTAP_P2_PARTS") && atoi(getenv("GRAPHTAP_P2_PARTS")) == 2;   // (measured slower: engine.hip, gt_program_parts_begin)
    d->sparse_now = false;     // (dense blocks only; a min program's last execute on this communicator may have left it set)
    int st = tick(d, T_START, s); if (st != GT_OK) return st;
    p->pack_deferred = true;   // scatter_gather writes the messages of the owned columns; the slices are packed where they are sent
    st = gt_program_scatter_gather(p); if (st != GT_OK) return st;
    for (uint32_t k = 0; k < K; k++) { st = exchange_issue_slice(d, p, s, k, 0); if (st != GT_OK) return st; }
    for (bool first = true;; first = false) {
        if (!first) { st = tick(d, T_START, s); if (st != GT_OK) return st; }
        st = gt_program_fuse_apply(p, iters, 0); if (st != GT_OK) return st;
        p->pr_state = gt_pr_state_mode(p, iters, false);
        GT_REQUIRE(gt_program_parts_begin(p), GT_ERR_STATE, "pipelined loop: the fused applicator is not available any more");
        st = tick(d, T_SEND_READY, s); if (st != GT_OK) return st;
        for (uint32_t k = 0; k < K; k++) {
            st = exchange_consume(d, p, k, s); if (st != GT_OK) return st;
            st = gt_program_combine_slice(p, k); if (st != GT_OK) return st;   // phase 1 of slice k (helper streams); phase 2 is left out
        }
        const bool more = p->iteration + 1 < iters;
        for (uint32_t k = 0; k < K; k++) {
            st = gt_program_phase2_part(p, k, iters, concurrent ? 1 : 0); if (st != GT_OK) return st;
            if (k + 1 == K) { st = tick(d, T_SPMV_DONE, s); if (st != GT_OK) return st; st = tick(d, T_APPLY_DONE, s); if (st != GT_OK) return st; }
            if (more) { st = exchange_issue_slice(d, p, s, k, d->t_iter + 1, concurrent ? p->part_done[k] : nullptr); if (st != GT_OK) return st; }
        }
        d->t_iter++;
        if (!more) break;
    }
    return GT_OK;
}

// Degree in _COL_ order: the column counts of all tile-rows are summed in the global column space
int all_reduce_y(gt_dist *d, gt_program *p, hipStream_t s) {
    if (d->loop) {
        LoopCtx &c = *d->loop;
        if (d->tmp_elems < p->y_elems) {
            if (d->tmp) GT_HIP(hipFree(d->tmp));
            d->tmp = nullptr; d->tmp_elems = 0;
            GT_HIP(hipMalloc((void **)&d->tmp, std::max<uint64_t>(p->y_elems, 1) * 4));
            d->tmp_elems = p->y_elems;
        }
        GT_HIP(hipStreamSynchronize(s));
        c.peer[d->rank].y = (const uint32_t *)p->y;
        GT_LOOP_BARRIER(c);
        GT_HIP(hipMemsetAsync(d->tmp, 0, p->y_elems * 4, s));
        for (int r = 0; r < c.n; r++) k_add_u32<<<1024, 256, 0, s>>>(d->tmp, c.peer[r].y, p->y_elems);
        GT_HIP(hipStreamSynchronize(s));
        GT_LOOP_BARRIER(c);                                          // everybody has read everybody's partial counts
        GT_HIP(hipMemcpyAsync(p->y, d->tmp, p->y_elems * 4, hipMemcpyDeviceToDevice, s));
        return GT_OK;
    }
    GT_NCCL(rccl()->AllReduce(p->y, p->y, p->y_elems, ncclUint32, ncclSum, d->comm, s));
    return GT_OK;
}

}  // namespace

// every rank sends bytes [send_off[q], send_off[q] + send_bytes[q]) of `send` to rank q and receives recv_bytes[q] bytes from it at
// recv_off[q] of `recv` (an all-to-all-v; the byte counts were agreed before: send_bytes[q] here = recv_bytes[me] on rank q)
int gt_dist_exchange_bytes(gt_dist *d, const void *send, const uint64_t *send_off, const uint64_t *send_bytes, void *recv, const uint64_t *recv_off,
                           const uint64_t *recv_bytes, hipStream_t s) {
    const int P = d->nranks;
    if (d->loop) {
        LoopCtx &c = *d->loop;
        struct Pub { const void *send; const uint64_t *off, *bytes; } mine{send, send_off, send_bytes};
        GT_HIP(hipStreamSynchronize(s));
        c.peer[d->rank].coll = &mine;
        GT_LOOP_BARRIER(c);
        for (int q = 0; q < P; q++) {
            const Pub *pq = (const Pub *)c.peer[q].coll;
            GT_REQUIRE(pq->bytes[d->rank] == recv_bytes[q], GT_ERR_STATE, "exchange of the build: rank %d sends %llu bytes to rank %d, which expects %llu", q,
                       (unsigned long long)pq->bytes[d->rank], d->rank, (unsigned long long)recv_bytes[q]);
            if (recv_bytes[q]) GT_HIP(hipMemcpyAsync((char *)recv + recv_off[q], (const char *)pq->send + pq->off[d->rank], recv_bytes[q], hipMemcpyDeviceToDevice, s));
        }
        GT_HIP(hipStreamSynchronize(s));
        GT_LOOP_BARRIER(c);
        return GT_OK;
    }
    // A rank's share for itself is a device copy. Between ranks: grouped rounds of at most 1 GiB per pair -- a single ncclSend of
    // the 8.6 GB that R-MAT-26 at world size 1 hands to itself came back as garbage (found by bench.py going through the
    // distributed build, round 4); chunk j of a pair is sent in the sender's round j and received in the receiver's round j, the
    // pairs match in order whatever the number of rounds either side needs for its other peers.
    GT_REQUIRE(send_bytes[d->rank] == recv_bytes[d->rank], GT_ERR_STATE, "exchange of the build: a rank's share for itself has two sizes");
    if (send_bytes[d->rank]) GT_HIP(hipMemcpyAsync((char *)recv + recv_off[d->rank], (const char *)send + send_off[d->rank], send_bytes[d->rank], hipMemcpyDeviceToDevice, s));
    const uint64_t CH = 1ull << 30;
    uint64_t rounds = 0;
    for (int q = 0; q < P; q++) if (q != d->rank) rounds = std::max(rounds, std::max((send_bytes[q] + CH - 1) / CH, (recv_bytes[q] + CH - 1) / CH));
    for (uint64_t j = 0; j < rounds; j++) {
        GT_NCCL(rccl()->GroupStart());
        for (int q = 0; q < P; q++) {
            if (q == d->rank) continue;
            const uint64_t so = j * CH, ro = j * CH;
            if (so < send_bytes[q]) GT_NCCL(rccl()->Send((const char *)send + send_off[q] + so, std::min(CH, send_bytes[q] - so), ncclUint8, q, d->comm, s));
            if (ro < recv_bytes[q]) GT_NCCL(rccl()->Recv((char *)recv + recv_off[q] + ro, std::min(CH, recv_bytes[q] - ro), ncclUint8, q, d->comm, s));
        }
        GT_NCCL(rccl()->GroupEnd());
    }
    return sync_deadline(d, s, "the record exchange of the distributed build");
}
// element-wise maximum over the ranks, in place (flags: a logical OR)
int gt_dist_all_reduce_max_u8(gt_dist *d, uint8_t *buf, uint64_t n, hipStream_t s) {
    if (d->loop) {
        LoopCtx &c = *d->loop;
        uint8_t *tmp = nullptr;
        GT_HIP(hipMalloc((void **)&tmp, n ? n : 1));
        GT_HIP(hipMemcpyAsync(tmp, buf, n, hipMemcpyDeviceToDevice, s));
        GT_HIP(hipStreamSynchronize(s));
        c.peer[d->rank].coll = tmp;   // the peers read the copy: everybody may overwrite its own buffer at once
        GT_LOOP_BARRIER(c);
        for (int q = 0; q < c.n; q++) if (q != d->rank) k_max_u8<<<1024, 256, 0, s>>>(buf, (const uint8_t *)c.peer[q].coll, n);
        GT_HIP(hipStreamSynchronize(s));
        GT_LOOP_BARRIER(c);
        (void)hipFree(tmp);
        return GT_OK;
    }
    GT_NCCL(rccl()->AllReduce(buf, buf, n, ncclUint8, ncclMax, d->comm, s));
    return sync_deadline(d, s, "the all-reduce of the column flags (distributed build)");
}
int gt_dist_all_reduce_sum_u64_host(gt_dist *d, uint64_t *v, uint32_t count) {
    for (uint32_t i = 0; i < count; i += 64) { int st = dist_all_reduce_words(d, v + i, std::min(64u, count - i), d->loop ? (hipStream_t)0 : d->comm_stream); if (st != GT_OK) return st; }
    return GT_OK;
}
int gt_dist_rank(const gt_dist *d) { return d->rank; }
int gt_dist_nranks(const gt_dist *d) { return d->nranks; }

extern "C" {

int gt_dist_unique_id(void *id_out) {
    GT_REQUIRE(id_out, GT_ERR_INVALID, "null argument");
    GT_REQUIRE(rccl(), GT_ERR_UNSUPPORTED, "librccl could not be loaded");
    ncclUniqueId id;
    GT_NCCL(rccl()->GetUniqueId(&id));
    static_assert(sizeof(id) == GT_DIST_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return GT_OK;
}

static int dist_init_common(gt_dist *d) {
    GT_HIP(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
    GT_HIP(hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming));
    GT_HIP(hipMalloc((void **)&d->d_word, 64 * 8));
    return GT_OK;
}

int gt_dist_create(gt_dist **out, const void *unique_id, int rank, int nranks) {
    GT_REQUIRE(out && unique_id && nranks >= 1 && rank >= 0 && rank < nranks, GT_ERR_INVALID, "gt_dist_create: bad arguments");
    GT_REQUIRE(rccl(), GT_ERR_UNSUPPORTED, "librccl could not be loaded");
    *out = nullptr;
    gt_dist *d = new gt_dist();
    d->rank = rank; d->nranks = nranks;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = rccl()->CommInitRank(&d->comm, nranks, id, rank);
    if (r != ncclSuccess) { gt_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, rccl()->GetErrorString(r)); delete d; return GT_ERR_HIP; }
    d->own_comm = true;
    int st = dist_init_common(d);
    if (st != GT_OK) { gt_dist_free(d); return st; }
    *out = d;
    return GT_OK;
}

int gt_dist_create_from_comm(gt_dist **out, void *nccl_comm, int rank, int nranks) {
    GT_REQUIRE(out && nccl_comm && nranks >= 1 && rank >= 0 && rank < nranks, GT_ERR_INVALID, "gt_dist_create_from_comm: bad arguments");
    GT_REQUIRE(rccl(), GT_ERR_UNSUPPORTED, "librccl could not be loaded");
    *out = nullptr;
    gt_dist *d = new gt_dist();
    d->rank = rank; d->nranks = nranks; d->comm = (ncclComm_t)nccl_comm; d->own_comm = false;
    int st = dist_init_common(d);
    if (st != GT_OK) { gt_dist_free(d); return st; }
    *out = d;
    return GT_OK;
}

int gt_dist_create_loopback(gt_dist **out, int nranks) {
    GT_REQUIRE(out && nranks >= 1, GT_ERR_INVALID, "gt_dist_create_loopback: bad arguments");
    auto ctx = std::make_shared<LoopCtx>();
    ctx->n = nranks; ctx->peer.resize(nranks);
    for (int r = 0; r < nranks; r++) {
        gt_dist *d = new gt_dist();
        d->rank = r; d->nranks = nranks; d->loop = ctx;
        out[r] = d;
    }
    return GT_OK;
}

int gt_dist_free(gt_dist *d) {
    if (!d) return GT_OK;
    if (d->comm && d->own_comm && rccl()) (void)rccl()->CommDestroy(d->comm);
    if (d->ev_ready) (void)hipEventDestroy(d->ev_ready);
    for (hipEvent_t e : d->ev_slice) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->ev_pack) (void)hipEventDestroy(e);
    if (d->comm_stream) (void)hipStreamDestroy(d->comm_stream);
    if (d->d_word) (void)hipFree(d->d_word);
    if (d->tmp) (void)hipFree(d->tmp);
    for (void *q : {(void *)d->d_stab, (void *)d->d_cnt, (void *)d->d_cur, (void *)d->d_smode, (void *)d->d_cx, (void *)d->pairs_send, (void *)d->pairs_recv, (void *)d->d_rtab,
                    (void *)d->d_ftab, (void *)d->d_rsel, (void *)d->d_flag, (void *)d->d_words, (void *)d->send_ptr, (void *)d->send_pos}) if (q) (void)hipFree(q);
    for (hipEvent_t e : d->tev) (void)hipEventDestroy(e);
    delete d;
    return GT_OK;
}

int gt_dist_exchange_stats(gt_dist *d, uint64_t *bytes_sent, uint64_t *bytes_dense, uint64_t *exchanges, int reset) {
    GT_REQUIRE(d, GT_ERR_INVALID, "null argument");
    if (bytes_sent) *bytes_sent = d->bytes_sent;
    if (bytes_dense) *bytes_dense = d->bytes_dense;
    if (exchanges) *exchanges = d->exchanges;
    if (reset) d->bytes_sent = d->bytes_dense = d->exchanges = 0;
    return GT_OK;
}

int gt_dist_iteration_times(gt_dist *d, double *out, uint32_t max_iterations, uint32_t *iterations) {
    GT_REQUIRE(d && iterations, GT_ERR_INVALID, "null argument");
    const uint32_t n = (uint32_t)(d->iter_ms.size() / GT_DIST_TIME_FIELDS);
    *iterations = n;
    if (out) memcpy(out, d->iter_ms.data(), (size_t)std::min(n, max_iterations) * GT_DIST_TIME_FIELDS * sizeof(double));
    return GT_OK;
}
int gt_dist_info(gt_dist *d, int32_t *transport_ranks, uint64_t *list_iterations, uint64_t *pair_spmspv_iterations, uint64_t *host_round_trips) {
    GT_REQUIRE(d, GT_ERR_INVALID, "null argument");
    if (transport_ranks) *transport_ranks = d->loop ? d->loop->n : d->rccl_ranks;
    if (list_iterations) *list_iterations = d->list_iters;
    if (pair_spmspv_iterations) *pair_spmspv_iterations = d->pair_spmspv_iters;
    if (host_round_trips) *host_round_trips = d->round_trips;
    return GT_OK;
}

int gt_dist_all_reduce_u64(gt_dist *d, uint64_t *host_values, uint32_t count) {
    GT_REQUIRE(d && host_values, GT_ERR_INVALID, "null argument");
    return dist_all_reduce_words(d, host_values, count, d->comm_stream);
}

// Vertex_Program::execute (vp:408-441) over the ranks of `d`: every rank calls it with its own program of the same kind.
static int dist_execute_impl(gt_dist *d, gt_program *p, uint32_t iters, gt_exec_stats *stats);
int gt_dist_execute(gt_dist *d, gt_program *p, uint32_t iters, gt_exec_stats *stats) {
    GT_REQUIRE(d && p, GT_ERR_INVALID, "null argument");
    if (d->loop && d->loop->aborted) { gt_set_error("rank %d: a peer of this loopback group failed in an earlier execute", d->rank); return GT_ERR_STATE; }
    const int st = dist_execute_impl(d, p, iters, stats);
    if (st != GT_OK && d->loop) d->loop->abort();   // the peers of a failed rank leave their barriers (with results that mean nothing) instead of hanging
    return st;
}
static int dist_execute_impl(gt_dist *d, gt_program *p, uint32_t iters, gt_exec_stats *stats) {
    const gt_graph *g = p->g;
    GT_REQUIRE(gt_has_exchange(g), GT_ERR_STATE, "gt_dist_execute needs a graph built with the exchange layout (nranks > 1, or GRAPHTAP_FORCE_EXCHANGE)");
    GT_REQUIRE((int)g->info.nranks == d->nranks && (int)g->info.rank == d->rank, GT_ERR_INVALID,
               "graph is tile-row %u of %u, the communicator says rank %d of %d", g->info.rank, g->info.nranks, d->rank, d->nranks);
    { int st = gt_program_prepare(p, iters); if (st != GT_OK) return st; }   // vp:410-413 + the message width of this run
    const bool check = p->check_sticky;
    const bool col = (p->prm.order == GT_COL);
    const uint32_t K = g->info.x_slices;
    hipStream_t s = p->stream;
    d->sp_graph = nullptr;   // block tables of the sparse exchange: rebuilt per call (a freed graph's address may come back)
    if (d->loop) { LoopCtx &c = *d->loop; c.peer[d->rank] = LoopPeer{(const char *)p->send, nullptr, g, p->x_bytes, 0}; GT_LOOP_BARRIER(c); }
    if (!col) { int st = plan_verify(d, g); if (st != GT_OK) return st; }
    (void)gt_program_enable_timing(p, stats != nullptr);
    d->timing = stats != nullptr; d->t_iter = 0; d->t_mode.clear();
    d->list_iters = d->pair_spmspv_iters = d->round_trips = 0;
    if (!d->loop && rccl()->CommCount && d->comm) (void)rccl()->CommCount(d->comm, &d->rccl_ranks);
    p->ev_used = 0; p->spmv_done = 0; p->spmspv_iters = 0; p->cf_filtered = 0; p->ev_acc_ms = 0; p->ev_acc_pairs = 0;
    k_dist_preload<<<1, 64, 0, s>>>();   // this file's code object is loaded at its first launch (milliseconds): not inside the timed loop
    GT_HIP(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    const char *penv = getenv("GRAPHTAP_DIST_PROTOCOL");   // "dense": the loop below for every program (A/B, tests)
    const char *sxenv = getenv("GRAPHTAP_SPARSE_EXCHANGE");   // "0": every block dense (the loop below)
    const bool lists = check && !col && !p->stationary && p->fl_enabled && !(penv && strcmp(penv, "dense") == 0) && !(sxenv && atoi(sxenv) == 0);
    // A fixed-count PageRank on K > 1 slices runs PIPELINED: phase 2 part by part, slice k of the next iteration's exchange issued
    // as soon as part k has written its messages (pipelined_execute). The decision is the same on every rank (program kind, K,
    // environment); GRAPHTAP_P2_PARTS=0 keeps the loop below.
    bool pipelined = false;
    if (!lists && !check && !col && p->stationary && K > 1 && iters > 0 && p->iteration < iters && !p->converged) {
        int st = gt_program_fuse_apply(p, iters, 0); if (st != GT_OK) return st;
        pipelined = gt_program_parts_begin(p);
        p->p2_by_parts = false;
        if (!pipelined) { p->fuse_armed = false; p->cf_hint = false; }
    }
    if (lists) { int st = lists_execute(d, p, stats, t0); if (st != GT_OK) { d->lists_protocol = false; d->sparse_now = false; return st; } }
    else if (pipelined) { int st = pipelined_execute(d, p, iters); p->pack_deferred = false; p->p2_by_parts = false; if (st != GT_OK) return st; }
    else for (;;) {
        int st = tick(d, T_START, s); if (st != GT_OK) return st;
        p->pack_deferred = !d->loop && !col && p->stationary;   // stationary programs send dense blocks: every slice is packed on the communication stream, right in front of its sends
        st = gt_program_scatter_gather(p); if (st != GT_OK) { p->pack_deferred = false; return st; }
        if (!col) { st = gt_program_fuse_apply(p, iters, check ? 1 : 0); if (st != GT_OK) return st; p->pr_state = gt_pr_state_mode(p, iters, check); }
        st = tick(d, T_SEND_READY, s); if (st != GT_OK) return st;
        if (!col && !p->converged) {
            st = exchange_issue(d, p, s); p->pack_deferred = false; if (st != GT_OK) return st;
            for (uint32_t k = 0; k < K; k++) {
                st = exchange_consume(d, p, k, s); if (st != GT_OK) return st;
                st = (K > 1) ? gt_program_combine_slice(p, k) : gt_program_combine(p); if (st != GT_OK) return st;
            }
        } else {
            p->pack_deferred = false;
            st = gt_program_combine(p); if (st != GT_OK) return st;
        }
        if (col && !p->converged) { st = all_reduce_y(d, p, s); if (st != GT_OK) return st; }
        st = tick(d, T_SPMV_DONE, s); if (st != GT_OK) return st;
        uint64_t active = 0;
        st = gt_program_apply(p, iters, check ? &active : nullptr); if (st != GT_OK) return st;
        st = tick(d, T_APPLY_DONE, s); if (st != GT_OK) return st;
        d->t_iter++;
        if (check) {
            d->round_trips += 2;   // the count read by apply, then the all-reduce (+ the sparse exchange's own, sparse_prepare)
            st = dist_all_reduce_words(d, &active, 1, s); if (st != GT_OK) return st;       // has_converged, vp:1918
            p->last_active = active;
            if (active == 0) { st = gt_program_finish_converged(p); if (st != GT_OK) return st; break; }
        } else if (p->iteration >= iters) break;
    }
    { int st = sync_deadline(d, s, "the iteration loop"); if (st != GT_OK) return st; }
    if (d->comm_stream) { int st = sync_deadline(d, d->comm_stream, "the exchange"); if (st != GT_OK) return st; }
    { int st = collect_times(d, p->iteration, !d->loop && !col); if (st != GT_OK) return st; }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->iterations = p->iteration; stats->converged = p->converged;
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        double ms = 0; uint32_t n = 0;
        (void)gt_program_timing(p, &ms, &n, 1);
        stats->spmv_ms = ms; stats->spmv_launches = n; stats->spmspv_iterations = p->spmspv_iters; stats->cf_filtered_iterations = p->cf_filtered;
        stats->list_iterations = lists ? (uint32_t)d->list_iters : 0;
    }
    return GT_OK;
}

}  // extern "C"
