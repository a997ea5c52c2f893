// ingest.hip -- device-side graph build: edge records -> owned tile-row in TCSC form.
//
// Replaces, for this rank's tile-row, the reference's host pipeline
//   Graph::parread_binary      src/mat/graph.hpp:308-372   (per-record flags)
//   Matrix::init_tiles         src/mat/matrix.hpp:538-560  (column-major sort, dedupe)
//   Matrix::init_filtering     src/mat/matrix.hpp:813-858, 861-1144 (I/IV/J/JV, classes)
//   TCSC_BASE::populate        src/ds/compressed_column.hpp:371-417
// with radix sorts / scans (rocPRIM through hipCUB) and a few streaming kernels.
// Every rank reads the full edge list (replicated read) and keeps -- compacted, before any sort -- tile-row `rank`.
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <vector>

#include "gt_internal.h"

namespace {

constexpr int TPB = 256;

inline unsigned grid_for(uint64_t n, int per_thread = 1) {
    uint64_t b = (n + (uint64_t)TPB * per_thread - 1) / ((uint64_t)TPB * per_thread);
    if (b < 1) b = 1;
    if (b > 256u * 32u) b = 256u * 32u;  // grid-stride the rest
    return (unsigned)b;
}

struct DevBuf {  // frees on scope exit: ingest scratch
    void *p = nullptr;
    ~DevBuf() { if (p) gt_scratch_free(p); }
    int alloc(uint64_t bytes) { return gt_scratch_malloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
    template <class T> T *as() { return (T *)p; }
};

// ---- pass 1: per-record flags, global non-empty bitmaps, localisation -------------------
// Order of the flag handling is the reference's (graph.hpp:337-356): drop self loop unless
// self_loops; acyclic swap; transpose swap; insert; mirrored insert when !directed.
constexpr int EXPAND_TPB = 1024;
__global__ void __launch_bounds__(EXPAND_TPB) k_expand(const uint32_t *__restrict__ rec, uint64_t m, int stride, gt_graph_flags f,
                         uint32_t nrows, uint32_t perm_a, uint32_t perm_mask, uint32_t H, uint32_t row_lo, uint32_t row_hi,
                         uint64_t *__restrict__ keys, uint32_t *__restrict__ wts, uint64_t room /* capacity of keys / wts */,
                         uint8_t *__restrict__ rowflag, uint8_t *__restrict__ colflag,
                         uint32_t *__restrict__ needme /* [span] entries this tile-row has in every column, or null on one rank */,
                         uint32_t *__restrict__ needby /* [nranks][H] entries tile-row d has in every owned column, or null */,
                         unsigned long long *__restrict__ counters /* [0]=kept [1]=out-of-range [2]=global entries */,
                         bool shuffled /* the records are this rank's after the shuffle: [2] counts the kept entries only (the ranks add up) */) {
    unsigned long long bad = 0, glob = 0;
    const uint32_t lane = threadIdx.x & 63;
    // non-empty ROWS are only needed for the owned segment (compressed row ids are segment-local); the non-empty COLUMNS of
    // every segment are needed everywhere (the exchange layout numbers a source segment's columns by compressed id)
    const bool all_rows = (needme == nullptr);   // one rank without the exchange layout
    auto below = [](uint64_t mk) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0)); };
    // COMPACTING: only the entries of this rank's tile-row are written (a rank of p sorts ~m/p keys, not m). A WORKGROUP
    // reserves room for what it keeps with one atomic per round (one per wave was 16.7 M atomics on one address for R-MAT-26:
    // 203 ms of kernel time, 26.5 ms with one per 1024 records -- profiles/r02_hubs_first/tilerow_ingest_kernels.txt); the order of the output is therefore arbitrary, which the sort that follows
    // does not care about (equal keys -- and, with weights, equal (key, weight) pairs -- are interchangeable).
    __shared__ uint32_t wave_kept[EXPAND_TPB / 64];
    __shared__ unsigned long long round_base;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t m64 = (m + EXPAND_TPB - 1) / EXPAND_TPB * EXPAND_TPB;   // every thread of a workgroup makes the same number of rounds
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < m64; e += (uint64_t)gridDim.x * blockDim.x) {
        bool keep0 = false, keep1 = false;
        uint64_t k0 = 0, k1 = 0;
        uint32_t w = 0;
        if (e < m) {
            uint32_t row = rec[e * stride], col = rec[e * stride + 1];
            w = (stride == 3) ? rec[e * stride + 2] : 0;
            if (row >= nrows || col >= nrows) {
                bad++;
            } else if (!(row == col && !f.self_loops)) {
                if (f.acyclic && col < row) { uint32_t t = row; row = col; col = t; }
                if (f.transpose) { uint32_t t = row; row = col; col = t; }
                row = (row * perm_a) & perm_mask; col = (col * perm_a) & perm_mask;   // internal ids (identity on one rank)
                // test before set: a vertex has ~16 records on average, so 15 of 16 visits find the flag set and a cached read
                // replaces a scattered one-byte write (a read-modify-write of a sector at the memory side)
                auto set = [](uint8_t *__restrict__ flags, uint64_t i) { if (!flags[i]) flags[i] = 1; };
                if (all_rows || (row >= row_lo && row < row_hi)) set(rowflag, row);
                set(colflag, col); glob += (!shuffled || (row >= row_lo && row < row_hi));
                // counts, not flags: both sides of an exchange block order its columns by these (descending), so the receiver's
                // count of column c and the owner's count for that receiver must come from the same records -- they do
                if (row >= row_lo && row < row_hi) { k0 = ((uint64_t)col << 32) | row; keep0 = true; if (needme) atomicAdd(&needme[col], 1u); }
                if (needby && col >= row_lo && col < row_hi) atomicAdd(&needby[(uint64_t)(row / H) * H + (col - row_lo)], 1u);
                if (!f.directed) {
                    if (all_rows || (col >= row_lo && col < row_hi)) set(rowflag, col);
                    set(colflag, row); glob += (!shuffled || (col >= row_lo && col < row_hi));
                    if (col >= row_lo && col < row_hi) { k1 = ((uint64_t)row << 32) | col; keep1 = true; if (needme) atomicAdd(&needme[row], 1u); }
                    if (needby && row >= row_lo && row < row_hi) atomicAdd(&needby[(uint64_t)(col / H) * H + (row - row_lo)], 1u);
                }
            }
        }
        const uint64_t b0 = __ballot(keep0), b1 = __ballot(keep1);
        const uint32_t n0 = (uint32_t)__popcll((unsigned long long)b0), n1 = (uint32_t)__popcll((unsigned long long)b1);
        if (lane == 0) wave_kept[wave] = n0 + n1;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (uint32_t v = 0; v < EXPAND_TPB / 64; v++) { const uint32_t c = wave_kept[v]; wave_kept[v] = total; total += c; }
            round_base = total ? atomicAdd(&counters[0], (unsigned long long)total) : 0ull;
        }
        __syncthreads();
        if (n0 + n1) {
            const unsigned long long base = round_base + wave_kept[wave];
            if (keep0) { const uint64_t o = base + below(b0); if (o < room) { keys[o] = k0; if (wts) wts[o] = w; } }
            if (keep1) { const uint64_t o = base + n0 + below(b1); if (o < room) { keys[o] = k1; if (wts) wts[o] = w; } }
        }
        __syncthreads();   // wave_kept / round_base are rewritten by the next round
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_down(bad, o); glob += __shfl_down(glob, o); }
    if (lane == 0) {
        if (bad) atomicAdd(&counters[1], bad);
        if (glob) atomicAdd(&counters[2], glob);
    }
}

// ---- distributed build: where a record has to go (Matrix::distribute, mat/matrix.hpp:693-810). A record becomes one stored entry
// (row, col) -- two for an undirected graph -- after the flag handling of k_expand; it travels to the owner(s) of the entries' rows
// (owner = internal row id / H). Records k_expand would drop or reject stay with this rank (it reports the bad ones).
__device__ __forceinline__ void record_owners(const uint32_t *__restrict__ rec, uint64_t e, int stride, gt_graph_flags f, uint32_t nrows, uint32_t perm_a,
                                              uint32_t perm_mask, uint32_t H, uint32_t me, uint32_t &d0, uint32_t &d1) {
    uint32_t row = rec[e * stride], col = rec[e * stride + 1];
    d0 = me; d1 = 0xFFFFFFFFu;
    if (row >= nrows || col >= nrows) return;            // out of range: counted as bad by my own k_expand
    if (row == col && !f.self_loops) return;             // dropped anyway
    if (f.acyclic && col < row) { uint32_t t = row; row = col; col = t; }
    if (f.transpose) { uint32_t t = row; row = col; col = t; }
    row = (row * perm_a) & perm_mask; col = (col * perm_a) & perm_mask;
    d0 = row / H;
    if (!f.directed) { const uint32_t o = col / H; if (o != d0) d1 = o; }
}
__global__ void __launch_bounds__(256) k_route_count(const uint32_t *__restrict__ rec, uint64_t m, int stride, gt_graph_flags f, uint32_t nrows, uint32_t perm_a,
                                                     uint32_t perm_mask, uint32_t H, uint32_t me, uint32_t p, unsigned long long *__restrict__ counts) {
    extern __shared__ unsigned int hist[];
    for (uint32_t i = threadIdx.x; i < p; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < m; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t d0, d1;
        record_owners(rec, e, stride, f, nrows, perm_a, perm_mask, H, me, d0, d1);
        atomicAdd(&hist[d0], 1u);
        if (d1 != 0xFFFFFFFFu) atomicAdd(&hist[d1], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < p; i += blockDim.x) if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}
// records into the send buffer, grouped by destination: a workgroup reserves room per destination once per round
__global__ void __launch_bounds__(256) k_route_fill(const uint32_t *__restrict__ rec, uint64_t m, int stride, gt_graph_flags f, uint32_t nrows, uint32_t perm_a,
                                                    uint32_t perm_mask, uint32_t H, uint32_t me, uint32_t p, const unsigned long long *__restrict__ base,
                                                    unsigned long long *__restrict__ cursor, uint32_t *__restrict__ out) {
    extern __shared__ unsigned int sh[];   // [p] this round's counts, then the workgroup's slots; [p] 64-bit bases behind them
    unsigned int *cnt = sh;
    unsigned long long *wbase = reinterpret_cast<unsigned long long *>(sh + ((p + 1) & ~1u));
    const uint64_t m_round = (m + blockDim.x - 1) / blockDim.x * blockDim.x;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < m_round; e += (uint64_t)gridDim.x * blockDim.x) {
        for (uint32_t i = threadIdx.x; i < p; i += blockDim.x) cnt[i] = 0;
        __syncthreads();
        uint32_t d0 = 0xFFFFFFFFu, d1 = 0xFFFFFFFFu, s0 = 0, s1 = 0;
        if (e < m) {
            record_owners(rec, e, stride, f, nrows, perm_a, perm_mask, H, me, d0, d1);
            s0 = atomicAdd(&cnt[d0], 1u);
            if (d1 != 0xFFFFFFFFu) s1 = atomicAdd(&cnt[d1], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < p; i += blockDim.x) wbase[i] = cnt[i] ? atomicAdd(&cursor[i], (unsigned long long)cnt[i]) : 0ull;
        __syncthreads();
        if (e < m) {
            for (int w = 0; w < 2; w++) {
                const uint32_t dd = w ? d1 : d0;
                if (dd == 0xFFFFFFFFu) continue;
                const unsigned long long o = base[dd] + wbase[dd] + (w ? s1 : s0);
                for (int j = 0; j < stride; j++) out[o * stride + j] = rec[e * stride + j];
            }
        }
        __syncthreads();
    }
}
// needby of destination q = what q's tile-row has in MY columns = q's needme over my segment

struct U8ToU32 {
    __host__ __device__ uint32_t operator()(const uint8_t &v) const { return v; }
};
struct NonZero {
    __host__ __device__ uint32_t operator()(const uint32_t &v) const { return v ? 1u : 0u; }
};

// ---- several ranks: the column space of a tile-row is LOCAL. Rank r keeps only the columns its tile-row has an
// entry in ("needed" columns, ~47 % of all non-empty columns at p = 8 on R-MAT-26), ordered [slice k][source segment s]
// [ascending compressed column j]; slice of a column = j / T. Block (k, s) is what rank s sends to rank r in the k-th
// all-to-all of an iteration, so the receive buffer of that collective IS slice k of the SpMV's message vector.
// INSIDE a block the columns are ordered by the number of entries the receiving tile-row has in them, descending (ties:
// ascending column): the phase-1 windows at the head of every block then hold its hub columns, whose same-row entries
// pre-aggregate several times better (pb.hip) -- the hubs-first layout of the single-rank build, block by block, at no
// cost at run time because the sender packs its buffer through an index list anyway (k_pack_send). `Pneed[col]` is that
// position; the owner computes the same order from the same counts (k_send_list, Pby).
struct BlockTab { uint32_t base; };   // first local column id of block (k,s)

__device__ __forceinline__ uint32_t local_col(const BlockTab *__restrict__ tab, const uint32_t *__restrict__ Scol,
                                              const uint32_t *__restrict__ Pneed, uint32_t H, uint32_t T, uint32_t p, uint32_t col) {
    const uint32_t seg = col / H, j = Scol[col] - Scol[seg * H], k = j / T;
    return tab[k * p + seg].base + Pneed[col];
}

// sort keys of the needed columns / of the (destination, owned column) pairs: (block, entries descending); the radix sort is
// stable and the items arrive in ascending column order
__global__ void k_need_keys(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ Sneed, uint64_t span, const uint32_t *__restrict__ Scol,
                            uint32_t H, uint32_t T, uint32_t p, uint32_t hub_min, uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
    for (uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; c < span; c += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t n = cnt[c];
        if (!n) continue;
        if (n < hub_min) n = 0;   // the tail of a block: one class, in column order (exchange_hub_min below)
        const uint32_t seg = (uint32_t)(c / H), j = Scol[c] - Scol[(uint64_t)seg * H], k = j / T;
        const uint32_t o = Sneed[c];
        key[o] = ((uint64_t)(k * p + seg) << 32) | (0xFFFFFFFFu - n);
        val[o] = (uint32_t)c;
    }
}
__global__ void k_by_keys(const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ Sby, uint32_t H, uint32_t T, uint32_t p,
                          const uint32_t *__restrict__ Scol, uint32_t col_lo, uint32_t hub_min, uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
    const uint64_t n_all = (uint64_t)p * H;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n_all; t += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t n = cnt[t];
        if (!n) continue;
        if (n < hub_min) n = 0;
        const uint32_t d = (uint32_t)(t / H), i = (uint32_t)(t - (uint64_t)d * H);
        const uint32_t j = Scol[col_lo + i] - Scol[col_lo], k = j / T;
        const uint32_t o = Sby[t];
        key[o] = ((uint64_t)(k * p + d) << 32) | (0xFFFFFFFFu - n);
        val[o] = (uint32_t)t;
    }
}
// position of every sorted item inside its block: sorted index - first sorted index of the block
__global__ void k_block_positions(const uint64_t *__restrict__ key, const uint32_t *__restrict__ val, uint32_t n,
                                  const uint32_t *__restrict__ block_start, uint32_t *__restrict__ P) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        P[val[i]] = i - block_start[(uint32_t)(key[i] >> 32)];
}

// Done BEFORE the sort so that the entries come out ordered by the id the kernels use.
__global__ void k_remap_cols(uint64_t *__restrict__ keys, uint64_t n, const uint32_t *__restrict__ Scol, const uint32_t *__restrict__ Pneed,
                             const BlockTab *__restrict__ tab, uint32_t H, uint32_t T, uint32_t p) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        const uint32_t col = (uint32_t)(k >> 32);
        const uint32_t c = tab ? local_col(tab, Scol, Pneed, H, T, p, col) : Scol[col];   // one rank: the compressed column id
        keys[i] = ((uint64_t)c << 32) | (uint32_t)k;
    }
}

// first internal column id of slice k of segment s: lower bound of Scol[sH] + k*T in Scol[sH .. (s+1)H]
__global__ void k_slice_bounds(const uint32_t *__restrict__ Scol, uint32_t H, uint32_t T, uint32_t p, uint32_t K, uint32_t *__restrict__ clo) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (K + 1) * p) return;
    const uint32_t k = t / p, s = t % p;
    uint64_t lo = (uint64_t)s * H, hi = (uint64_t)(s + 1) * H;
    if (k < K) {
        const uint64_t target = (uint64_t)Scol[lo] + (uint64_t)k * T;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (Scol[mid] < target) lo = mid + 1; else hi = mid; }
    } else lo = hi;
    clo[t] = (uint32_t)lo;
}
__global__ void k_gather_u32(const uint32_t *__restrict__ src, const uint64_t *__restrict__ idx, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = src[idx[t]];
}
// local column -> slot of the global [segment][seg_stride] column space (Degree in _COL_ order all-reduces there)
__global__ void k_local_to_global(const uint32_t *__restrict__ needme, uint64_t span, const uint32_t *__restrict__ Scol,
                                  const uint32_t *__restrict__ Pneed, const BlockTab *__restrict__ tab, uint32_t H, uint32_t T, uint32_t p,
                                  uint32_t S, uint32_t *__restrict__ loc2glob) {
    for (uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; c < span; c += (uint64_t)gridDim.x * blockDim.x) {
        if (!needme[c]) continue;
        const uint32_t seg = (uint32_t)(c / H);
        loc2glob[local_col(tab, Scol, Pneed, H, T, p, (uint32_t)c)] = seg * S + (Scol[c] - Scol[(uint64_t)seg * H]);
    }
}
// send list: element i of the send buffer is the message of owned compressed column send_idx[i]
struct SendTab { uint32_t base; };   // first send-buffer element of block (k,d)
__global__ void k_send_list(const uint32_t *__restrict__ needby, const uint32_t *__restrict__ Pby, const uint32_t *__restrict__ Scol,
                            const SendTab *__restrict__ tab, uint32_t H, uint32_t T, uint32_t p, uint32_t col_lo, uint32_t *__restrict__ send_idx) {
    const uint64_t n = (uint64_t)p * H;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n; t += (uint64_t)gridDim.x * blockDim.x) {
        if (!needby[t]) continue;
        const uint32_t d = (uint32_t)(t / H), i = (uint32_t)(t - (uint64_t)d * H);
        const uint32_t j = Scol[col_lo + i] - Scol[col_lo], k = j / T;
        const SendTab b = tab[k * p + d];
        send_idx[b.base + Pby[t]] = j;
    }
}

// ---- dedupe: head flags over the sorted keys (matrix.hpp:545, 553: same row and col) ----
__global__ void k_head_flags(const uint64_t *__restrict__ keys, uint64_t n, int dedupe, uint32_t *__restrict__ head) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        head[i] = (!dedupe || i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// ---- populate: compressed ids of every kept entry (compressed_column.hpp:382-394) -------
__global__ void k_populate(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ wts, uint64_t n,
                           const uint32_t *__restrict__ head, const uint32_t *__restrict__ pos,
                           const uint32_t *__restrict__ Srow, uint32_t row_lo,
                           uint32_t *__restrict__ IA, uint32_t *__restrict__ JI, uint32_t *__restrict__ A) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        uint32_t o = pos[i];
        const uint32_t row = (uint32_t)keys[i];
        IA[o] = Srow[row] - Srow[row_lo];
        JI[o] = (uint32_t)(keys[i] >> 32);   // already the message-vector slot (k_remap_cols)
        if (A) A[o] = wts[i];
    }
}

// JA[c] = first entry whose column id is >= c (entries are sorted by column id)
__global__ void k_col_ptr(const uint32_t *__restrict__ JI, uint32_t nnz, uint32_t ncols, uint32_t *__restrict__ JA) {
    for (uint64_t c = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; c <= ncols; c += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nnz;
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if (JI[mid] < (uint32_t)c) lo = mid + 1; else hi = mid;
        }
        JA[c] = lo;
    }
}

// ---- owned-segment filter vectors (matrix.hpp:822-849, 1125-1144) -----------------------
__global__ void k_segment_vectors(const uint8_t *__restrict__ rowflag, const uint8_t *__restrict__ colflag,
                                  const uint32_t *__restrict__ Srow, const uint32_t *__restrict__ Scol,
                                  uint32_t H, uint32_t lo, uint8_t *__restrict__ IJ, uint32_t *__restrict__ IV,
                                  uint32_t *__restrict__ JV, uint32_t *__restrict__ IR, uint32_t *__restrict__ JC,
                                  uint32_t *__restrict__ R2C, unsigned int *__restrict__ classes) {
    unsigned reg = 0, src = 0, snk = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < H; i += gridDim.x * blockDim.x) {
        uint32_t v = lo + i;
        uint8_t r = rowflag[v], c = colflag[v];
        uint32_t iv = Srow[v] - Srow[lo], jv = Scol[v] - Scol[lo];
        IJ[i] = (uint8_t)(r | (c << 1)); IV[i] = iv; JV[i] = jv;
        if (r) { IR[iv] = i; R2C[iv] = c ? jv : 0xFFFFFFFFu; }
        if (c) JC[jv] = i;
        reg += (r && c); src += (r && !c); snk += (!r && c);
    }
    for (int o = 32; o > 0; o >>= 1) { reg += __shfl_down(reg, o); src += __shfl_down(src, o); snk += __shfl_down(snk, o); }
    if ((threadIdx.x & 63) == 0) {
        if (reg) atomicAdd(&classes[0], reg);
        if (src) atomicAdd(&classes[1], src);
        if (snk) atomicAdd(&classes[2], snk);
    }
}

}  // namespace

#define ING_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            gt_set_error("ingest: %s failed: %s (line %d)", #call, hipGetErrorString(e_), __LINE__); \
            return GT_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)
#define ING_ALLOC(buf, bytes)                                                                      \
    do {                                                                                           \
        if ((buf).alloc(bytes)) { gt_set_error("ingest: out of device memory (%llu bytes)", (unsigned long long)(bytes)); return GT_ERR_HIP; } \
    } while (0)

// Distributed build: a rank that fails ALONE (an out-of-range record in its share, an allocation that only it could not get)
// must not leave its peers inside the next collective until the deadline: before every collective stage the ranks sum a
// status word and return the error TOGETHER. `local` = this rank's status of the stage just finished; the failing rank keeps
// its own message, the others say who failed. A null `dist` is the replicated build (every rank sees the same records and
// fails the same way): nothing to agree on.
static int ing_agree(gt_dist *dist, int local, const char *stage) {
    if (!dist) return local;
    const int P = gt_dist_nranks(dist), me = gt_dist_rank(dist);
    std::vector<uint64_t> w((size_t)P, 0);
    if (local != GT_OK) w[(size_t)me] = (uint64_t)(uint32_t)(-local);
    { int st = gt_dist_all_reduce_sum_u64_host(dist, w.data(), (uint32_t)P); if (st != GT_OK) return local != GT_OK ? local : st; }
    if (local != GT_OK) return local;
    for (int q = 0; q < P; q++)
        if (w[(size_t)q]) { gt_set_error("distributed build: rank %d failed while %s (status %d); every rank returns", q, stage, -(int)w[(size_t)q]); return -(int)w[(size_t)q]; }
    return GT_OK;
}

int gt_ingest(gt_graph *g, const void *edges_dev, uint64_t m, int weighted, gt_dist *dist) {
    const gt_graph_flags f = g->flags;
    const uint32_t p = g->info.nranks, k = g->info.rank, H = g->info.tile_height, nrows = g->info.nrows;
    const uint64_t span = (uint64_t)p * H;             // vertex slots of the whole grid (>= nrows)
    const uint32_t row_lo = k * H;
    const uint32_t row_hi = (uint32_t)(((uint64_t)(k + 1) * H < g->nint) ? (uint64_t)(k + 1) * H : g->nint);
    const int stride = weighted ? 3 : 2;
    const int slots = f.directed ? 1 : 2;
    hipStream_t s = 0;
    // the exchange layout (local column space, send list, K slices): always on several ranks; on ONE rank only when
    // asked for (GRAPHTAP_FORCE_EXCHANGE), so that the whole multi-rank driver path -- RCCL self-exchange included --
    // runs on a one-GPU box
    const bool multi = p > 1 || g->force_exchange;
    // ---- distributed build: this rank holds a SHARE of the records; they go to the owners of their rows first
    DevBuf shuffled_buf;
    if (dist) {
        GT_REQUIRE(multi, GT_ERR_STATE, "the distributed build needs the exchange layout");
        GT_REQUIRE(gt_dist_nranks(dist) == (int)p && gt_dist_rank(dist) == (int)k, GT_ERR_INVALID, "the communicator and the graph disagree on rank / nranks");
        DevBuf cnt_d, base_d, cur_d, sendbuf;
        std::vector<uint64_t> scount(p), soff(p), rcount(p), roff(p), matrix((size_t)p * p, 0);
        const unsigned rgrid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((m + 255) / 256, 256u * 16u));
        uint64_t stotal = 0, rtotal = 0;
        const uint64_t rb = (uint64_t)stride * 4;
        // every rank-local stage is a lambda whose status the ranks agree on before the collective that follows it (ing_agree)
        int local = [&]() -> int {
            ING_ALLOC(cnt_d, p * 8); ING_ALLOC(base_d, p * 8); ING_ALLOC(cur_d, p * 8);
            ING_HIP(hipMemsetAsync(cnt_d.p, 0, p * 8, s)); ING_HIP(hipMemsetAsync(cur_d.p, 0, p * 8, s));
            if (m) k_route_count<<<rgrid, 256, p * 4, s>>>((const uint32_t *)edges_dev, m, stride, f, nrows, g->perm_a, g->perm_mask, H, k, p, cnt_d.as<unsigned long long>());
            ING_HIP(hipMemcpyAsync(scount.data(), cnt_d.p, p * 8, hipMemcpyDeviceToHost, s));
            ING_HIP(hipStreamSynchronize(s));
            return GT_OK;
        }();
        { int st = ing_agree(dist, local, "counting where its records go"); if (st != GT_OK) return st; }
        for (uint32_t q = 0; q < p; q++) { soff[q] = stotal; stotal += scount[q]; matrix[(size_t)k * p + q] = scount[q]; }
        { int st = gt_dist_all_reduce_sum_u64_host(dist, matrix.data(), p * p); if (st != GT_OK) return st; }   // everybody learns every count
        for (uint32_t q = 0; q < p; q++) { rcount[q] = matrix[(size_t)q * p + k]; roff[q] = rtotal; rtotal += rcount[q]; }
        local = [&]() -> int {
            ING_ALLOC(sendbuf, std::max<uint64_t>(stotal, 1) * rb); ING_ALLOC(shuffled_buf, std::max<uint64_t>(rtotal, 1) * rb);
            ING_HIP(hipMemcpyAsync(base_d.p, soff.data(), p * 8, hipMemcpyHostToDevice, s));
            if (m) k_route_fill<<<rgrid, 256, ((p + 1) & ~1u) * 4 + p * 8, s>>>((const uint32_t *)edges_dev, m, stride, f, nrows, g->perm_a, g->perm_mask, H, k, p,
                                                                                base_d.as<unsigned long long>(), cur_d.as<unsigned long long>(), sendbuf.as<uint32_t>());
            ING_HIP(hipGetLastError());
            return GT_OK;
        }();
        { int st = ing_agree(dist, local, "packing its records for the shuffle"); if (st != GT_OK) return st; }
        std::vector<uint64_t> sob(p), scb(p), rob(p), rcb(p);
        for (uint32_t q = 0; q < p; q++) { sob[q] = soff[q] * rb; scb[q] = scount[q] * rb; rob[q] = roff[q] * rb; rcb[q] = rcount[q] * rb; }
        { int st = gt_dist_exchange_bytes(dist, sendbuf.p, sob.data(), scb.data(), shuffled_buf.p, rob.data(), rcb.data(), s); if (st != GT_OK) return st; }
        edges_dev = shuffled_buf.p; m = rtotal;
    }
    const uint64_t cap = m * slots;
    DevBuf keys, keys2, wts, wts2, rowflag, colflag, Srow, Scol, counters, tmp, needme, needby, Sneed, Sby, Pneed, Pby;
    uint64_t room = 0;
    unsigned long long hc[3] = {0, 0, 0};
    // (a lambda: on several ranks with a distributed build its status is agreed on before the collectives that follow)
    int expand_status = [&]() -> int {

        // Room for the kept entries: everything on one rank; on several, the hashed id space spreads the entries evenly, so
        // 1.25 x the mean share (+ slack for small graphs) is reserved and the pass is repeated with the exact count in the
        // rare case it does not fit (a multi-GB hipMalloc is not free: 8.6 GB for every rank of 8 at R-MAT-26 otherwise).
        room = (p == 1 || dist) ? cap : std::min<uint64_t>(cap, cap / p + cap / (4 * p) + 65536);   // (after the shuffle every record is mine)
        ING_ALLOC(keys, room * 8);
        if (weighted) ING_ALLOC(wts, room * 4);
        ING_ALLOC(rowflag, span + 1); ING_ALLOC(colflag, span + 1);
        ING_ALLOC(Srow, (span + 1) * 4); ING_ALLOC(Scol, (span + 1) * 4);
        ING_ALLOC(counters, 8 * sizeof(unsigned long long));
        ING_HIP(hipMemsetAsync(rowflag.p, 0, span + 1, s));
        ING_HIP(hipMemsetAsync(colflag.p, 0, span + 1, s));
        ING_HIP(hipMemsetAsync(counters.p, 0, 8 * sizeof(unsigned long long), s));
        if (multi) {
            ING_ALLOC(needme, (span + 1) * 4); ING_ALLOC(needby, (span + 1) * 4);
            ING_ALLOC(Sneed, (span + 1) * 4); ING_ALLOC(Sby, (span + 1) * 4);
            ING_ALLOC(Pneed, (span + 1) * 4); ING_ALLOC(Pby, (span + 1) * 4);
            ING_HIP(hipMemsetAsync(needme.p, 0, (span + 1) * 4, s));
            ING_HIP(hipMemsetAsync(needby.p, 0, (span + 1) * 4, s));
        }

        for (int attempt = 0; attempt < 2 && m; attempt++) {
            k_expand<<<(unsigned)std::min<uint64_t>((m + EXPAND_TPB - 1) / EXPAND_TPB, 256u * 8u), EXPAND_TPB, 0, s>>>((const uint32_t *)edges_dev, m, stride, f, nrows, g->perm_a, g->perm_mask, H, row_lo, row_hi,
                                                 keys.as<uint64_t>(), weighted ? wts.as<uint32_t>() : nullptr, room,
                                                 rowflag.as<uint8_t>(), colflag.as<uint8_t>(),
                                                 multi ? needme.as<uint32_t>() : nullptr, (multi && !dist) ? needby.as<uint32_t>() : nullptr,
                                                 counters.as<unsigned long long>(), dist != nullptr);
            ING_HIP(hipMemcpyAsync(hc, counters.p, sizeof(hc), hipMemcpyDeviceToHost, s));
            ING_HIP(hipStreamSynchronize(s));
            if (hc[0] <= room) break;
            // did not fit (nothing was written past `room`): exact size, second pass (flags are idempotent, the counts start over)
            if (multi) { ING_HIP(hipMemsetAsync(needme.p, 0, (span + 1) * 4, s)); ING_HIP(hipMemsetAsync(needby.p, 0, (span + 1) * 4, s)); }
            room = hc[0];
            gt_scratch_free(keys.p); keys.p = nullptr; ING_ALLOC(keys, room * 8);
            if (weighted) { gt_scratch_free(wts.p); wts.p = nullptr; ING_ALLOC(wts, room * 4); }
            ING_HIP(hipMemsetAsync(counters.p, 0, 8 * sizeof(unsigned long long), s));
        }
        if (hc[1]) {
            gt_set_error("%llu edge record(s) name a vertex id > num_vertices=%u (the reference overflows its tile grid silently, "
                         "src/mat/matrix.hpp:218-220; this library rejects the input)", hc[1], g->info.num_vertices);
            return GT_ERR_INVALID;
        }
        GT_REQUIRE(hc[0] < 0xFFFFFFFFull, GT_ERR_UNSUPPORTED,
                   "tile-row holds %llu entries; column pointers are 32-bit like the reference's Integer_Type "
                   "(src/ds/compressed_column.hpp:294): use more ranks", (unsigned long long)hc[0]);
        return GT_OK;
    }();
    { int st = ing_agree(dist, expand_status, "expanding its records into entries"); if (st != GT_OK) return st; }
    if (dist) {
        // the global pieces, from collectives: a column is non-empty if ANY tile-row has an entry in it; what tile-row q has in MY
        // columns (the order of my send blocks) is q's own count over my segment
        { int st = gt_dist_all_reduce_max_u8(dist, colflag.as<uint8_t>(), span + 1, s); if (st != GT_OK) return st; }
        std::vector<uint64_t> off(p), bytes(p);
        for (uint32_t q = 0; q < p; q++) { off[q] = (uint64_t)q * H * 4; bytes[q] = (uint64_t)H * 4; }
        { int st = gt_dist_exchange_bytes(dist, needme.p, off.data(), bytes.data(), needby.p, off.data(), bytes.data(), s); if (st != GT_OK) return st; }
    }
    const uint64_t nvalid = hc[0];
    GT_REQUIRE(nvalid < 0xFFFFFFFFull, GT_ERR_UNSUPPORTED,
               "tile-row holds %llu entries; column pointers are 32-bit like the reference's Integer_Type "
               "(src/ds/compressed_column.hpp:294): use more ranks", (unsigned long long)nvalid);

    // exclusive prefix sums of the non-empty flags over the whole grid -> IV / JV of every segment
    {
        hipcub::TransformInputIterator<uint32_t, U8ToU32, const uint8_t *> rin(rowflag.as<const uint8_t>(), U8ToU32());
        hipcub::TransformInputIterator<uint32_t, U8ToU32, const uint8_t *> cin(colflag.as<const uint8_t>(), U8ToU32());
        size_t tb = 0;
        ING_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, rin, Srow.as<uint32_t>(), span + 1, s));
        ING_ALLOC(tmp, tb);
        ING_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, rin, Srow.as<uint32_t>(), span + 1, s));
        ING_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, cin, Scol.as<uint32_t>(), span + 1, s));
        if (multi) {
            hipcub::TransformInputIterator<uint32_t, NonZero, const uint32_t *> nin(needme.as<const uint32_t>(), NonZero());
            hipcub::TransformInputIterator<uint32_t, NonZero, const uint32_t *> bin(needby.as<const uint32_t>(), NonZero());
            ING_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, nin, Sneed.as<uint32_t>(), span + 1, s));
            ING_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, bin, Sby.as<uint32_t>(), span + 1, s));
        }
    }
    std::vector<uint32_t> segr(p + 1), segc(p + 1);
    for (uint32_t q = 0; q <= p; q++) {
        ING_HIP(hipMemcpyAsync(&segr[q], Srow.as<uint32_t>() + (uint64_t)q * H, 4, hipMemcpyDeviceToHost, s));
        ING_HIP(hipMemcpyAsync(&segc[q], Scol.as<uint32_t>() + (uint64_t)q * H, 4, hipMemcpyDeviceToHost, s));
    }
    ING_HIP(hipStreamSynchronize(s));
    uint32_t seg_stride = 0;
    for (uint32_t q = 0; q < p; q++) seg_stride = std::max(seg_stride, segc[q + 1] - segc[q]);
    if (seg_stride == 0) seg_stride = 1;
    // Several ranks: the exchange of an iteration is cut into K slices (slice of compressed column j = j / T) so that
    // the all-to-all of slice k+1 can overlap phase 1 of slice k.
    uint32_t K = 1;
    if (multi) {
        const char *e = gt_cfg(g, "GRAPHTAP_X_SLICES");
        // Slicing is not free (engine.hip, combine_impl: +0.025 ms per step at K = 2, +0.05 ms at K = 4 on a tile-row of 8)
        // and hides (K-1)/K of the exchange: worth four slices while a rank receives tens of MB over one to three xGMI
        // links (2 or 4 ranks), two at 8 ranks (6 MB per link).
        K = e ? (uint32_t)atoi(e) : (p <= 4 ? 4u : 2u);
        if (K < 1) K = 1;
        if (K > 64) K = 64;
    }
    const uint32_t T = (seg_stride + K - 1) / K;
    g->info.x_slices = K; g->info.slice_width = T;
    g->info.nnzrows = segr[k + 1] - segr[k];
    g->info.nnzcols = segc[k + 1] - segc[k];
    g->info.seg_stride = seg_stride;
    g->info.nnzrows_global = multi ? g->info.nnzrows : segr[p];   // several ranks: only the owned rows are flagged; the driver sums
    g->info.nnzcols_global = segc[p];
    GT_REQUIRE((uint64_t)p * seg_stride < 0xFFFFFFFFull, GT_ERR_UNSUPPORTED, "column id space exceeds 32 bits");
    g->send_counts.assign((size_t)K * p, 0); g->recv_counts.assign((size_t)K * p, 0);
    g->send_off.assign(K + 1, 0); g->recv_off.assign(K + 1, 0);
    DevBuf rtab_d;
    if (!multi) {
        g->ncols_total = seg_stride;
        g->recv_off[1] = seg_stride;
    } else {
        // block (k, s) of the local column space / of the send buffer: boundaries, sizes (rounded up to 4 elements so
        // that every block of a collective starts 16-byte aligned; both sides round the same count), offsets
        const uint32_t nb = (K + 1) * p;
        DevBuf clo_d, idx_d, val_d;
        ING_ALLOC(clo_d, nb * 4); ING_ALLOC(idx_d, 2ull * nb * 8); ING_ALLOC(val_d, 2ull * nb * 4);
        k_slice_bounds<<<(nb + TPB - 1) / TPB, TPB, 0, s>>>(Scol.as<uint32_t>(), H, T, p, K, clo_d.as<uint32_t>());
        std::vector<uint32_t> clo(nb), sneed_at(nb), sby_at(nb);
        ING_HIP(hipMemcpyAsync(clo.data(), clo_d.p, nb * 4, hipMemcpyDeviceToHost, s));
        ING_HIP(hipStreamSynchronize(s));
        std::vector<uint64_t> idx(2ull * nb);
        for (uint32_t kk = 0; kk <= K; kk++)
            for (uint32_t q = 0; q < p; q++) {
                idx[kk * p + q] = clo[kk * p + q];                                               // Sneed at block (kk, source q)
                idx[nb + kk * p + q] = (uint64_t)q * H + (clo[kk * p + k] - (uint64_t)k * H);    // Sby of destination q at my slice kk
            }
        ING_HIP(hipMemcpyAsync(idx_d.p, idx.data(), idx.size() * 8, hipMemcpyHostToDevice, s));
        k_gather_u32<<<(nb + TPB - 1) / TPB, TPB, 0, s>>>(Sneed.as<uint32_t>(), idx_d.as<uint64_t>(), nb, val_d.as<uint32_t>());
        k_gather_u32<<<(nb + TPB - 1) / TPB, TPB, 0, s>>>(Sby.as<uint32_t>(), idx_d.as<uint64_t>() + nb, nb, val_d.as<uint32_t>() + nb);
        ING_HIP(hipMemcpyAsync(sneed_at.data(), val_d.p, nb * 4, hipMemcpyDeviceToHost, s));
        ING_HIP(hipMemcpyAsync(sby_at.data(), val_d.as<uint32_t>() + nb, nb * 4, hipMemcpyDeviceToHost, s));
        ING_HIP(hipStreamSynchronize(s));
        std::vector<BlockTab> rtab((size_t)K * p);
        std::vector<SendTab> stab((size_t)K * p);
        uint64_t xo = 0, so = 0;
        for (uint32_t kk = 0; kk < K; kk++) {
            g->recv_off[kk] = xo; g->send_off[kk] = so;
            for (uint32_t q = 0; q < p; q++) {
                const uint32_t nrecv = (sneed_at[(kk + 1) * p + q] - sneed_at[kk * p + q] + 3u) & ~3u;
                const uint32_t nsend = (sby_at[(kk + 1) * p + q] - sby_at[kk * p + q] + 3u) & ~3u;
                GT_REQUIRE(xo + nrecv < 0xFFFFFFF0ull && so + nsend < 0xFFFFFFF0ull, GT_ERR_UNSUPPORTED, "exchange buffers exceed 32-bit indexing");
                rtab[kk * p + q] = BlockTab{(uint32_t)xo};
                stab[kk * p + q] = SendTab{(uint32_t)so};
                g->recv_counts[kk * p + q] = nrecv; g->send_counts[kk * p + q] = nsend;
                xo += nrecv; so += nsend;
            }
            xo = (xo + GT_PB_WINDOW - 1) / GT_PB_WINDOW * GT_PB_WINDOW;   // no phase-1 window straddles two slices
        }
        if (xo == 0) xo = GT_PB_WINDOW;
        g->recv_off[K] = xo; g->send_off[K] = so;
        g->ncols_total = (uint32_t)xo;
        g->send_elems = so;
        // positions inside the blocks: one stable sort of the needed columns by (block, entries descending), one of the
        // (destination, owned column) pairs; block b = k * p + q starts where the blocks before it (in b order) end
        {
            const uint32_t nblk = K * p;
            // Inside a block the columns with at least `hub_min` entries in the receiving tile-row come first, by descending count (the
            // receiver's hub windows); ALL the others follow as one class in column order. Sorted by count all the way down, a block was
            // ~20 count classes, each a sparse sweep of the sender's gather over its whole segment buffer (k_pack_send: 12.6 M gathers
            // pulling ~1.6 GB of lines, 0.08 ms per step on a tile-row of 8 of R-MAT-26); the receiver gains nothing from an order among
            // columns that hold one to a few entries. Both sides derive the order from the same counts. GRAPHTAP_EXCHANGE_HUB_MIN.
            const char *hm = gt_cfg(g, "GRAPHTAP_EXCHANGE_HUB_MIN");
            const uint32_t hub_min = hm ? (uint32_t)atoi(hm) : 8u;
            std::vector<uint32_t> start_me(nblk + 1, 0), start_by(nblk + 1, 0);
            for (uint32_t b = 0; b < nblk; b++) {   // b = kk * p + q, the order of the sort key
                start_me[b + 1] = start_me[b] + (sneed_at[b + p] - sneed_at[b]);
                start_by[b + 1] = start_by[b] + (sby_at[b + p] - sby_at[b]);
            }
            const uint32_t n_me = start_me[nblk], n_by = start_by[nblk];
            DevBuf k1, k2, v1, v2, st_d, srt;
            const uint32_t nmax = std::max(std::max(n_me, n_by), 1u);
            ING_ALLOC(k1, (uint64_t)nmax * 8); ING_ALLOC(k2, (uint64_t)nmax * 8); ING_ALLOC(v1, (uint64_t)nmax * 4); ING_ALLOC(v2, (uint64_t)nmax * 4);
            ING_ALLOC(st_d, (uint64_t)(nblk + 1) * 4);
            size_t tb = 0;
            {
                hipcub::DoubleBuffer<uint64_t> dk(k1.as<uint64_t>(), k2.as<uint64_t>());
                hipcub::DoubleBuffer<uint32_t> dv(v1.as<uint32_t>(), v2.as<uint32_t>());
                ING_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, nmax, 0, 48, s));
            }
            ING_ALLOC(srt, tb);
            for (int side = 0; side < 2; side++) {
                const uint32_t n = side ? n_by : n_me;
                if (!n) continue;
                if (side == 0) k_need_keys<<<grid_for(span), TPB, 0, s>>>(needme.as<uint32_t>(), Sneed.as<uint32_t>(), span, Scol.as<uint32_t>(), H, T, p, hub_min, k1.as<uint64_t>(), v1.as<uint32_t>());
                else k_by_keys<<<grid_for(span), TPB, 0, s>>>(needby.as<uint32_t>(), Sby.as<uint32_t>(), H, T, p, Scol.as<uint32_t>(), row_lo, hub_min, k1.as<uint64_t>(), v1.as<uint32_t>());
                hipcub::DoubleBuffer<uint64_t> dk(k1.as<uint64_t>(), k2.as<uint64_t>());
                hipcub::DoubleBuffer<uint32_t> dv(v1.as<uint32_t>(), v2.as<uint32_t>());
                size_t tb2 = tb;
                ING_HIP(hipcub::DeviceRadixSort::SortPairs(srt.p, tb2, dk, dv, n, 0, 48, s));
                ING_HIP(hipMemcpyAsync(st_d.p, (side ? start_by : start_me).data(), (uint64_t)(nblk + 1) * 4, hipMemcpyHostToDevice, s));
                k_block_positions<<<grid_for(n), TPB, 0, s>>>(dk.Current(), dv.Current(), n, st_d.as<uint32_t>(), (side ? Pby : Pneed).as<uint32_t>());
                ING_HIP(hipStreamSynchronize(s));   // the host vectors and the double buffers are reused by the other side
            }
        }
        ING_ALLOC(rtab_d, rtab.size() * sizeof(BlockTab));
        ING_HIP(hipMemcpyAsync(rtab_d.p, rtab.data(), rtab.size() * sizeof(BlockTab), hipMemcpyHostToDevice, s));
        DevBuf stab_d;
        ING_ALLOC(stab_d, stab.size() * sizeof(SendTab));
        ING_HIP(hipMemcpyAsync(stab_d.p, stab.data(), stab.size() * sizeof(SendTab), hipMemcpyHostToDevice, s));
        if (hipMalloc((void **)&g->loc2glob, (uint64_t)g->ncols_total * 4) != hipSuccess ||
            hipMalloc((void **)&g->send_idx, std::max<uint64_t>(so, 1) * 4) != hipSuccess) {
            gt_set_error("ingest: out of device memory for the exchange tables");
            return GT_ERR_HIP;
        }
        ING_HIP(hipMemsetAsync(g->loc2glob, 0xFF, (uint64_t)g->ncols_total * 4, s));
        ING_HIP(hipMemsetAsync(g->send_idx, 0, std::max<uint64_t>(so, 1) * 4, s));
        k_local_to_global<<<grid_for(span), TPB, 0, s>>>(needme.as<uint32_t>(), span, Scol.as<uint32_t>(), Pneed.as<uint32_t>(),
                                                         rtab_d.as<BlockTab>(), H, T, p, seg_stride, g->loc2glob);
        k_send_list<<<grid_for(span), TPB, 0, s>>>(needby.as<uint32_t>(), Pby.as<uint32_t>(), Scol.as<uint32_t>(), stab_d.as<SendTab>(),
                                                   H, T, p, row_lo, g->send_idx);
        ING_HIP(hipStreamSynchronize(s));   // stab / idx scratch go out of scope here
    }
    g->info.ncols_local = g->ncols_total;
    g->info.send_elems = (uint32_t)g->send_elems;

    if (nvalid) k_remap_cols<<<grid_for(nvalid), TPB, 0, s>>>(keys.as<uint64_t>(), nvalid, Scol.as<uint32_t>(), multi ? Pneed.as<uint32_t>() : nullptr,
                                                        multi ? rtab_d.as<BlockTab>() : nullptr, H, T, p);
    // column-major order (ColSort, ds/triple.hpp:78-98): (col,row); with weights (col,row,weight)
    // so that the first copy of a duplicate (row,col) carries its minimum weight. Only the nvalid kept
    // entries are sorted (k_expand compacts). Radix sort is stable.
    uint64_t *sorted_keys = keys.as<uint64_t>();
    uint32_t *sorted_wts = weighted ? wts.as<uint32_t>() : nullptr;
    if (nvalid) {
        ING_ALLOC(keys2, nvalid * 8);
        if (weighted) ING_ALLOC(wts2, nvalid * 4);
        size_t tb1 = 0, tb2 = 0;
        hipcub::DoubleBuffer<uint64_t> dk(keys.as<uint64_t>(), keys2.as<uint64_t>());
        if (weighted) {
            hipcub::DoubleBuffer<uint32_t> dw(wts.as<uint32_t>(), wts2.as<uint32_t>());
            ING_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, dw, dk, nvalid, 0, 32, s));
            ING_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, dk, dw, nvalid, 0, 64, s));
            DevBuf st; ING_ALLOC(st, std::max(tb1, tb2));
            size_t tb = std::max(tb1, tb2);
            ING_HIP(hipcub::DeviceRadixSort::SortPairs(st.p, tb, dw, dk, nvalid, 0, 32, s));   // by weight
            tb = std::max(tb1, tb2);
            ING_HIP(hipcub::DeviceRadixSort::SortPairs(st.p, tb, dk, dw, nvalid, 0, 64, s));   // then by (col,row)
            ING_HIP(hipStreamSynchronize(s));
            sorted_keys = dk.Current(); sorted_wts = dw.Current();
        } else {
            ING_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb1, dk, nvalid, 0, 64, s));
            DevBuf st; ING_ALLOC(st, tb1);
            ING_HIP(hipcub::DeviceRadixSort::SortKeys(st.p, tb1, dk, nvalid, 0, 64, s));
            ING_HIP(hipStreamSynchronize(s));
            sorted_keys = dk.Current();
        }
    }

    // dedupe (only adjacent equal (row,col) exist after the sort) + compaction positions
    DevBuf head, pos;
    ING_ALLOC(head, (nvalid + 1) * 4); ING_ALLOC(pos, (nvalid + 1) * 4);
    uint32_t nnz = 0;
    if (nvalid) {
        k_head_flags<<<grid_for(nvalid), TPB, 0, s>>>(sorted_keys, nvalid, !f.parallel_edges, head.as<uint32_t>());
        size_t tb = 0;
        ING_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, head.as<uint32_t>(), pos.as<uint32_t>(), nvalid, s));
        DevBuf st; ING_ALLOC(st, tb);
        ING_HIP(hipcub::DeviceScan::ExclusiveSum(st.p, tb, head.as<uint32_t>(), pos.as<uint32_t>(), nvalid, s));
        uint32_t lastpos = 0, lasthead = 0;
        ING_HIP(hipMemcpyAsync(&lastpos, pos.as<uint32_t>() + (nvalid - 1), 4, hipMemcpyDeviceToHost, s));
        ING_HIP(hipMemcpyAsync(&lasthead, head.as<uint32_t>() + (nvalid - 1), 4, hipMemcpyDeviceToHost, s));
        ING_HIP(hipStreamSynchronize(s));
        nnz = lastpos + lasthead;
    }
    g->info.nnz_local = nnz;
    g->info.nnz_global = (p == 1) ? nnz : 0;  // multi-rank: the driver sums nnz_local over ranks

    // persistent arrays
    auto dmalloc = [&](uint32_t **ptr, uint64_t elems) -> int {
        return hipMalloc((void **)ptr, (elems ? elems : 1) * sizeof(uint32_t)) == hipSuccess ? 0 : -1;
    };
    const uint32_t nr = g->info.nnzrows, nc = g->info.nnzcols;
    if (dmalloc(&g->IA, nnz) || dmalloc(&g->JI, nnz) || (weighted && dmalloc(&g->A, nnz)) ||
        dmalloc(&g->JA, (uint64_t)g->ncols_total + 1) || dmalloc(&g->IV, H) || dmalloc(&g->JV, H) ||
        dmalloc(&g->IR, nr) || dmalloc(&g->JC, nc) || dmalloc(&g->R2C, nr) ||
        hipMalloc((void **)&g->IJ, H) != hipSuccess) {
        gt_set_error("ingest: out of device memory for the tile arrays");
        return GT_ERR_HIP;
    }
    if (nvalid)
        k_populate<<<grid_for(nvalid), TPB, 0, s>>>(sorted_keys, sorted_wts, nvalid, head.as<uint32_t>(), pos.as<uint32_t>(),
                                                    Srow.as<uint32_t>(), row_lo,
                                                    g->IA, g->JI, g->A);
    k_col_ptr<<<grid_for((uint64_t)g->ncols_total + 1), TPB, 0, s>>>(g->JI, nnz, g->ncols_total, g->JA);
    unsigned int *classes = (unsigned int *)(counters.as<unsigned long long>() + 4);
    k_segment_vectors<<<grid_for(H), TPB, 0, s>>>(rowflag.as<uint8_t>(), colflag.as<uint8_t>(), Srow.as<uint32_t>(),
                                                  Scol.as<uint32_t>(), H, row_lo, g->IJ, g->IV, g->JV, g->IR, g->JC,
                                                  g->R2C, classes);
    unsigned int hcl[3];
    ING_HIP(hipMemcpyAsync(hcl, classes, sizeof(hcl), hipMemcpyDeviceToHost, s));
    ING_HIP(hipStreamSynchronize(s));
    ING_HIP(hipGetLastError());
    g->info.regular = hcl[0]; g->info.source_rows = hcl[1]; g->info.sink_cols = hcl[2];
    return GT_OK;
}
