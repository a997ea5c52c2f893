// pb.hip -- "propagation blocking" SpMV for the owned tile-row: the production kernels.
//
// Same result as the edge-parallel kernel in kernels.hip (and as the reference's column-major
// loop, src/vp/vertex_program.hpp:1162-1173 / 1490-1503), organised so that every HBM access is a
// stream and every random access hits LDS:
//
//   phase 1  "scatter"   one workgroup per CHUNK of the entry stream. A chunk belongs to one WINDOW of consecutive
//            slots of the message vector x, whose messages are staged in LDS (coalesced load); for every entry, in
//            (chunk, row-bin) order, the value x[col] (+ w) goes to the entry's slot of the row-bin-major value stream VAL.
//            Two kinds of windows (x is laid out HUBS FIRST on a single rank, gt_layout_build):
//              DENSE  windows (W = 16383 slots, the columns of largest out-degree): consecutive entries of the same row
//                     are PRE-AGGREGATED (+ or min) over whole 256-entry groups before they leave the chip -- a lane
//                     combines its quad in registers, what a quad leaves open reaches the lane that holds the stretch's end
//                     through one segmented DPP scan whose flags are a scalar mask -- and the outputs of a group leave in
//                     stores of 64 consecutive slots (staged through an LDS row; the min kernels store directly) (k_pb_scatter);
//              SPARSE windows (WS = 16384 slots, low-degree columns, nothing to aggregate): one slot per entry, the
//                     slot is the entry's position: four values per lane stored with one 16-byte store (same kernel, other branch).
//   phase 2  "gather"    one workgroup per ROW BIN (R = 16384 consecutive compressed rows; heavy
//            bins are split by slot count): the bin's partial accumulators live in LDS (R x 8 B =
//            128 KiB for f64), the bin's slice of VAL and of the static bin-local row ids LROW is
//            streamed once (16-byte / 8-byte loads) and combined with LDS atomics (ds_add_f64 /
//            ds_min_u32), then merged into y (plain RMW when the bin has one workgroup, device atomics
//            when it was split) -- or, for PageRank, applied on the spot (fused applicator).
//
// A RUN is the set of entries of one (chunk, bin) pair; it is contiguous in both orders and, inside a run,
// entries are sorted by (row, col). Runs are padded to a multiple of four entries in the v-order and to a multiple of
// four outputs in the k-order (dense pad inputs read the neutral message from a spare LDS slot; pad outputs target
// the dummy accumulator row R), so quads never straddle runs and every access is aligned.
// Static per-graph data, built once on the device (rocPRIM sorts / scans):
//   LCOL[v]  u16  v-order = runs sorted by (chunk, bin), entries inside by (row, col); bits 0-13: slot - window start
//                 (W = pad, dense), bit 15: last entry of its (256-entry group, row) stretch (dense), bit 14: first
//                 entry of a run
//   LROW[k]  u16  k-order = outputs, runs sorted by (bin, chunk); row & (R-1), R for a pad
//   WT[v]    u8 / u16 / u32 (the narrowest the largest weight fits) weights in v-order (min-plus only)
//   G[g]     32 B per 256 entries of the v-order: the k-slot of lane 0's first output and, for the first six
//                 run heads of the group, (k-slot of the run - outputs of the group before the head), so that no
//                 load of phase 1 depends on another load; lane j of a wave holds dword j & 7 and the run constants are
//                 read off it with scalar readlanes (the direct-store form: one ds_bpermute per lane)
//   KSTART[s]     k-slot where run s starts - outputs of the whole v-order before it (groups with 7+ run heads)
// HBM traffic per entry per SpMV: 2 + 0.125 (phase 1 in) + (F + F + 2) / D (value stream out and back, LROW),
// D = entries per output, F = bytes of a message in flight.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#include "gt_internal.h"

namespace {

constexpr int RB = GT_PB_ROW_BIN_BITS;  // log2 rows per bin
constexpr uint32_t R = 1u << RB;       // 16384 rows: 128 KiB of f64 accumulators in LDS (+ one dummy row for pads)
constexpr uint32_t W = GT_PB_WINDOW;   // 16383 slots per dense window (+ the neutral slot): 64 KiB (4-byte messages) or 128 KiB (f64) of LDS
static_assert(W <= 0x3FFF, "a dense column offset and the pad slot W must fit the 14 column bits of LCOL");
constexpr uint32_t WS = GT_PB_SPARSE_WINDOW;   // 16384 slots per sparse window: 64 KiB (4-byte messages) or 128 KiB (f64)
// Entries per chunk. A window is one chunk unless it holds more than `ch` entries; such windows are cut by row bin (k_win_plan),
// which leaves their runs whole, so `ch` only sets the granularity of the launch. Small graphs: the grid must
// still be several times the 512 resident phase-1 workgroups, so `ch` shrinks until there are >= ~2048 chunks.
static uint32_t ch_default(const gt_graph *g, uint32_t nnz, uint32_t nwin) {
    const char *e = gt_cfg(g, "GRAPHTAP_PB_CH");
    if (e) return 1u << atoi(e);
    if (nwin >= 2048) return 1u << 19;   // the windows alone give enough chunks: only the heavy windows are cut
    uint32_t ch = 1u << 14;
    while (ch < (1u << 19) && (uint64_t)ch * 2048 <= nnz) ch <<= 1;
    return ch;
}
constexpr uint32_t EPW = 1u << 19;     // value-stream slots per phase-2 workgroup: a bin below it keeps ONE workgroup, whose flush
                                       // can then apply PageRank's rows directly (2^18 and 2^19 stream equally fast, 2^20 is 2-10 % slower)
constexpr int P1_THREADS = 1024;
constexpr uint32_t P1_QUEUES = 64;   // counters of the persistent phase 1: launches that may be in flight at once
constexpr int P2_THREADS = 1024;
#ifndef GT_P2_U
#define GT_P2_U 2
#endif
constexpr int P2_U = GT_P2_U;            // quads in flight per lane in phase 2
// Dense windows: consecutive entries of one row inside one aligned block of AGG entries of the v-order share an output.
// R-MAT-26, natural column order: 1.40 entries per output with AGG = 4, 1.50 with 8, 1.56 with 16, 1.60 with 64, 1.62 with 256
// and without a limit; with the hubs-first layout the dense windows hold 4+ entries per (window, row) pair and only whole
// groups collect them.
constexpr uint32_t AGG_MASK = 255;
// flag bits of an LCOL entry. GEND sits in the SIGN bit of the 16-bit word: phase 1 tests it four times per quad, and a sign test of
// either half of a dword is one compare (v_cmp_gt_i16 / v_cmp_gt_i32), a test of any other bit two instructions
constexpr uint16_t GEND = 0x8000;      // dense: last entry of its (group, row) stretch
constexpr uint16_t COLMASK = 0x3FFF;
constexpr uint16_t HEAD = 0x4000;      // first entry of a run (entry 0 of a quad only)
constexpr int TPB = 256;

inline unsigned grid_for(uint64_t n) {
    uint64_t b = (n + TPB - 1) / TPB;
    if (b < 1) b = 1;
    if (b > 256u * 32u) b = 256u * 32u;
    return (unsigned)b;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) gt_scratch_free(p); }
    int alloc(uint64_t bytes) { return gt_scratch_malloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : -1; }
    template <class T> T *as() { return (T *)p; }
};

struct GroupRec { uint32_t k0, k[6], s; };   // 32 bytes, k-slots in output units
struct BinWork { uint32_t bin, k0, k1, single, c_lo, c_hi, pad0, pad1; };   // [c_lo, c_hi]: chunks whose runs overlap [k0, k1)

// Windows of the message vector: ndw dense windows of W slots, then sparse windows of WS slots.
// Entries are further split by ROW CLASS (ncls = 2 on graphs without an exchange layout): class 1 = entries of SOURCE rows
// (vertices with in-edges but no out-edges, R2C == ~0u). Under TCSC_CF the reference leaves them out of every iteration but
// the last (computation filtering, compressed_column.hpp:671-708, vp:1264-1317); here they get chunks of their own -- the
// "virtual window" of an entry is cls * nwin + window -- which PageRank/TCSC_CF does not launch until the last iteration.
// wd / ws: slots per dense / sparse window of THIS build: W / WS, or -- the WIDE build (gt_pb::wide) -- 2 W / 2 WS: a wide window is a
// pair of consecutive narrow ones (the layout of x is the same for both builds), its neutral slot sits at LDS index 2 W.
struct WinGeom { uint32_t ndw, dense_end, nwin, x_len, ncls, nvwin, wd = GT_PB_WINDOW, ws = GT_PB_SPARSE_WINDOW; };
__device__ __forceinline__ uint32_t row_class(const uint32_t *__restrict__ srcbits, uint32_t r) { return srcbits ? (srcbits[r >> 5] >> (r & 31)) & 1u : 0u; }
__host__ __device__ inline uint32_t win_of(const WinGeom &g, uint32_t slot) { return slot < g.dense_end ? slot / g.wd : g.ndw + (slot - g.dense_end) / g.ws; }
__host__ __device__ inline uint32_t win_col0(const WinGeom &g, uint32_t q) { return q < g.ndw ? q * g.wd : g.dense_end + (q - g.ndw) * g.ws; }
__device__ __forceinline__ uint32_t slot_of(const uint32_t *__restrict__ xslot, uint32_t c) { return xslot ? xslot[c] : c; }

// ------------------------------------------------------------------ layout kernels (gt_layout_build)
__global__ void k_col_degrees(const uint32_t *__restrict__ JA, uint32_t nc, uint32_t thr, uint32_t *__restrict__ deg, uint32_t *__restrict__ hubflag,
                              uint32_t *__restrict__ tailflag) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) {
        const uint32_t d = JA[c + 1] - JA[c];
        deg[c] = d; hubflag[c] = d >= thr ? 1u : 0u; tailflag[c] = d >= thr ? 0u : 1u;
    }
}
__global__ void k_hub_list(const uint32_t *__restrict__ deg, const uint32_t *__restrict__ hubflag, const uint32_t *__restrict__ hubpos, uint32_t nc,
                           uint32_t *__restrict__ key, uint32_t *__restrict__ col) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x)
        if (hubflag[c]) { key[hubpos[c]] = ~deg[c]; col[hubpos[c]] = c; }   // ascending ~degree = descending degree
}
__global__ void k_slots_hub(const uint32_t *__restrict__ col_sorted, uint32_t nhub, uint32_t *__restrict__ xslot) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nhub; i += gridDim.x * blockDim.x) xslot[col_sorted[i]] = i;
}
__global__ void k_slots_tail(const uint32_t *__restrict__ tailflag, const uint32_t *__restrict__ tailpos, uint32_t nc, uint32_t base,
                             uint32_t *__restrict__ xslot) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x)
        if (tailflag[c]) xslot[c] = base + tailpos[c];
}
__global__ void k_slot_vertices(const uint32_t *__restrict__ xslot, const uint32_t *__restrict__ JC, uint32_t nc, uint32_t *__restrict__ XV,
                                uint32_t *__restrict__ xcol) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nc; c += gridDim.x * blockDim.x) { XV[xslot[c]] = JC[c]; xcol[xslot[c]] = c; }
}
__global__ void k_row_slots(const uint32_t *__restrict__ xslot, const uint32_t *__restrict__ R2C, uint32_t nr, uint32_t *__restrict__ R2X) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += gridDim.x * blockDim.x) {
        const uint32_t c = R2C[r];
        R2X[r] = c == 0xFFFFFFFFu ? c : xslot[c];
    }
}

// ------------------------------------------------------------------ build kernels
// entries per window. Consecutive columns mostly share a window: one atomic per wave when they all do.
__global__ void k_win_count(const uint32_t *__restrict__ JA, uint32_t nc, const uint32_t *__restrict__ xslot, WinGeom geom, uint32_t *__restrict__ wcount) {
    const uint32_t n64 = (nc + 63) & ~63u;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n64; c += gridDim.x * blockDim.x) {
        const bool in = c < nc;
        const uint32_t q = in ? win_of(geom, slot_of(xslot, c)) : 0xFFFFFFFFu;
        uint32_t d = in ? JA[c + 1] - JA[c] : 0u;
        const uint32_t q0 = __builtin_amdgcn_readfirstlane(q);
        if (__all(q == q0 || !in)) {
            for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o);
            if ((threadIdx.x & 63) == 0 && d && q0 != 0xFFFFFFFFu) atomicAdd(&wcount[q0], d);
        } else if (in && d) atomicAdd(&wcount[q], d);
    }
}
__global__ void k_source_bits(const uint32_t *__restrict__ R2C, uint32_t nr, uint32_t *__restrict__ bits) {
    const uint32_t n64 = (nr + 63) & ~63u;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n64; r += gridDim.x * blockDim.x) {
        const uint64_t b = __ballot(r < nr && R2C[r] == 0xFFFFFFFFu);
        if ((threadIdx.x & 63) == 0) { bits[r >> 5] = (uint32_t)b; bits[(r >> 5) + 1] = (uint32_t)(b >> 32); }
    }
}
// Histogram of per-entry keys whose neighbours mostly agree (the entries arrive column-major with ascending rows: a wave's 64
// consecutive entries share the column, hence the window, and almost always the row bin): every maximal run of equal keys
// inside a wave does ONE atomic with its length. `key` = ~0u: not counted. All 64 lanes must call.
__device__ __forceinline__ void wave_run_add(uint32_t *__restrict__ counters, uint32_t key, uint32_t lane) {
    const uint32_t prev = __shfl_up(key, 1);
    const bool head = lane == 0 || key != prev;
    const uint64_t heads = __ballot(head);
    if (head && key != 0xFFFFFFFFu) {
        const uint64_t later = lane == 63 ? 0ull : (heads >> (lane + 1));   // heads after this lane
        const uint32_t len = later ? (uint32_t)__builtin_ctzll((unsigned long long)later) + 1u : 64u - lane;
        atomicAdd(&counters[key], len);
    }
}
// moves the entries of source rows from their window's count to the class-1 copy of the window
__global__ void k_win_count_src(const uint32_t *__restrict__ JI, const uint32_t *__restrict__ IA, uint64_t nnz, const uint32_t *__restrict__ xslot,
                                WinGeom geom, const uint32_t *__restrict__ srcbits, uint32_t *__restrict__ wcount_src /* wcount + nwin */) {
    const uint64_t n64 = (nnz + 63) & ~63ull;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n64; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t key = 0xFFFFFFFFu;
        if (e < nnz && row_class(srcbits, IA[e])) key = win_of(geom, slot_of(xslot, JI[e]));
        wave_run_add(wcount_src, key, lane);
    }
}
__global__ void k_win_count_move(uint32_t *__restrict__ wcount, uint32_t nwin) {   // class 0 keeps what class 1 did not take
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nwin; q += gridDim.x * blockDim.x) wcount[q] -= wcount[nwin + q];
}
// A window (dense: W slots, sparse: WS slots) is one chunk, unless it holds more than `ch` entries: such a window is cut into
// several chunks by ROW BIN -- consecutive bins are packed into chunks of ~n/ceil(n/ch) entries, so every run holds
// ALL entries its window has in that bin: as few and as long runs as the window allows and every same-row neighbour to
// pre-aggregate. A bin that alone exceeds 1.5x the chunk size (hub rows x hub columns) gets m chunks of its own and its
// entries go to them by slot mod m.
// plan[h * nbins + bin] = first chunk of the bin inside its window | (m << 20); h = index of the window among the cut ones.
constexpr uint32_t PLAN_SUB_MASK = (1u << 20) - 1;
__global__ void k_win_sizes(const uint32_t *__restrict__ wcount, uint32_t nwin, uint32_t ch, uint32_t *__restrict__ nsub, uint32_t *__restrict__ cutflag) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nwin; q += gridDim.x * blockDim.x) {
        const uint32_t n = wcount[q];
        nsub[q] = (n + ch - 1) / ch;          // final for the windows that stay whole
        cutflag[q] = (n > ch) ? 1u : 0u;
    }
}
// entries of every cut window per row bin, counted into plan[] (one atomic per run of equal (window, bin) inside a wave: per
// entry it was 0.31 s for R-MAT-26 -- the counters of hub windows x hub bins take hundreds of thousands of increments each)
__global__ void k_win_hist(const uint32_t *__restrict__ JI, const uint32_t *__restrict__ IA, uint64_t nnz, const uint32_t *__restrict__ xslot, WinGeom geom,
                           const uint32_t *__restrict__ srcbits, uint32_t nbins, const uint32_t *__restrict__ cutflag, const uint32_t *__restrict__ cutidx,
                           uint32_t *__restrict__ plan) {
    const uint64_t n64 = (nnz + 63) & ~63ull;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n64; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t key = 0xFFFFFFFFu;
        if (e < nnz) {
            const uint32_t r = IA[e], q = row_class(srcbits, r) * geom.nwin + win_of(geom, slot_of(xslot, JI[e]));
            if (cutflag[q]) key = cutidx[q] * nbins + (r >> RB);
        }
        wave_run_add(plan, key, lane);
    }
}
// counts -> plan, in place; one thread per cut window (a few thousand bins each)
__global__ void k_win_plan(const uint32_t *__restrict__ wcount, uint32_t nwin, uint32_t ch, uint32_t nbins,
                           const uint32_t *__restrict__ cutflag, const uint32_t *__restrict__ cutidx, uint32_t *__restrict__ plan,
                           uint32_t *__restrict__ nsub) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nwin; q += gridDim.x * blockDim.x) {
        if (!cutflag[q]) continue;
        const uint32_t n = wcount[q];
        const uint32_t ns = (n + ch - 1) / ch, target = (n + ns - 1) / ns;
        uint32_t *pl = plan + (uint64_t)cutidx[q] * nbins;
        uint32_t sub = 0, acc = 0;
        for (uint32_t b = 0; b < nbins; b++) {
            const uint32_t h = pl[b];
            if (h > target + target / 2) {                       // a bin for several chunks of its own
                if (acc) { sub++; acc = 0; }
                uint32_t m = (h + target - 1) / target;
                if (m > 4095) m = 4095;
                pl[b] = sub | (m << 20); sub += m;
            } else {
                if (acc && acc + h > target + target / 4) { sub++; acc = 0; }
                pl[b] = sub | (1u << 20); acc += h;
            }
        }
        if (acc) sub++;
        nsub[q] = sub;
    }
}
__global__ void k_fill_chunks(uint32_t nwin, const uint32_t *__restrict__ nsub, const uint32_t *__restrict__ cbase, WinGeom geom, uint32_t *__restrict__ ccol0) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nwin; q += gridDim.x * blockDim.x)
        for (uint32_t k = 0; k < nsub[q]; k++) ccol0[cbase[q] + k] = win_col0(geom, q % geom.nwin);
}
// sort key of every entry: (chunk, row bin, row inside the bin) -> runs come out sorted by (row, col)
__global__ void k_keys(const uint32_t *__restrict__ JI, const uint32_t *__restrict__ IA, uint64_t nnz, const uint32_t *__restrict__ xslot, WinGeom geom,
                       const uint32_t *__restrict__ srcbits, const uint32_t *__restrict__ cbase, const uint32_t *__restrict__ cutflag, const uint32_t *__restrict__ cutidx,
                       const uint32_t *__restrict__ plan, uint32_t nbins, int binbits, uint64_t *__restrict__ key, uint32_t *__restrict__ idx) {
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < nnz; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t slot = slot_of(xslot, JI[e]), qw = win_of(geom, slot);
        const uint32_t r = IA[e], bin = r >> RB, q = row_class(srcbits, r) * geom.nwin + qw;
        uint32_t sub = 0;
        if (cutflag[q]) { const uint32_t p = plan[(uint64_t)cutidx[q] * nbins + bin], m = p >> 20; sub = (p & PLAN_SUB_MASK) + (m > 1 ? (slot - win_col0(geom, qw)) % m : 0u); }
        key[e] = ((((uint64_t)(cbase[q] + sub) << binbits) | bin) << RB) | (r & (R - 1));
        idx[e] = (uint32_t)e;
    }
}
// run key (chunk, bin) of every sorted entry
__global__ void k_run_keys(const uint64_t *__restrict__ key64, uint64_t n, uint32_t *__restrict__ key) {
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x)
        key[v] = (uint32_t)(key64[v] >> RB);
}
__global__ void k_count_run_rows(const uint64_t *__restrict__ key64, uint64_t n, unsigned long long *__restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        c += (i == 0 || key64[i] != key64[i - 1]);   // key = (chunk, bin, row): equal keys = same run and row
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
__global__ void k_iota(uint32_t *__restrict__ p, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}
template <class T> __global__ void k_fill_t(T *__restrict__ p, uint64_t n, T v) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_heads(const uint32_t *__restrict__ key, uint64_t n, uint32_t *__restrict__ head) {
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x)
        head[v] = (v == 0 || key[v] != key[v - 1]) ? 1u : 0u;
}
// sid[v] = inclusive scan of head; run s = sid - 1
__global__ void k_runs(const uint32_t *__restrict__ key, const uint32_t *__restrict__ sid, uint64_t n,
                       uint32_t *__restrict__ vstart, uint32_t *__restrict__ runkey) {
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x)
        if (v == 0 || key[v] != key[v - 1]) { uint32_t s = sid[v] - 1; vstart[s] = (uint32_t)v; runkey[s] = key[v]; }
}
__global__ void k_run_lens(const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ runkey, uint32_t nrun, uint32_t nnz,
                           uint32_t binmask, uint32_t *__restrict__ len, uint32_t *__restrict__ lenpad, uint32_t *__restrict__ runbin) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nrun; s += gridDim.x * blockDim.x) {
        uint32_t l = (s + 1 < nrun ? vstart[s + 1] : nnz) - vstart[s];
        len[s] = l; lenpad[s] = (l + 3) & ~3u; runbin[s] = runkey[s] & binmask;
    }
}
__global__ void k_gather_u32(const uint32_t *__restrict__ order, const uint32_t *__restrict__ src, uint32_t n, uint32_t *__restrict__ dst) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) dst[t] = src[order[t]];
}
__global__ void k_scatter_u32(const uint32_t *__restrict__ order, const uint32_t *__restrict__ src, uint32_t n, uint32_t *__restrict__ dst) {
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) dst[order[t]] = src[t];
}
// first index i in [0,n) with a[i] >= key (a sorted ascending)
__host__ __device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t *__restrict__ a, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ void k_bin_offsets(const uint32_t *__restrict__ bins_sorted, const uint32_t *__restrict__ kscan, uint32_t nrun, uint32_t np,
                              uint32_t nbins, uint32_t *__restrict__ bin_off) {
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b <= nbins; b += gridDim.x * blockDim.x) {
        uint32_t lo = lower_bound_u32(bins_sorted, nrun, b);
        bin_off[b] = lo < nrun ? kscan[lo] : np;
    }
}
// the last run of every chunk takes the pads that round the chunk's padded entry count up to a multiple of 256
__global__ void k_align_chunks(const uint32_t *__restrict__ runkey, uint32_t nrun, int binbits, const uint32_t *__restrict__ pvstart,
                               uint32_t *__restrict__ lenpad) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nrun; s += gridDim.x * blockDim.x) {
        const uint32_t c = runkey[s] >> binbits;
        if (s + 1 < nrun && (runkey[s + 1] >> binbits) == c) continue;
        const uint32_t a = lower_bound_u32(runkey, nrun, c << binbits);
        const uint32_t total = pvstart[s + 1] - pvstart[a];
        lenpad[s] += (256u - (total & 255u)) & 255u;
    }
}
// chunk c covers the runs whose key >> binbits == c (keys are sorted): its padded v-range
__global__ void k_chunk_ranges(const uint32_t *__restrict__ runkey, uint32_t nrun, int binbits, const uint32_t *__restrict__ pvstart,
                               uint32_t nchunks, uint32_t *__restrict__ cv0, uint32_t *__restrict__ cv1) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += gridDim.x * blockDim.x) {
        uint32_t a = lower_bound_u32(runkey, nrun, c << binbits);
        uint32_t b = (c + 1 < nchunks) ? lower_bound_u32(runkey, nrun, (c + 1) << binbits) : nrun;
        cv0[c] = pvstart[a]; cv1[c] = pvstart[b];
    }
}
// per chunk: the pads' defaults. Sparse chunks: every padded position is an output (E = 1) and pads read column 0 of the window;
// dense chunks: pads read the neutral slot PADCOL and end nothing.
__global__ void k_chunk_defaults(const uint32_t *__restrict__ cv0, const uint32_t *__restrict__ cv1, const uint32_t *__restrict__ ccol0, uint32_t dense_end,
                                 uint32_t *__restrict__ E, uint16_t *__restrict__ LCOL, uint16_t padcol) {
    const uint32_t c = blockIdx.x;
    const bool sparse = ccol0[c] >= dense_end;
    for (uint32_t pv = cv0[c] + threadIdx.x; pv < cv1[c]; pv += blockDim.x) {
        if (E) E[pv] = sparse ? 1u : 0u;
        if (LCOL) LCOL[pv] = sparse ? (uint16_t)0 : padcol;
    }
}
// E[pv] = 1 when the entry at padded v-position pv ends an output. Dense chunks: the last entry of a maximal
// stretch of one row inside one run and inside one aligned block of AGG_MASK + 1 entries. Sparse chunks: every position
// (pads included, k_chunk_defaults) is its own output.
__global__ void k_group_ends(const uint64_t *__restrict__ key64, const uint32_t *__restrict__ sid, uint64_t n,
                             const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ len, const uint32_t *__restrict__ pvstart,
                             const uint32_t *__restrict__ ccol0, uint32_t dense_end, int chunk_shift, uint32_t *__restrict__ E) {
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = sid[v] - 1, o = (uint32_t)v - vstart[s], pv = pvstart[s] + o;
        const bool sparse = ccol0[(uint32_t)(key64[v] >> chunk_shift)] >= dense_end;
        const bool last_of_run = (o + 1 == len[s]);
        const bool last_of_block = ((pv & AGG_MASK) == AGG_MASK);
        const bool row_changes = !last_of_run && ((key64[v + 1] & (R - 1)) != (key64[v] & (R - 1)));
        E[pv] = (sparse || last_of_run || last_of_block || row_changes) ? 1u : 0u;
    }
}
// [pb] stats: how many outputs the DENSE chunks would have if stretches could span 2^k consecutive entries (mask = 2^k - 1)
__global__ void k_count_ends(const uint64_t *__restrict__ key64, const uint32_t *__restrict__ sid, uint64_t n, const uint32_t *__restrict__ vstart,
                             const uint32_t *__restrict__ len, const uint32_t *__restrict__ pvstart, const uint32_t *__restrict__ ccol0,
                             uint32_t dense_end, int chunk_shift, uint32_t mask, unsigned long long *__restrict__ out) {
    unsigned long long c = 0;
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = sid[v] - 1, o = (uint32_t)v - vstart[s], pv = pvstart[s] + o;
        if (ccol0[(uint32_t)(key64[v] >> chunk_shift)] >= dense_end) continue;
        const bool last_of_run = (o + 1 == len[s]);
        c += (last_of_run || (pv & mask) == mask || ((key64[v + 1] & (R - 1)) != (key64[v] & (R - 1))));
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}
// outputs of run s = X[pvstart[s+1]] - X[pvstart[s]] (X = exclusive scan of E), padded to a multiple of 4
__global__ void k_run_outputs(const uint32_t *__restrict__ pvstart, const uint32_t *__restrict__ X, uint32_t nrun,
                              uint32_t *__restrict__ noutpad, uint32_t align) {
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nrun; s += gridDim.x * blockDim.x)
        noutpad[s] = (X[pvstart[s + 1]] - X[pvstart[s]] + align - 1) & ~(align - 1);
}
__global__ void k_static_streams(const uint64_t *__restrict__ key64, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ sid,
                                 uint64_t n, int binbits, const uint32_t *__restrict__ ccol0, const uint32_t *__restrict__ JI,
                                 const uint32_t *__restrict__ xslot, const uint32_t *__restrict__ A, const uint32_t *__restrict__ vstart,
                                 const uint32_t *__restrict__ pvstart, const uint32_t *__restrict__ pkstart, const uint32_t *__restrict__ E,
                                 const uint32_t *__restrict__ X, uint32_t dense_end, uint16_t *__restrict__ LCOL, uint16_t *__restrict__ LROW,
                                 void *__restrict__ WT, int wt_bytes, uint16_t head_flag) {
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t e = idx[v], c = (uint32_t)(key64[v] >> (RB + binbits)), s = sid[v] - 1, o = (uint32_t)v - vstart[s];
        const uint32_t pv = pvstart[s] + o, ge = E[pv];
        LCOL[pv] = (uint16_t)((slot_of(xslot, JI[e]) - ccol0[c]) | ((ge && ccol0[c] < dense_end) ? GEND : 0) | (o == 0 ? head_flag : 0));   // (the wide build has no bit left for HEAD: its heads are a mask in the group record)
        if (ge) LROW[pkstart[s] + (X[pv] - X[pvstart[s]])] = (uint16_t)(key64[v] & (R - 1));
        if (WT) { if (wt_bytes == 1) ((uint8_t *)WT)[pv] = (uint8_t)A[e]; else if (wt_bytes == 2) ((uint16_t *)WT)[pv] = (uint16_t)A[e]; else ((uint32_t *)WT)[pv] = A[e]; }
    }
}
// Group table: one record per 256 padded entries (64 lanes x 4) of the v-order, k-slots in output units.
__global__ void k_group_table(const uint32_t *__restrict__ pvstart, const uint32_t *__restrict__ pkstart, const uint32_t *__restrict__ X,
                              uint32_t nrun, uint32_t np, GroupRec *__restrict__ G) {
    const uint64_t ngroups = ((uint64_t)np + 255) / 256;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t pv = (uint32_t)(g * 256);
        // run containing pv: last s with pvstart[s] <= pv
        uint32_t lo = 0, hi = nrun;
        while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (pvstart[mid] <= pv) lo = mid + 1; else hi = mid; }
        const uint32_t s0 = lo - 1;
        // k-slot of an output = its position among the group's outputs + a per-run constant ("delta"): the group's
        // first run continues at k0; a run that starts inside the group at k-slot pkstart has delta = pkstart - (outputs
        // of the group before its head) = pkstart - X[pvstart] + X[pv]
        GroupRec r;
        r.s = s0; r.k0 = pkstart[s0] + (X[pv] - X[pvstart[s0]]);
        for (int i = 0; i < 6; i++) r.k[i] = (s0 + 1 + i < nrun) ? pkstart[s0 + 1 + i] - X[pvstart[s0 + 1 + i]] + X[pv] : 0;
        G[g] = r;
    }
}

// The WIDE build's record: dword 0 = k0, 1..4 = the constants of the first FOUR run heads, 5 = the run the group starts in, 6 / 7 = the
// mask of the lanes (quads) whose first entry begins a run -- LCOL has no bit left for it (15 column bits + the end bit)
__global__ void k_group_table_wide(const uint32_t *__restrict__ pvstart, const uint32_t *__restrict__ pkstart, const uint32_t *__restrict__ X,
                                   uint32_t nrun, uint32_t np, GroupRec *__restrict__ G) {
    const uint64_t ngroups = ((uint64_t)np + 255) / 256;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < ngroups; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t pv = (uint32_t)(g * 256);
        uint32_t lo = 0, hi = nrun;
        while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (pvstart[mid] <= pv) lo = mid + 1; else hi = mid; }
        const uint32_t s0 = lo - 1;
        uint32_t w[8];
        w[0] = pkstart[s0] + (X[pv] - X[pvstart[s0]]);
        for (int i = 0; i < 4; i++) w[1 + i] = (s0 + 1 + i < nrun) ? pkstart[s0 + 1 + i] - X[pvstart[s0 + 1 + i]] + X[pv] : 0;
        w[5] = s0;
        uint64_t hm = 0;
        for (uint32_t r = s0 + 1; r < nrun && pvstart[r] < pv + 256u; r++) hm |= 1ull << ((pvstart[r] - pv) >> 2);   // runs start at multiples of four padded entries
        w[6] = (uint32_t)hm; w[7] = (uint32_t)(hm >> 32);
        GroupRec rec; rec.k0 = w[0]; for (int i = 0; i < 6; i++) rec.k[i] = w[1 + i]; rec.s = w[7];
        G[g] = rec;
    }
}

// KSTART[s] = pkstart[s] - X[pvstart[s]]: the delta of run s is KSTART[s] + X[group start] (groups with 7+ run heads)
__global__ void k_kstart(const uint32_t *__restrict__ pvstart, const uint32_t *__restrict__ pkstart, const uint32_t *__restrict__ X,
                         uint32_t nrun, uint32_t *__restrict__ KSTART) {
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nrun; r += gridDim.x * blockDim.x) KSTART[r] = pkstart[r] - X[pvstart[r]];
}
// chunk range of every phase-2 work item: runs are sorted by (bin, chunk) in the k-order, kscan = their k-starts
__global__ void k_work_chunks(BinWork *__restrict__ work, uint32_t nwork, const uint32_t *__restrict__ kscan, const uint32_t *__restrict__ order,
                              const uint32_t *__restrict__ runkey, uint32_t nrun, int binbits) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nwork; i += gridDim.x * blockDim.x) {
        BinWork w = work[i];
        // last run starting at or before k: upper_bound - 1
        auto run_at = [&](uint32_t k) -> uint32_t {
            uint32_t lo = 0, hi = nrun;
            while (lo < hi) { uint32_t mid = lo + ((hi - lo) >> 1); if (kscan[mid] <= k) lo = mid + 1; else hi = mid; }
            return lo - 1;
        };
        w.c_lo = runkey[order[run_at(w.k0)]] >> binbits;
        w.c_hi = runkey[order[run_at(w.k1 - 1)]] >> binbits;
        work[i] = w;
    }
}
// inclusive prefix of the per-chunk activity flags written by phase 1 (one block; nchunks is a few thousand)
__global__ void k_active_prefix(const uint32_t *__restrict__ active, uint32_t nchunks, uint32_t *__restrict__ prefix) {
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) { carry = 0; prefix[0] = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < nchunks; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v = (i < nchunks) ? active[i] : 0;
        // block-wide inclusive scan through LDS (simple Hillis-Steele on 1024 threads)
        __shared__ uint32_t buf[1024];
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < blockDim.x; d <<= 1) {
            uint32_t t = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nchunks) prefix[i + 1] = carry + buf[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry += buf[threadIdx.x];
        __syncthreads();
    }
}

// ---- window activity (min programs): how many stored entries the ACTIVE columns of every window hold
__global__ void k_slot_degrees(const uint32_t *__restrict__ JA, const uint32_t *__restrict__ xcol, uint32_t x_len, uint32_t *__restrict__ xdeg) {
    for (uint32_t sl = blockIdx.x * blockDim.x + threadIdx.x; sl < x_len; sl += gridDim.x * blockDim.x) {
        const uint32_t c = xcol ? xcol[sl] : sl;
        xdeg[sl] = c == 0xFFFFFFFFu ? 0u : JA[c + 1] - JA[c];
    }
}
__global__ void k_win_total(const uint32_t *__restrict__ wcount, uint32_t nwin, uint32_t ncls, uint32_t *__restrict__ win_entries) {
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nwin; q += gridDim.x * blockDim.x) {
        uint32_t n = 0;
        for (uint32_t k = 0; k < ncls; k++) n += wcount[k * nwin + q];
        win_entries[q] = n;
    }
}
// one pass over x: a wave's 64 slots lie in one window almost always -> one atomic per wave
__global__ void __launch_bounds__(256) k_window_activity(const uint32_t *__restrict__ x, const uint32_t *__restrict__ xdeg, uint32_t x_len, WinGeom geom,
                                                         unsigned long long *__restrict__ win_act) {
    const uint32_t n64 = (x_len + 63) & ~63u;
    for (uint32_t sl = blockIdx.x * blockDim.x + threadIdx.x; sl < n64; sl += gridDim.x * blockDim.x) {
        const bool in = sl < x_len;
        unsigned long long d = (in && x[sl] != GT_INF) ? xdeg[sl] : 0u;
        const uint32_t q = in ? win_of(geom, sl) : 0xFFFFFFFFu;
        const uint32_t q0 = __builtin_amdgcn_readfirstlane(q);
        if (__all(q == q0 || !in)) {
            for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o);
            if ((threadIdx.x & 63) == 0 && d && q0 != 0xFFFFFFFFu) atomicAdd(&win_act[q0], d);
        } else if (d) atomicAdd(&win_act[q], d);
    }
}

// HYBRID pass, step 1 -- one workgroup per CANDIDATE window (the few windows that hold at least 1/256 of all entries each: the hub
// windows, where the saving is): stages the window's messages and per-slot entry counts, adds up the entries of the ACTIVE
// columns, and decides on the spot: none -> mode 2; at most 1/F of the window's entries -> mode 1, and the active columns are
// appended to the (column, message, entries) list of the column-driven kernels (one reservation per workgroup); else mode 0 (stream).
// (The first version counted all windows in a pass over x, chose in a second kernel, listed in a third: ~0.25 ms of small launches
// per pass, more than it saved -- profiles/r04/ab_hybrid_pass_first_version.txt.)
__global__ void __launch_bounds__(1024) k_hybrid_windows(const uint32_t *__restrict__ cand, const uint32_t *__restrict__ x, const uint32_t *__restrict__ xcol,
                                                         const uint32_t *__restrict__ xdeg, uint32_t x_len, WinGeom geom, const uint32_t *__restrict__ win_entries,
                                                         unsigned long long F, uint8_t *__restrict__ win_mode, unsigned int *__restrict__ cursor, uint32_t cap,
                                                         uint32_t *__restrict__ col, uint32_t *__restrict__ val, uint32_t *__restrict__ deg,
                                                         unsigned long long *__restrict__ stat) {
    constexpr int PER = (WS + 1023) / 1024;
    __shared__ unsigned long long part[16];
    __shared__ unsigned int wcnt[16];
    __shared__ unsigned long long total_s;
    __shared__ unsigned int base_s;
    const uint32_t q = cand[blockIdx.x];
    const uint32_t col0 = win_col0(geom, q), wlim = q < geom.ndw ? W : WS;
    const uint32_t wn = x_len - col0 < wlim ? x_len - col0 : wlim;
    uint32_t v[PER], d[PER];
    unsigned long long mine = 0; unsigned int ncols = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t j = threadIdx.x + i * 1024;
        v[i] = GT_INF; d[i] = 0;
        if (j < wn) { v[i] = x[col0 + j]; d[i] = xdeg[col0 + j]; }
        if (v[i] != GT_INF && d[i]) { mine += d[i]; ncols++; } else d[i] = 0;
    }
    unsigned long long m2 = mine;
    for (int o = 32; o > 0; o >>= 1) m2 += __shfl_down(m2, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m2;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < 16; w++) t += part[w]; total_s = t; }
    __syncthreads();
    const unsigned long long total = total_s;
    const uint8_t mode = total == 0 ? 2 : (total * F <= win_entries[q] ? 1 : 0);
    if (threadIdx.x == 0) {
        win_mode[q] = mode;
        if (stat && mode == 1) { atomicAdd(&stat[1], total); atomicAdd(&stat[2], (unsigned long long)win_entries[q]); atomicAdd(&stat[3], 1ull); }
        if (stat && blockIdx.x == 0) atomicAdd(&stat[0], 1ull);
    }
    if (mode != 1) return;
    // the active columns of this window -> the list: exclusive scan of the per-thread counts (wave scan + wave totals)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned int incl = ncols;
    for (int o = 1; o < 64; o <<= 1) { const unsigned int t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
    if (lane == 63) wcnt[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned int t = 0; for (int w = 0; w < 16; w++) { const unsigned int n = wcnt[w]; wcnt[w] = t; t += n; } base_s = t ? atomicAdd(cursor, t) : 0u; }
    __syncthreads();
    uint32_t o = base_s + wcnt[wave] + (incl - ncols);
#pragma unroll
    for (int i = 0; i < PER; i++)
        if (d[i]) { const uint32_t sl = col0 + threadIdx.x + i * 1024; if (o < cap) { col[o] = xcol ? xcol[sl] : sl; val[o] = v[i]; deg[o] = d[i]; } o++; }
}
// eight lanes per listed column; the long columns (hubs: 10^4 .. 10^6 entries) are handed to k_hybrid_long
template <bool WEIGHTED>
__global__ void __launch_bounds__(256) k_hybrid_cols(const uint32_t *__restrict__ col, const uint32_t *__restrict__ val, const uint32_t *__restrict__ deg,
                                                     const unsigned int *__restrict__ n_dev, uint32_t cap, uint32_t big, const uint32_t *__restrict__ JA, const uint32_t *__restrict__ IA,
                                                     const uint32_t *__restrict__ A, uint32_t *__restrict__ y, uint32_t *__restrict__ long_list, unsigned int *__restrict__ long_n) {
    constexpr uint32_t LPC = 8, GPB = 256 / LPC;
    const uint32_t n = min(*n_dev, cap), sub = threadIdx.x & (LPC - 1);
    for (uint32_t gi = blockIdx.x * GPB + threadIdx.x / LPC; gi < n; gi += gridDim.x * GPB) {
        const uint32_t d = deg[gi];
        if (d > big) { if (sub == 0) long_list[atomicAdd(long_n, 1u)] = gi; continue; }
        const uint32_t e0 = JA[col[gi]], m0 = val[gi];
        for (uint32_t k = sub; k < d; k += LPC) {
            const uint32_t r = IA[e0 + k];
            const uint32_t m = WEIGHTED ? m0 + A[e0 + k] : m0;
            if (m < y[r]) atomicMin(&y[r], m);
        }
    }
}
template <bool WEIGHTED>
__global__ void __launch_bounds__(256) k_hybrid_long(const uint32_t *__restrict__ long_list, const unsigned int *__restrict__ long_n, const uint32_t *__restrict__ col,
                                                     const uint32_t *__restrict__ val, const uint32_t *__restrict__ deg, const uint32_t *__restrict__ JA,
                                                     const uint32_t *__restrict__ IA, const uint32_t *__restrict__ A, uint32_t *__restrict__ y) {
    // a long column is cut into pieces of 4096 entries, a workgroup per piece (blockIdx.y = the piece's position among gridDim.y)
    const uint32_t n = *long_n;
    for (uint32_t li = blockIdx.x; li < n; li += gridDim.x) {
        const uint32_t gi = long_list[li], d = deg[gi], e0 = JA[col[gi]], m0 = val[gi];
        for (uint32_t k = blockIdx.y * 256 + threadIdx.x; k < d; k += gridDim.y * 256) {
            const uint32_t r = IA[e0 + k];
            const uint32_t m = WEIGHTED ? m0 + A[e0 + k] : m0;
            if (m < y[r]) atomicMin(&y[r], m);
        }
    }
}

#ifdef GT_EXP_TRACE   // timing experiment: per-workgroup start / end (100-MHz wall clock), what it worked on and where it ran (tools/p1_trace.py)
__device__ unsigned long long gt_trace_p1[4 * 16384], gt_trace_p2[4 * 16384], gt_trace_w[16 * 8192];   // (gt_trace_w: the end of each of a phase-1 workgroup's 16 waves)
__device__ __forceinline__ unsigned long long gt_trace_where() {
    return ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);   // XCC_ID, HW_ID
}
#endif
// ------------------------------------------------------------------ phase 1
// T  = type of y and of the LDS accumulators (double or uint32_t)
// TV = type of the value stream VAL and of the LDS message window: T, or float for the
//      "f32 messages" PageRank variant (messages rounded to f32, sums still accumulated in f64)
// TX = type of the message vector x in HBM (T, or float when the program keeps x itself in f32)
template <class T, class TV> struct Msg;
template <> struct Msg<double, double> { static __device__ __forceinline__ double val(double x, uint32_t) { return x; } };
template <> struct Msg<double, float> { static __device__ __forceinline__ float val(float x, uint32_t) { return x; } };
template <> struct Msg<uint32_t, uint32_t> { static __device__ __forceinline__ uint32_t val(uint32_t x, uint32_t w) { return x == GT_INF ? GT_INF : x + w; } };

// Cache policy of the streams. tools/hbm_ceiling2.hip on this pool: a pure read stream runs at 6.1 TB/s with default-policy
// loads and at 7.1-7.2 TB/s with non-temporal ones (LDS-DMA nt: 7.0-7.3); writes 5.4-5.8 TB/s either way, copies 5.0-5.4.
// In the kernels (A/B on one box, three rounds, profiles/r03/ab_nt.txt): nt loads of VAL / LROW make phase 2 ~8 % faster
// (0.67-0.72 -> 0.60-0.66 ms); nt loads of LCOL / WT / group records change nothing in phase 1 (0.94-1.0 ms both ways); nt
// STORES of the value stream cost phase 1 20 % (0.95 -> 1.19 ms) for 5 % in phase 2. Hence 2.
#ifndef GT_PB_NT
#define GT_PB_NT 2   // bit 0: loads of phase 1, bit 1: loads of phase 2, bit 2: stores of the value stream (A/B below)
#endif
constexpr bool NT_P1 = (GT_PB_NT & 1) != 0, NT_P2 = (GT_PB_NT & 2) != 0, NT_ST = (GT_PB_NT & 4) != 0, NT_EPI = (GT_PB_NT & 8) != 0;   // bit 3: the row -> slot map and the degrees in phase 2's flush
typedef uint32_t gt_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t gt_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT, class S> __device__ __forceinline__ S ld_stream(const S *__restrict__ p) {
    if constexpr (!NT) return *p;
    else if constexpr (sizeof(S) == 4) { const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p)); return __builtin_bit_cast(S, v); }
    else if constexpr (sizeof(S) == 8) { const gt_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const gt_u32x2 *>(p)); return __builtin_bit_cast(S, v); }
    else if constexpr (sizeof(S) == 16) { const gt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const gt_u32x4 *>(p)); return __builtin_bit_cast(S, v); }
    else {
        static_assert(sizeof(S) == 32, "stream element of 4, 8, 16 or 32 bytes");
        struct P2 { gt_u32x4 a, b; } v;
        v.a = __builtin_nontemporal_load(reinterpret_cast<const gt_u32x4 *>(p)); v.b = __builtin_nontemporal_load(reinterpret_cast<const gt_u32x4 *>(p) + 1);
        return __builtin_bit_cast(S, v);
    }
}
template <class S> __device__ __forceinline__ void st_stream(S *__restrict__ p, const S &x) {
    if constexpr (!NT_ST) *p = x;
    else if constexpr (sizeof(S) == 4) __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, x), reinterpret_cast<uint32_t *>(p));
    else if constexpr (sizeof(S) == 8) __builtin_nontemporal_store(__builtin_bit_cast(gt_u32x2, x), reinterpret_cast<gt_u32x2 *>(p));
    else if constexpr (sizeof(S) == 16) __builtin_nontemporal_store(__builtin_bit_cast(gt_u32x4, x), reinterpret_cast<gt_u32x4 *>(p));
    else {
        static_assert(sizeof(S) == 32, "stream element of 4, 8, 16 or 32 bytes");
        struct P2 { gt_u32x4 a, b; };
        const P2 v = __builtin_bit_cast(P2, x);
        __builtin_nontemporal_store(v.a, reinterpret_cast<gt_u32x4 *>(p)); __builtin_nontemporal_store(v.b, reinterpret_cast<gt_u32x4 *>(p) + 1);
    }
}

template <class TV> struct alignas(sizeof(TV) * 4 > 16 ? 16 : sizeof(TV) * 4) V4 { TV a[4]; };
struct alignas(8) C4 { uint16_t c[4]; };
template <class WTy> struct alignas(sizeof(WTy) * 4) WQ { WTy w[4]; };   // the weights of a quad: 4, 8 or 16 bytes
__global__ void k_max_u32(const uint32_t *__restrict__ a, uint64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m = a[i] > m ? a[i] : m;
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_down(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// the window of x a chunk works on -> LDS; returns false when the chunk can be skipped (min programs, no active column)
template <class TV, class TX, bool IS_MIN, int THREADS, uint32_t WIN>
__device__ __forceinline__ bool stage_window(TV *__restrict__ xwin, const TX *__restrict__ x, uint32_t col0, uint32_t wn, uint32_t *__restrict__ chunk_active, uint32_t c) {
    // all loads of a lane in flight together
    constexpr int PER = (WIN + THREADS - 1) / THREADS;
    TX t[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) { const uint32_t j = threadIdx.x + i * THREADS; t[i] = (j < wn) ? x[col0 + j] : (IS_MIN ? (TX)GT_INF : TX(0)); }   // beyond the window: the neutral message (slot W of a dense window is what pad entries read)
#pragma unroll
    for (int i = 0; i < PER; i++) { const uint32_t j = threadIdx.x + i * THREADS; if (j < WIN) xwin[j] = (TV)t[i]; }
    if constexpr (IS_MIN) {
        // Activity filtering (the reference's sparse path, vp:754-784, 1475-1489, at window granularity): a chunk
        // whose window holds no active column (every message is infinity()) produces only neutral values. With
        // chunk_active != nullptr the caller guarantees that VAL already holds, in this chunk's slots, either
        // the neutral value or messages of an earlier iteration of the SAME program -- harmless to re-combine
        // because y is a running min (vp:1785) -- so the chunk is skipped entirely.
        if (chunk_active) {
            int any = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) { const uint32_t j = threadIdx.x + i * THREADS; any |= (j < wn && t[i] != (TX)GT_INF); }
            any = __syncthreads_or(any);
            if (threadIdx.x == 0) chunk_active[c] = any ? 1u : 0u;
            if (!any) return false;
        }
    }
    return true;
}

// k-slot constant ("delta") of the lane's run: dword 0 of the group record for the group's first run, dword nh for the run
// that starts at the nh-th run head of the group (nh = 1..6; the build folds "outputs of the group before that head" into
// it). A quad never straddles runs.
__device__ __forceinline__ uint32_t run_delta(uint32_t lane, bool head, uint32_t gw, const uint32_t *__restrict__ KSTART) {
    auto below = [](uint64_t m) -> uint32_t { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)); };
    const uint64_t Hb = __ballot(head);
    const uint32_t nh = below(Hb) + (head ? 1u : 0u);                       // run heads at or before this lane
    uint32_t delta = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nh < 7 ? nh : 0u) << 2), (int)gw);
    if (__popcll((unsigned long long)Hb) >= 7) {
        // rare: seven or more run heads in one 256-entry group. Walks the heads beyond the sixth with wave-uniform
        // (scalar) loads: a vector load here would put a vmcnt(0) wait -- prefetches and stores included -- into
        // every group of the common path. delta(run s) = KSTART[s] + X[group start], X[group start] = k0 - KSTART[s0].
        const uint32_t s0 = __builtin_amdgcn_readlane(gw, 7);
        const uint32_t xg = __builtin_amdgcn_readlane(gw, 0) - KSTART[s0];
        uint64_t Hm = Hb;
        for (int i = 0; i < 6; i++) Hm &= Hm - 1;
        for (uint32_t i = 7; Hm; i++) {
            const uint32_t hlane = (uint32_t)__ffsll((unsigned long long)Hm) - 1;
            Hm &= Hm - 1;
            const uint32_t ks = KSTART[s0 + i] + xg;
            if (lane >= hlane) delta = ks;
        }
    }
    return delta;
}

// the same for the WIDE build's group record: the head lanes are a mask (dwords 6, 7), four inline constants (dwords 1..4), the
// group's first run in dword 5
__device__ __forceinline__ uint32_t run_delta_wide(uint32_t lane, uint64_t Hb, uint32_t gw, const uint32_t *__restrict__ KSTART) {
    const uint32_t nh = __builtin_amdgcn_mbcnt_hi((uint32_t)(Hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Hb, 0)) + (uint32_t)((Hb >> lane) & 1ull);   // run heads at or before this lane
    uint32_t delta = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((nh < 5 ? nh : 0u) << 2), (int)gw);
    if (__popcll((unsigned long long)Hb) >= 5) {
        const uint32_t s0 = __builtin_amdgcn_readlane(gw, 5);
        const uint32_t xg = __builtin_amdgcn_readlane(gw, 0) - KSTART[s0];
        uint64_t Hm = Hb;
        for (int i = 0; i < 4; i++) Hm &= Hm - 1;
        for (uint32_t i = 5; Hm; i++) {
            const uint32_t hlane = (uint32_t)__ffsll((unsigned long long)Hm) - 1;
            Hm &= Hm - 1;
            const uint32_t ks = KSTART[s0 + i] + xg;
            if (lane >= hlane) delta = ks;
        }
    }
    return delta;
}

// ---- wave-wide DPP scans (no LDS): lanes are the 64 quads of a 256-entry group, in order
// source lane = lane - d inside a row of 16 (d = 1, 2, 4, 8), lane 15 of the previous row (0x142) / lane 31 (0x143) for the
// row-crossing steps; lanes without a source read `fill`
template <int CTRL, int ROW_MASK, class TV> __device__ __forceinline__ TV dpp_get(TV v, TV fill) {
    if constexpr (sizeof(TV) == 8) {
        const unsigned long long u = __double_as_longlong((double)v), f = __double_as_longlong((double)fill);
        const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
        const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
        return (TV)__longlong_as_double(((unsigned long long)hi << 32) | lo);
    } else if constexpr (std::is_same<TV, float>::value) {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
    } else {
        return (TV)__builtin_amdgcn_update_dpp((int)fill, (int)v, CTRL, ROW_MASK, 0xf, false);
    }
}
// Segmented EXCLUSIVE scan over the wave: lane L gets the combination of t over the lanes below it back to (and including) the
// nearest lane whose flag is set -- what the quads before it leave open of the output stretch that is running when lane L
// begins. The flags are ONE wave-uniform 64-bit mask (F: lanes that hold an end): the flag half of the scan runs on the scalar
// unit (a shift, an AND with the row pattern and an OR per step) and a step costs two vector instructions -- the DPP combine and
// a select on ~F through an SGPR pair. (With per-lane flags a step was five: two DPP moves, compare, select, OR; phase 1 is
// bound by instruction issue: 526 M vector instructions per SpMV of R-MAT-26 = 0.86 ms of its 0.93, profiles/r03/. Later in round 3,
// with 20 / 40 extra dependent SCALAR instructions per group phase 1 lost 2 % / 11 %, with as many vector ones 1 % / 5.5 %
// (GT_P1_SBURN / GT_P1_VBURN below, profiles/r03/ab_phase2_flush_and_issue_burn.txt): neither port is saturated any more, and
// keeping the flags complemented to save the s_not of every step changed nothing.)
// (the four row masks of the in-row steps arrive in scalar register pairs the kernel fills ONCE -- `m`: a 64-bit AND instead of two
// 32-bit ones with literals in every step)
struct RowMasks { uint64_t m1, m2, m4, m8; };
template <class TV, bool IS_MIN> __device__ __forceinline__ TV wave_carry_masked(TV t, uint64_t F, const RowMasks &m) {
    const TV neutral = IS_MIN ? (TV)GT_INF : (TV)0;
    auto comb = [](TV a, TV b) -> TV { if constexpr (IS_MIN) return a < b ? a : b; else return a + b; };
    TV v = t;
#define GT_MSEG_STEP(CTRL, RM, NEXT_F)                                             \
    {                                                                              \
        const TV vs = dpp_get<CTRL, RM, TV>(v, neutral);                           \
        v = __builtin_amdgcn_inverse_ballot_w64(F) ? v : comb(v, vs);              \
        F |= (NEXT_F);                                                             \
    }
    GT_MSEG_STEP(0x111, 0xf, (F << 1) & m.m1)   // row_shr:1 .. 8: lane l takes lane l - d of its row of 16 (masks 0xFFFE.., 0xFFFC.., 0xFFF0.., 0xFF00.. per row)
    GT_MSEG_STEP(0x112, 0xf, (F << 2) & m.m2)
    GT_MSEG_STEP(0x114, 0xf, (F << 4) & m.m4)
    GT_MSEG_STEP(0x118, 0xf, (F << 8) & m.m8)
    GT_MSEG_STEP(0x142, 0xa, ((F >> 47) & 1ull) ? 0xFFFF000000000000ull : 0ull)   // row_bcast:15 into rows 1, 3 (the last step writes rows 2, 3: of the flags only "row 3 is closed by a flag of row 2" is still needed)
    GT_MSEG_STEP(0x143, 0xc, 0ull)                                // row_bcast:31 into rows 2, 3 (the flags are not needed afterwards)
#undef GT_MSEG_STEP
    return dpp_get<0x138, 0xf, TV>(v, neutral);   // wave_shr:1: the inclusive result of the lane below
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t m, uint32_t acc) {   // acc + the set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, acc));
}

// ---- dense windows: pre-aggregation over whole 256-entry groups
// A lane holds a quad of consecutive entries. Stretches of one row are summed inside the quad in registers; what a quad leaves
// open flows to the lane that holds the stretch's end through a segmented DPP scan (no LDS, no atomics: on gfx950 an LDS
// float atomic costs 192+ cycles per wave instruction, and the conflict-ridden f64 ones kept the LDS 73 % busy,
// tools/lds_atomic_bench.hip); every lane then stores its outputs itself -- the k-slots of a wave's
// outputs are consecutive, so the 4 predicated stores of a group fill the same few cache lines.
template <class T, class TV, class TX, bool WEIGHTED, bool IS_MIN, class WTy, bool WIDE = false>
__global__ void __launch_bounds__(P1_THREADS) k_pb_scatter(const uint32_t *__restrict__ cv0, const uint32_t *__restrict__ cv1,
                                                           const uint32_t *__restrict__ ccol0, uint32_t ncols,
                                                           const C4 *__restrict__ LCOL4, const WQ<WTy> *__restrict__ WT4,
                                                           const uint32_t *__restrict__ KSTART, const GroupRec *__restrict__ G,
                                                           const TX *__restrict__ x, TV *__restrict__ VAL, uint32_t *__restrict__ chunk_active,
                                                           const gt_u32x4 *__restrict__ ldesc, uint32_t chunk0, uint32_t dense_end,
                                                           const uint8_t *__restrict__ win_mode, uint32_t *__restrict__ queue, uint32_t nlaunch) {
    static_assert(W + 1 == WS, "dense and sparse chunks share one LDS window");
    // the WIDE build (gt_pb::wide): windows of 2 W / 2 WS slots, 15 column bits, run heads as a mask in the group record
    constexpr uint32_t WIN = WIDE ? 2 * WS : WS, WD = WIDE ? 2 * W : W;
    constexpr uint16_t CM = WIDE ? (uint16_t)0x7FFF : COLMASK;
    constexpr uint32_t NHI = WIDE ? 5u : 7u, SIDX = WIDE ? 5u : 7u;   // inline run constants: heads 1 .. NHI - 1; the dword that holds the group's first run
    static_assert(!WIDE || sizeof(TV) == 4, "the wide window is 128 KiB of 4-byte messages");
    __shared__ TV xwin[WIN];
    // the outputs of a wave's 256-entry group, dense, before they leave in coalesced stores: four predicated dword stores per lane
    // straight to VAL (each covering a strided subset of the group's ~1 KiB of slots) were bound by the write REQUESTS they
    // make, not by bytes or instructions -- a second DPP scan added to the kernel cost nothing, staging the outputs through
    // this row and storing 64 consecutive slots per instruction gave 8 % (1.69 -> 1.55 ms per SpMV)
    // The min programs keep the direct stores: the staged form is 10-20 % SLOWER for them (CC R-MAT-26: 3.39 -> 4.06 ms for a
    // full pass), measured by A/B on one box; neither occupancy nor the number of run heads explains it.
#ifndef GT_P1_STAGE_MIN
#define GT_P1_STAGE_MIN 0
#endif
#ifndef GT_P1_FIRST_LAST
#define GT_P1_FIRST_LAST 1   // (-1.8 % of phase 1, ten rounds of A/B on one box: profiles/r04/ab_first_last_and_persistent.txt)
#endif
#ifndef GT_P1_PERSIST_DEFAULT
#define GT_P1_PERSIST_DEFAULT 1
#endif
    constexpr bool STAGED = !IS_MIN || GT_P1_STAGE_MIN != 0;
    // DYNAMIC TRIPS (the wide kernel): a wave takes its next trip of U groups from a counter of the workgroup instead of every 16th
    // one. The 16 waves of a workgroup are four per SIMD, and a SIMD's arbiter prefers its OLDEST wave: with equal shares waves 0-3
    // are done at 75 % of the chunk's duration, 4-7 at 81 %, 8-11 at 89 % (tools/p1_trace.py, workgroup_timeline_waves.txt) -- 15 % of
    // the wave-time of phase 1 idles in that drain, once per chunk. The counter lives in a seventeenth staging row (the wide kernel has
    // 16 KiB of LDS to spare; the 64-KiB-window and f64 kernels have none).
#ifndef GT_P1_DYN_TRIPS
#define GT_P1_DYN_TRIPS 1
#endif
    constexpr bool DYN = GT_P1_DYN_TRIPS != 0 && WIDE && STAGED;
    __shared__ TV stage[STAGED ? P1_THREADS / 64 + (DYN ? 1 : 0) : 1][STAGED ? 256 : 8];
#ifndef GT_P1_PERSIST_ALL
#define GT_P1_PERSIST_ALL 0
#endif
    // PERSISTENT form (queue != nullptr; below): compiled in only where it is used -- see the loop at the end of the kernel
    constexpr bool CAN_PERSIST = GT_P1_PERSIST_ALL != 0 || (!IS_MIN && (WIDE || sizeof(TV) == 8));
    RowMasks rowm{0xFFFEFFFEFFFEFFFEull, 0xFFFCFFFCFFFCFFFCull, 0xFFF0FFF0FFF0FFF0ull, 0xFF00FF00FF00FF00ull};
    asm volatile("" : "+s"(rowm.m1), "+s"(rowm.m2), "+s"(rowm.m4), "+s"(rowm.m8));   // (opaque to the compiler: they stay in scalar registers, see wave_carry_masked)
    // one chunk: position bi of the launch = chunk c (largest chunks first, see gt_pb_build), its entry range [v0, v1) and first column
    auto chunk = [&](const uint32_t bi, const uint32_t c, const uint32_t v0c, const uint32_t v1c, const uint32_t col0) {
    const uint32_t q0c = v0c >> 2, q1c = v1c >> 2;   // the chunk's quad range (multiples of 64)
    // ONE launch for both kinds of chunks (dense ones first, largest first; the light sparse ones fill the tail): two launches
    // cost a drain of the 64-KiB workgroups in between
    const bool sparse = col0 >= dense_end;
#ifdef GT_EXP_TRACE
    uint32_t trace_out = 0;
    if (threadIdx.x == 0 && bi < 8192) {
        gt_trace_p1[8 * bi] = wall_clock64(); gt_trace_p1[8 * bi + 2] = ((unsigned long long)c << 32) | (q1c - q0c);
        gt_trace_p1[8 * bi + 3] = gt_trace_where();
    }
#endif
    if constexpr (IS_MIN) {
        // hybrid pass (pb_run): a window whose few active columns went to the column-driven SpMSpV -- or that has none -- is left out
        // here without even staging it; its value-stream slots keep older messages of the same program (y is a running min)
        if (win_mode) {
            const uint32_t q = sparse ? (dense_end + WD - 1) / WD + (col0 - dense_end) / WIN : col0 / WD;
            if (win_mode[q] != 0) { if (threadIdx.x == 0 && chunk_active) chunk_active[c] = 0u; return; }
        }
    }
    const uint32_t wlim = sparse ? WIN : WD;
    const uint32_t wn = (ncols - col0 < wlim) ? ncols - col0 : wlim;
    const TV neutral = IS_MIN ? (TV)GT_INF : (TV)0;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // The first trip's loads go out BEFORE the window is staged for the plus semirings (-0.5 to -1 % of phase 1, A/B of three rounds,
    // profiles/r03/ab_prefetch_before_staging.txt); the min programs stage first: a chunk their activity filter skips should not have
    // fetched 48 KiB of its entry stream
    constexpr bool PREFETCH = !IS_MIN;
    if constexpr (!PREFETCH) {
        if (!stage_window<TV, TX, IS_MIN, P1_THREADS, WIN>(xwin, x, col0, wn, chunk_active, c)) return;
        __syncthreads();
    }
    constexpr uint32_t NW = P1_THREADS / 64;
#ifndef GT_P1_U_WIDE
#define GT_P1_U_WIDE 6
#endif
    // 256-entry groups per trip and wave. The wide kernel (one workgroup per CU, trips handed out dynamically): 6 -- phase 1 0.905 -> 0.886 ms
    // against 4, 2: 0.978, 3: 0.927 (three rounds, profiles/r04/ab_trip_size.txt); the others 4 (the two-per-CU kernels must stay at
    // <= 64 VGPRs; f64 messages: 4 and 6 run alike)
    constexpr int U = WIDE ? GT_P1_U_WIDE : 4;
    const uint32_t gend = q1c >> 6;   // chunk ranges are multiples of 256 entries = 64 quads (k_align_chunks)
    const uint32_t *__restrict__ Gw = reinterpret_cast<const uint32_t *>(G);
    char *__restrict__ VALb = reinterpret_cast<char *>(VAL);
    // Software pipeline: the loads of trip t+1 are issued BEFORE the stores of trip t, so a wave does not wait
    // on its own store acknowledgements (vmcnt retires in order).
    C4 lc[U], nlc[U]; uint32_t gw[U], ngw[U]; WQ<WTy> w[U], nw[U];
    auto issue_loads = [&](uint32_t g0, C4 (&olc)[U], uint32_t (&ogw)[U], WQ<WTy> (&ow)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t g = (g0 + u < gend) ? g0 + u : gend - 1;
            const uint32_t q = g * 64 + lane;
            olc[u] = ld_stream<NT_P1>(LCOL4 + q);                        // 8 B/lane
            ogw[u] = ld_stream<NT_P1>(Gw + ((uint64_t)g * 8 + (lane & 7)));   // lane i holds dword i & 7 of the 32-byte group record
            if constexpr (WEIGHTED) ow[u] = ld_stream<NT_P1>(WT4 + q); else ow[u] = WQ<WTy>{{0, 0, 0, 0}};
        }
    };
    const uint32_t gb = q0c >> 6;
    uint32_t g0 = gb + wave * U;   // the first trip: the wave's own (its loads go out before the window is staged)
    if (g0 < gend) issue_loads(g0, lc, gw, w);
    uint32_t *tctr = reinterpret_cast<uint32_t *>(&stage[DYN ? NW : 0][0]);
    if constexpr (DYN) { if (threadIdx.x == 0) *tctr = NW; }   // (every wave of the previous chunk is behind a barrier; the first draw is behind the next one)
    if constexpr (PREFETCH) {
        if (!stage_window<TV, TX, IS_MIN, P1_THREADS, WIN>(xwin, x, col0, wn, chunk_active, c)) return;
        __syncthreads();
#ifdef GT_EXP_TRACE
        if (threadIdx.x == 0 && bi < 8192) gt_trace_p1[8 * bi + 4] = wall_clock64();   // the window is staged
#endif
    }
    // ROTATING PRIORITY (the kernels without LDS to spare for a trip counter): the waves of a SIMD take turns at its priority, one trip
    // each (s_setprio; waves 4s .. 4s+3 of a workgroup are the s-th waves of the four SIMDs) -- the arbiter then has no favourite.
    // f64 messages, R-MAT-26: waves done at 73 / 80 / 88 / 99 % of a chunk's duration -> 93-100 %, wave-time lost to the drain
    // 16.9 -> 8.0 %, phase 1 1.372 -> 1.312 ms (four rounds of A/B, profiles/r04/ab_rotating_priority.txt); f32 messages on the narrow
    // build R-MAT-24 +3 %, R-MAT-22 +-0; the min programs +-1 %.
    // (The counter in HBM instead -- one per workgroup, a wave asking for the trip AFTER the next one while it works on the current one
    // -- was built for these kernels and measured: f64 phase 1 1.30 -> 1.45 ms, the 64-KiB-window kernel +17 to +28 %: the returning
    // atomic sits in the same in-order vmcnt queue as the loads and stores of the software pipeline, and waiting for it drains the
    // queue. profiles/r04/ab_trip_counter_in_hbm_rejected.txt.)
#ifndef GT_P1_ROTATE_PRIO
#define GT_P1_ROTATE_PRIO 1
#endif
    uint32_t trip_no = wave >> 2;
    while (g0 < gend) {
        if constexpr (GT_P1_ROTATE_PRIO != 0 && !DYN) {
            switch (trip_no++ & 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        uint32_t gn;
        if constexpr (DYN) { uint32_t t = 0; if (lane == 0) t = atomicAdd(tctr, 1u); gn = gb + (uint32_t)__builtin_amdgcn_readfirstlane((int)t) * U; }
        else gn = g0 + NW * U;
        if (gn < gend) issue_loads(gn, nlc, ngw, nw);
        // (Round 4, wide build: gathering the messages of all the trip's groups first, and scanning the carries of all of them in one batch
        // -- N dependency chains side by side instead of one after the other -- were both built and measured: nothing / +6 % in phase 1,
        // profiles/r04/ab_wide_windows.txt. The 0.13 ms its scan costs in the open at 4 waves per SIMD is issue throughput, not latency.)
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (g0 + u >= gend) break;
            if (sparse) {   // one output per entry, its k-slot is its position: four values per lane, one 16-byte store
                uint32_t delta;
                if constexpr (WIDE) delta = run_delta_wide(lane, ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(gw[u], 7) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(gw[u], 6)   /* (readlane returns int: no sign extension into the upper half) */, gw[u], KSTART);
                else delta = run_delta(lane, lane != 0 && (lc[u].c[0] & HEAD) != 0, gw[u], KSTART);
                V4<TV> o;
#pragma unroll
                for (int j = 0; j < 4; j++) o.a[j] = Msg<T, TV>::val(xwin[lc[u].c[j] & CM], w[u].w[j]);
                st_stream(reinterpret_cast<V4<TV> *>(VAL + (lane * 4 + delta)), o);   // runs start at multiples of four slots in both orders
                continue;
            }
            // output-end bits of the quad, as four wave-uniform masks
            const bool e0 = (lc[u].c[0] & GEND) != 0, e1 = (lc[u].c[1] & GEND) != 0, e2 = (lc[u].c[2] & GEND) != 0, e3 = (lc[u].c[3] & GEND) != 0;
            const uint64_t E0 = __ballot(e0), E1 = __ballot(e1), E2 = __ballot(e2), E3 = __ballot(e3);
            const uint32_t n0 = e0 ? 1u : 0u, n1 = n0 + (e1 ? 1u : 0u), n2 = n1 + (e2 ? 1u : 0u);
            // dense index of the quad's first output inside the group = the ends in the lanes below (four mbcnt pairs; the DPP prefix
            // scan of the per-lane counts was twenty instructions)
            const uint32_t i0 = lanes_below(E3, lanes_below(E2, lanes_below(E1, lanes_below(E0, 0u))));
            uint64_t HbG = 0;   // the WIDE build: the head lanes come off the group record
            if constexpr (WIDE) HbG = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(gw[u], 7) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane(gw[u], 6)   /* (readlane returns int: no sign extension into the upper half) */;
            const bool head = WIDE ? ((HbG >> lane) & 1ull) != 0 : (lane != 0 && (lc[u].c[0] & HEAD) != 0);
#ifdef GT_EXP_P1_NO_GATHER   // timing experiment (wrong results): no LDS gathers, the column offsets stand in for the messages
            const TV v0 = (TV)(lc[u].c[0] & CM), v1 = (TV)(lc[u].c[1] & CM), v2 = (TV)(lc[u].c[2] & CM), v3 = (TV)(lc[u].c[3] & CM);
#else
            const TV v0 = Msg<T, TV>::val(xwin[lc[u].c[0] & CM], w[u].w[0]), v1 = Msg<T, TV>::val(xwin[lc[u].c[1] & CM], w[u].w[1]),
                     v2 = Msg<T, TV>::val(xwin[lc[u].c[2] & CM], w[u].w[2]), v3 = Msg<T, TV>::val(xwin[lc[u].c[3] & CM], w[u].w[3]);
#endif
            auto comb = [](TV a, TV b) -> TV { if constexpr (IS_MIN) return a < b ? a : b; else return a + b; };
            // a_i = value of the stretch that entry i belongs to, up to i, inside the quad
            const TV a1 = e0 ? v1 : comb(v0, v1);
            const TV a2 = e1 ? v2 : comb(a1, v2);
            const TV a3 = e2 ? v3 : comb(a2, v3);
            // what the quad leaves open for the lanes above: everything when it holds no end, else what follows its last end
#ifdef GT_EXP_P1_NO_SCAN   // timing experiment (wrong results): no segmented scan across the lanes
            const TV carry = e3 ? neutral : a3;
#else
            // (A wave-uniform shortcut for the groups whose 64 lanes ALL hold an end -- the windows of short runs; their carry is one DPP
            // move -- was measured: phase 1 +1.3 %, six rounds: the branch costs every group more than the scan it saves a few.)
            const TV carry = wave_carry_masked<TV, IS_MIN>(e3 ? neutral : a3, E0 | E1 | E2 | E3, rowm);
#endif
#if defined(GT_P1_SBURN) || defined(GT_P1_VBURN)   // experiment: extra dependent scalar / vector instructions per group (which issue port binds phase 1?)
            {
#ifdef GT_P1_SBURN
                uint32_t sb = wave;
#pragma unroll
                for (int b = 0; b < GT_P1_SBURN; b++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sb) : : "scc");
                if (sb == 0xdeadbeefu) stage[wave][0] = (TV)0;
#endif
#ifdef GT_P1_VBURN
                uint32_t vb = lane;
#pragma unroll
                for (int b = 0; b < GT_P1_VBURN; b++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(vb));
                if (vb == 0xdeadbeefu) stage[wave][0] = (TV)0;
#endif
            }
#endif
            // the quad's first end also closes what the lanes below left open
            if constexpr (!STAGED) {   // every lane stores its outputs itself: the k-slots of a wave's outputs are consecutive
                const uint32_t k0 = i0 + (WIDE ? run_delta_wide(lane, HbG, gw[u], KSTART) : run_delta(lane, head, gw[u], KSTART));   // + the constant of the lane's run
                if (e0) st_stream(reinterpret_cast<TV *>(VALb + (size_t)(k0 * (uint32_t)sizeof(TV))), comb(carry, v0));
                if (e1) st_stream(reinterpret_cast<TV *>(VALb + (size_t)((k0 + n0) * (uint32_t)sizeof(TV))), n0 ? a1 : comb(carry, a1));
                if (e2) st_stream(reinterpret_cast<TV *>(VALb + (size_t)((k0 + n1) * (uint32_t)sizeof(TV))), n1 ? a2 : comb(carry, a2));
                if (e3) st_stream(reinterpret_cast<TV *>(VALb + (size_t)((k0 + n2) * (uint32_t)sizeof(TV))), n2 ? a3 : comb(carry, a3));
                continue;
            }
            TV *__restrict__ st = stage[wave];
#if GT_P1_FIRST_LAST
            // The lane's FIRST output -- the one that also closes what the lanes below left open -- is written LAST, over whatever
            // the predicated stores put into its slot (LDS stores of one wave land in order): one select chain and one combine
            // instead of a combine, a compare and a select per output. Bit-identical (the same two operands are combined).
            {
                TV *__restrict__ sp = st + i0;
                const TV first = comb(carry, e0 ? v0 : (e1 ? a1 : (e2 ? a2 : a3)));
                if (e1) sp[n0] = a1;
                if (e2) sp[n1] = a2;
                if (e3) sp[n2] = a3;
                if (__builtin_amdgcn_inverse_ballot_w64(E0 | E1 | E2 | E3)) sp[0] = first;   // (the wave-uniform mask the scan already holds)
            }
#else
            if (e0) st[i0] = comb(carry, v0);
            if (e1) st[i0 + n0] = n0 ? a1 : comb(carry, a1);
            if (e2) st[i0 + n1] = n1 ? a2 : comb(carry, a2);
            if (e3) st[i0 + n2] = n2 ? a3 : comb(carry, a3);
#endif
            // out: output i of the group goes to k-slot i + (the constant of its run); 64 consecutive outputs per store. The
            // constants come off the group record with scalar reads: dword 0 for the run the group starts in, dword j for the run
            // of its j-th head (lane j of gw holds dword j), KSTART for heads beyond the sixth.
            const uint32_t nout = (uint32_t)(__popcll((unsigned long long)E0) + __popcll((unsigned long long)E1) + __popcll((unsigned long long)E2) + __popcll((unsigned long long)E3));
#ifdef GT_EXP_TRACE
            trace_out += nout;
#endif
            const uint32_t d0 = __builtin_amdgcn_readlane(gw[u], 0);
            const uint64_t Hb = WIDE ? HbG : __ballot(head);
#ifdef GT_EXP_P1_NO_STORES   // timing experiment (wrong results): the outputs are staged and not stored
            if (nout == 0xFFFFFFFFu)
#endif
            for (uint32_t i = lane; i < nout; i += 64) {
                uint32_t d = d0, j = 0;
                for (uint64_t Hm = Hb; Hm; Hm &= Hm - 1) {
                    const uint32_t hl = (uint32_t)__ffsll((unsigned long long)Hm) - 1;
                    j++;
                    const uint32_t S = __builtin_amdgcn_readlane(i0, hl);
                    uint32_t D;
                    if (j < NHI) D = __builtin_amdgcn_readlane(gw[u], j);
                    else { const uint32_t s0r = __builtin_amdgcn_readlane(gw[u], SIDX); D = KSTART[s0r + j] + (d0 - KSTART[s0r]); }   // delta(run s) = KSTART[s] + X[group start]
                    d = i >= S ? D : d;
                }
#ifdef GT_EXP_P1_SEQ_STORES   // timing experiment (wrong results): the same stores, each workgroup to ONE contiguous span of the value stream (the chunk's share)
                {
                    const uint32_t span0 = (uint32_t)(((uint64_t)q0c * 4u) / 3u) & ~63u;   // ~ the chunk's first slot if outputs were spread evenly (2.76 entries per slot)
                    const uint32_t off = ((g0 + u - (q0c >> 6)) * 96u + i) ;                // ~96 outputs per group, consecutive groups -> consecutive slots
                    st_stream(reinterpret_cast<TV *>(VALb + (size_t)((span0 + off) * (uint32_t)sizeof(TV))), (TV)st[i]);
                }
#else
                st_stream(reinterpret_cast<TV *>(VALb + (size_t)((i + d) * (uint32_t)sizeof(TV))), (TV)st[i]);
#endif
            }
        }
        g0 = gn;
#pragma unroll
        for (int u = 0; u < U; u++) { lc[u] = nlc[u]; gw[u] = ngw[u]; w[u] = nw[u]; }
    }
#ifdef GT_EXP_TRACE
    if (lane == 0 && bi < 8192) {
        atomicMax(&gt_trace_p1[8 * bi + 1], (unsigned long long)wall_clock64());
        atomicAdd(&gt_trace_p1[8 * bi + 3], (unsigned long long)(sparse ? 0u : trace_out) << 36);   // the chunk's outputs (a sparse chunk: one per entry)
        if (threadIdx.x == 0) gt_trace_p1[8 * bi + 5] = wall_clock64();   // wave 0 is done
        gt_trace_w[16 * bi + wave] = wall_clock64();
    }
#endif
    };
    // PERSISTENT form (queue != nullptr): as many workgroups as the chip holds at once, each drawing the next position of the launch
    // order from a counter -- no dispatch between a CU's chunks, and a free CU never waits behind a workgroup that the dispatcher
    // has promised to another XCD. Every workgroup overshoots exactly once; the one that draws the very last number of the launch
    // (nlaunch + gridDim.x - 1) puts the counter back to zero for the next launch that uses it.
    // ONE call site for both forms (the body inlined twice doubled the kernel's registers -- 39 -> 85 VGPRs in the 64-KiB-window
    // kernels, which then fit one workgroup per CU instead of two: every program on the narrow build lost 5-12 %).
    // (the kernel's LDS is spoken for to the last byte -- 2 x 80 KiB / 160 KiB per CU: what was drawn travels through the first
    // words of wave 0's staging row, which wave 0 writes again only behind the barrier that follows the staging of the window)
    volatile uint32_t *mb = reinterpret_cast<volatile uint32_t *>(&stage[0][0]);
    // Thread 0 draws a position and loads that chunk's description when the workgroup NEEDS it. (Drawing the next one at the START of
    // the current chunk -- the atomic and the dependent loads, ~3 us, in flight while the chunk runs -- was measured: phase 1
    // 0.968 -> 1.015 ms, six rounds of A/B. A workgroup that has committed itself one chunk ahead is no longer the first free one
    // when that chunk's turn comes: the list scheduling that persistence buys is lost again. Drawing when wave 0 enters its LAST
    // trip of the chunk, a few microseconds ahead: 0.961 -> 0.978 ms, six rounds -- profiles/r04/ab_first_last_and_persistent.txt (10).)
    // Compiled in only where it is used: the kernels that fit ONE workgroup per CU (the wide window, f64 messages), where nothing else
    // covers a dispatch gap. The two-per-CU kernels lose with it (f32 messages on the narrow build: R-MAT-22 460 -> 425 GTEPS, R-MAT-24
    // 652 -> 640, R-MAT-25 715 -> 690) and so do the min programs (see pb_run); GT_P1_PERSIST_ALL=1 compiles it into every
    // instantiation (A/B with GRAPHTAP_PB_PERSIST=1). A position's description is ONE 16-byte load (ldesc, in launch order).
    if constexpr (!CAN_PERSIST) {
        const gt_u32x4 d = ldesc[chunk0 + blockIdx.x];
        chunk(blockIdx.x, d.x, d.y, d.z, d.w);
    } else
    for (;;) {
        uint32_t bi, c, v0c, v1c, col0;
        if (queue) {
            __syncthreads();   // every wave is done with the window and with its staging row
            if (threadIdx.x == 0) {
#ifdef GT_EXP_TRACE
                const unsigned long long t_b1 = wall_clock64();   // every wave of the previous chunk has arrived
#endif
                const uint32_t nb = atomicAdd(queue, 1u);
                if (nb == nlaunch + gridDim.x - 1) atomicExch(queue, 0u);
                mb[0] = nb;
                if (nb < nlaunch) { const gt_u32x4 d = ldesc[chunk0 + nb]; mb[1] = d.x; mb[2] = d.y; mb[3] = d.z; mb[4] = d.w; }
#ifdef GT_EXP_TRACE
                if (nb < nlaunch && nb < 8192) { gt_trace_p1[8 * nb + 6] = t_b1; gt_trace_p1[8 * nb + 7] = wall_clock64(); }   // ... and the draw is back
#endif
            }
            __syncthreads();
            // (wave-uniform values: through readfirstlane into scalar registers, or every address of the chunk is computed per lane)
            bi = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[0]);
            if (bi >= nlaunch) break;
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[1]); v0c = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[2]);
            v1c = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[3]); col0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb[4]);
        } else {
            const gt_u32x4 d = ldesc[chunk0 + blockIdx.x];
            bi = blockIdx.x; c = d.x; v0c = d.y; v1c = d.z; col0 = d.w;
        }
        chunk(bi, c, v0c, v1c, col0);
        if (!queue) break;
    }
}

// ------------------------------------------------------------------ phase 2
template <class T, bool IS_MIN> __device__ __forceinline__ void lds_combine(T *acc, uint32_t r, T a) {
#ifdef GT_EXP_P2_NO_ATOMICS   // timing experiment (wrong results): the stream is loaded, nothing is combined -- one plain LDS store per lane keeps the loads alive
    if (r == 0x7FFFu && a == (T)12345) acc[r & (R - 1)] = a;
    return;
#endif
    if constexpr (IS_MIN) { if (a != GT_INF) atomicMin(&acc[r], a); }
    else if constexpr (sizeof(T) == 8) unsafeAtomicAdd(&acc[r], a);   // ds_add_f64
    else atomicAdd(&acc[r], a);
}

// FUSE: 0 = plain SpMV; 1 / 2 = PageRank epilogue (gt_pr_epilogue) with an f32 / f64 message vector
template <class T, class TV, bool IS_MIN, int FUSE>
__global__ void __launch_bounds__(P2_THREADS) k_pb_gather(const BinWork *__restrict__ work, const C4 *__restrict__ LROW4,
                                                          const V4<TV> *__restrict__ VAL4, uint32_t nrows, T *__restrict__ y,
                                                          const uint32_t *__restrict__ active_prefix, gt_pr_epilogue epi,
                                                          uint32_t *__restrict__ queue, uint32_t nlaunch) {
    __shared__ T acc[R + 1];   // row R: the dummy row every pad output targets
    __shared__ unsigned wsum[P2_THREADS / 64];
    __shared__ uint32_t next_bi, p2_trip;
    auto item = [&](const uint32_t bi) {   // one entry of the work list: position bi of the launch
    const BinWork wk = work[bi];
    if (active_prefix && active_prefix[wk.c_hi + 1] == active_prefix[wk.c_lo]) return;   // no active chunk feeds this slice
    const T neutral = IS_MIN ? (T)GT_INF : (T)0;
#ifdef GT_EXP_TRACE
    if (threadIdx.x == 0 && bi < 8192) {
        gt_trace_p2[8 * bi] = wall_clock64(); gt_trace_p2[8 * bi + 3] = gt_trace_where();
        gt_trace_p2[8 * bi + 4] = ((unsigned long long)wk.bin << 32) | (wk.k1 - wk.k0); gt_trace_p2[8 * bi + 5] = wk.single;
    }
#define GT_TRACE_P2_END() do { if ((threadIdx.x & 63) == 0 && bi < 8192) atomicMax(&gt_trace_p2[8 * bi + 2], (unsigned long long)wall_clock64()); } while (0)
#else
#define GT_TRACE_P2_END() do { } while (0)
#endif
    for (uint32_t i = threadIdx.x; i <= R; i += P2_THREADS) acc[i] = neutral;
#ifndef GT_P2_DYN
#define GT_P2_DYN 1
#endif
    if (GT_P2_DYN && threadIdx.x == 0) p2_trip = P2_THREADS / 64;   // the first trip of a wave is its own
    __syncthreads();
    // 4 consecutive entries per lane per load (16-byte VAL loads for f32/u32 streams, 2 x 16 B for f64;
    // 8-byte LROW loads), P2_U quads in flight per lane (one workgroup per CU: the loads in flight have to cover
    // the HBM latency by themselves); every range is a multiple of 4 (padded runs).
    const uint32_t qb = wk.k1 >> 2;
#if GT_P2_DYN
    // DYNAMIC TRIPS, as in phase 1's wide kernel (k_pb_scatter, DYN): a wave takes its next P2_U x 64 quads from a counter of the
    // workgroup -- with equal shares the SIMD's oldest waves are done long before its youngest
    {
        const uint32_t lane = threadIdx.x & 63, q0 = wk.k0 >> 2;
        constexpr uint32_t TQ = P2_U * 64;   // quads of a wave's trip
        uint32_t t = threadIdx.x >> 6;
        for (;;) {
            const uint32_t base = q0 + t * TQ;
            if (base >= qb) break;
            uint32_t tn = 0;
            if (lane == 0) tn = atomicAdd(&p2_trip, 1u);
            if (base + TQ <= qb) {
                C4 r[P2_U]; V4<TV> a[P2_U];
#pragma unroll
                for (int u = 0; u < P2_U; u++) { r[u] = ld_stream<NT_P2>(LROW4 + (base + u * 64 + lane)); a[u] = ld_stream<NT_P2>(VAL4 + (base + u * 64 + lane)); }
#pragma unroll
                for (int u = 0; u < P2_U; u++)
#pragma unroll
                    for (int j = 0; j < 4; j++) lds_combine<T, IS_MIN>(acc, r[u].c[j], (T)a[u].a[j]);
            } else {
                for (uint32_t qq = base + lane; qq < qb; qq += 64) {
                    const C4 r0 = ld_stream<NT_P2>(LROW4 + qq); const V4<TV> a0 = ld_stream<NT_P2>(VAL4 + qq);
#pragma unroll
                    for (int j = 0; j < 4; j++) lds_combine<T, IS_MIN>(acc, r0.c[j], (T)a0.a[j]);
                }
            }
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tn);
        }
    }
    uint32_t q = qb;
#else
    uint32_t q = (wk.k0 >> 2) + threadIdx.x;
#endif
    for (; q + (P2_U - 1) * P2_THREADS < qb; q += P2_U * P2_THREADS) {
        C4 r[P2_U]; V4<TV> a[P2_U];
#pragma unroll
        for (int u = 0; u < P2_U; u++) { r[u] = ld_stream<NT_P2>(LROW4 + (q + u * P2_THREADS)); a[u] = ld_stream<NT_P2>(VAL4 + (q + u * P2_THREADS)); }
#pragma unroll
        for (int u = 0; u < P2_U; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) lds_combine<T, IS_MIN>(acc, r[u].c[j], (T)a[u].a[j]);
    }
    for (; q < qb; q += P2_THREADS) {
        const C4 r0 = ld_stream<NT_P2>(LROW4 + q); const V4<TV> a0 = ld_stream<NT_P2>(VAL4 + q);
#pragma unroll
        for (int j = 0; j < 4; j++) lds_combine<T, IS_MIN>(acc, r0.c[j], (T)a0.a[j]);
    }
    __syncthreads();
#ifdef GT_EXP_TRACE
    if (threadIdx.x == 0 && bi < 8192) gt_trace_p2[8 * bi + 1] = wall_clock64();   // the stream has been combined
#endif
    const uint32_t row0 = wk.bin << RB;
    const uint32_t rn = (nrows - row0 < R) ? nrows - row0 : R;
#ifdef GT_EXP_P2_NO_FLUSH   // timing experiment (wrong results): the workgroup ends when its stream has been combined
    return;
#endif
    if constexpr (FUSE != 0) {
        if (wk.single) {   // complete sums of the bin's rows: apply them here (same arithmetic as k_pr_apply_msg, engine.hip)
            using TX = typename std::conditional<FUSE == 1, float, double>::type;
            TX *__restrict__ xo = (TX *)epi.x;
            unsigned act = 0;
            // EB rows per thread and trip, every load of the trip issued before the first store: the flush of a 16 384-row bin is
            // 16 dependent round trips per thread otherwise (latency-bound for ~30 % of the workgroup's life)
#ifndef GT_P2_EB
#define GT_P2_EB 16   // with the lean applicator (two loads per row) a whole 16 384-row bin in one trip: phase 2 0.571 -> 0.557 ms (A/B, gpurun_out/s3/ab1.txt); 4: +2 %
#endif
            constexpr int EB = GT_P2_EB;
            for (uint32_t i0 = threadIdx.x; i0 < rn; i0 += EB * P2_THREADS) {
                uint32_t c[EB], d[EB]; double tmp[EB]; bool live[EB];
#pragma unroll
                for (int u = 0; u < EB; u++) {
                    const uint32_t i = i0 + u * P2_THREADS, r = row0 + i;
                    live[u] = i < rn;
                    c[u] = live[u] ? ld_stream<NT_EPI>(epi.R2X + r) : 0xFFFFFFFFu;
                    tmp[u] = (live[u] && epi.state == 0) ? epi.rank_c[r] : 0.0;
                    d[u] = live[u] ? ld_stream<NT_EPI>(epi.deg_c + r) : 0u;
                }
#pragma unroll
                for (int u = 0; u < EB; u++) {
                    const uint32_t i = i0 + u * P2_THREADS, r = row0 + i;
                    const bool source = (c[u] == 0xFFFFFFFFu);
                    if (!live[u] || (epi.cf && source && !epi.last)) continue;   // vp:1671-1691
                    const double nv = epi.alpha + (1.0 - epi.alpha) * (double)acc[i];
                    if (epi.state != 2) epi.rank_c[r] = nv;
                    if (epi.state == 0) {   // (1, 2: an iteration whose rank / changed flag nobody can see -- gt_internal.h, pr_state)
                        const uint8_t ch = fabs(nv - tmp[u]) > epi.tol;
                        epi.C_c[r] = ch;
                        act += (ch && !(epi.cf && source));
                    }
#ifdef GT_EXP_P2_NO_XSTORE   // timing experiment (wrong results): the next messages are computed and not stored (one store per workgroup keeps the work)
                    if (!source && nv == 1.2345e300) xo[c[u]] = (TX)(d[u] ? nv / (double)d[u] : 0.0);
#elif defined(GT_EXP_P2_SEQ_XSTORE)   // timing experiment (wrong results): the same stores, to consecutive addresses instead of through the row -> slot map
                    if (!source) xo[r] = (TX)(d[u] ? nv / (double)d[u] : 0.0);
#else
                    if (!source) xo[c[u]] = (TX)(d[u] ? nv / (double)d[u] : 0.0);
#endif
                }
            }
            if (epi.d_active) {   // one atomic per workgroup
                for (int o = 32; o > 0; o >>= 1) act += __shfl_down(act, o);
                if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = act;
                __syncthreads();
                if (threadIdx.x == 0) {
                    unsigned t = 0;
                    for (int w = 0; w < P2_THREADS / 64; w++) t += wsum[w];
                    if (t) atomicAdd(epi.d_active, (unsigned long long)t);
                }
            }
            GT_TRACE_P2_END();
            return;
        }
    }
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
    GT_TRACE_P2_END();
    };
    // persistent form, as in phase 1 (k_pb_scatter): one workgroup per slot of the chip, the work list drawn from a counter
    // (ONE call site for both forms: see k_pb_scatter)
#ifndef GT_P2_PERSIST
#define GT_P2_PERSIST 0   // (measured: no gain -- phase 2 runs at its mix's HBM rate with or without dispatch gaps; -DGT_P2_PERSIST=1 + GRAPHTAP_PB_PERSIST=2 for the A/B)
#endif
    if constexpr (GT_P2_PERSIST == 0) item(blockIdx.x);
    else
    for (;;) {
        uint32_t bi = blockIdx.x;
        if (queue) {
            __syncthreads();   // every wave is done with the accumulators (and has read next_bi)
            if (threadIdx.x == 0) { const uint32_t i = atomicAdd(queue, 1u); if (i == nlaunch + gridDim.x - 1) atomicExch(queue, 0u); next_bi = i; }
            __syncthreads();
            bi = next_bi;
            if (bi >= nlaunch) break;
        }
        item(bi);
        if (!queue) break;
    }
}

}  // namespace
#ifdef GT_EXP_TRACE
extern "C" int gt_exp_trace_dump(unsigned long long *p1, unsigned long long *p2) {   // 4 x 16384 words each (experiment builds only; not in the header)
    GT_HIP(hipDeviceSynchronize());
    GT_HIP(hipMemcpyFromSymbol(p1, HIP_SYMBOL(gt_trace_p1), sizeof(unsigned long long) * 4 * 16384));
    GT_HIP(hipMemcpyFromSymbol(p2, HIP_SYMBOL(gt_trace_p2), sizeof(unsigned long long) * 4 * 16384));
    return GT_OK;
}
extern "C" int gt_exp_trace_dump_waves(unsigned long long *w) {   // 16 x 8192 words
    GT_HIP(hipDeviceSynchronize());
    GT_HIP(hipMemcpyFromSymbol(w, HIP_SYMBOL(gt_trace_w), sizeof(unsigned long long) * 16 * 8192));
    return GT_OK;
}
#endif

struct gt_pb {
    bool wide = false;   // the WIDE build: windows of 2 W / 2 WS slots (128 KiB of 4-byte messages in LDS, one phase-1 workgroup per CU): ~12 % fewer value-stream slots
    uint32_t nbins = 0, nchunks = 0, nwork = 0, nnz = 0;
    uint32_t ndense = 0;       // graphs with an exchange layout (one class, K slices): chunks [0, ndense) are dense (all of them)
    // chunk ranges by (row class, kind): [b[0], b[1]) regular rows / dense windows, [b[1], b[2]) regular / sparse,
    // [b[2], b[3]) source rows / dense, [b[3], b[4]) source rows / sparse
    uint32_t bound[5] = {0, 0, 0, 0, 0};
    uint64_t nnz_source = 0;   // entries of source rows (class 1)
    uint32_t np = 0;           // padded entries of the v-order (multiple of 4)
    uint32_t nout = 0;         // padded outputs of the k-order (multiple of 4): slots of VAL / LROW
    uint32_t *cv0 = nullptr, *cv1 = nullptr, *ccol0 = nullptr;
    uint16_t *LCOL = nullptr, *LROW = nullptr;
    void *WT = nullptr;        // weights of the v-order, wt_bytes (1, 2 or 4) each
    int wt_bytes = 4;
    uint32_t *KSTART = nullptr;
    void *G = nullptr;         // GroupRec per 256 padded entries
    BinWork *work = nullptr;
    std::vector<uint32_t> slice_chunk;   // chunks of slice k of the message vector: [slice_chunk[k], slice_chunk[k+1])
    void *VAL = nullptr;       // value stream scratch, 8 B/entry once an f64 SpMV ran, else 4 B/entry
    uint32_t val_bytes = 0;
    uint32_t val_cap = 0;      // bytes per slot VAL was allocated for (gt_pb_reserve_val)
    // GRAPHTAP_PB_PHASE_TIMING=1: HIP events around phase 1 and phase 2 of every whole SpMV (gt_graph_phase_times)
    std::vector<hipEvent_t> pt_ev;   // triples: before phase 1, between, after phase 2
    size_t pt_used = 0;
    uint32_t val_allocs = 0;   // allocations of VAL so far (gt_exec_stats.allocs_in_execute counts those made inside execute())
    uint32_t *chunk_active = nullptr, *active_prefix = nullptr;   // activity filtering of the min programs
    uint32_t *launch_order = nullptr;   // [nchunks] phase-1 workgroup -> chunk: inside every slice (and kind), largest chunk first
    uint32_t *ldesc = nullptr;          // [nchunks][4] the same order with each chunk's description: {chunk, cv0, cv1, ccol0}
    // persistent phase 1 (k_pb_scatter, `queue`): a ring of zeroed counters, one per launch in flight (slices of an exchange layout run
    // side by side on streams of their own), and the CUs of the device
    uint32_t *p1_queue = nullptr; uint32_t p1_seq = 0; int ncu = 0;
    uint8_t *bin_single = nullptr;      // [nbins] 1 = the bin has exactly one phase-2 workgroup
    uint32_t *split_bins = nullptr;     // [nsplit] the other bins (with entries): their rows go through y and the apply kernel
    uint32_t nsplit = 0;
    // Exchange layout, K slices: the phase-2 work list in K PARTS -- part k = the row bins whose first row's column travels in
    // slice k of the NEXT iteration's exchange (rows and the columns of the same vertices ascend together, the slice of compressed
    // column j is j / slice_width) -- so that the messages of slice k are complete, packed and on the wire while parts k+1..
    // still run (gt_program_phase2_part, dist.hip). work[work_part[k] .. work_part[k+1]) (largest first inside a part),
    // split_bins[split_part[k] .. split_part[k+1]).
    std::vector<uint32_t> work_part, split_part;
    // per window of x: stored entries (both row classes), for the activity statistics / the hybrid pass of the min programs
    uint32_t *win_entries = nullptr; uint32_t nwin = 0; uint32_t ndw_ = 0, dense_end_ = 0;
    uint32_t *xdeg = nullptr;          // [x_len] entries of the column in slot sl (0 for an unused slot)
    unsigned long long *win_act = nullptr;   // [nwin] scratch: entries of the ACTIVE columns of every window
    // HYBRID pass of the min programs (pb_run): windows whose active columns hold at most 1/32 of their entries are left out of
    // the streaming pass; those columns' entries go through a column-driven SpMSpV (atomicMin on y) instead
    uint8_t *win_mode = nullptr;             // [nwin] 0 = stream, 1 = its active columns go to the SpMSpV, 2 = no active column
    uint32_t *hy_col = nullptr, *hy_val = nullptr, *hy_deg = nullptr, *hy_long = nullptr;   // [hy_cap] the columns of the mode-1 windows; indices of the long ones
    uint32_t hy_cap = 0;
    uint32_t *hy_cand = nullptr; uint32_t hy_ncand = 0;   // the candidate windows (at least nnz / 256 entries each)
    unsigned int *hy_cnt = nullptr;          // [2] columns listed, long columns among them
    unsigned long long *hy_stat = nullptr;   // [4] running totals: hybrid passes, entries left to the SpMSpV, entries left out of the stream, windows left out
    uint32_t rows_single = 0;           // rows of those bins
    const void *val_owner = nullptr;   // program (and its initialize epoch) whose messages VAL currently holds
    uint64_t val_epoch = 0;
    int val_kind = 0;          // 1: f32 messages of an f64 sum, 2: f64, 3: u32
    int val_min = -1;          // which neutral value the pad slots of VAL currently hold (0: zero, 1: INF)
};

void gt_pb_free(gt_pb *pb) {
    if (!pb) return;
    void *ptrs[] = {pb->cv0, pb->cv1, pb->ccol0, pb->LCOL, pb->LROW, pb->WT, pb->KSTART, pb->G, pb->work, pb->VAL, pb->chunk_active, pb->active_prefix, pb->launch_order, pb->ldesc, pb->p1_queue, pb->bin_single, pb->split_bins,
                    pb->win_entries, pb->xdeg, pb->win_act, pb->win_mode, pb->hy_col, pb->hy_val, pb->hy_deg, pb->hy_long, pb->hy_cnt, pb->hy_stat, pb->hy_cand};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : pb->pt_ev) (void)hipEventDestroy(e);
    delete pb;
}

#define LAY_HIP(call)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            gt_set_error("layout build: %s failed: %s (line %d)", #call, hipGetErrorString(e_), __LINE__); \
            return GT_ERR_HIP;                                                                      \
        }                                                                                           \
    } while (0)

// Layout of the message vector (gt_internal.h): hubs first on a single rank without an exchange layout.
int gt_layout_build(gt_graph *g) {
    const uint32_t nc = g->info.nnzcols, nr = g->info.nnzrows;
    g->x_len = g->ncols_total;
    g->ndw = (g->ncols_total + W - 1) / W;     // identity layout: every window is a dense window
    const char *e = gt_cfg(g, "GRAPHTAP_PB_HUBS");
    if (gt_has_exchange(g) || nc == 0 || (e && atoi(e) == 0)) return GT_OK;
    hipStream_t s = 0;
    // Hub = a column with at least `thr` entries: its window is worth the aggregating kernel. R-MAT-26: the 1.05 M columns of
    // degree >= 100 hold 73 % of the entries and 4.3 entries per (window, row) pair; columns below ~20 entries sit in windows
    // with 1.0x entries per pair, where the aggregation machinery only costs (tools/layout_stats.py).
    // Since the outputs of the plus kernels leave through the staged stores a sparsely filled dense window costs less: 12
    // instead of 24 is 393 M instead of 400 M slots on R-MAT-26 and ~3 % (PageRank 1.59 -> 1.54 ms, A/B); the weighted graphs
    // (SSSP: direct stores) keep 24, where 12 was 3-4 % slower; BFS / CC are indifferent.
    const char *et = gt_cfg(g, "GRAPHTAP_PB_HUB_DEG");
    // Re-measured with round 3's final kernels (profiles/r03/ab_hub_threshold_final.txt, four rounds on one box): 8 or 10 instead of 12 is a
    // steady 1 % for PageRank (phase 2 -2 %: fewer slots), nothing for BFS and +1-2 % for CC on their symmetrised graphs: 8 on directed graphs.
    const uint32_t thr = et ? (uint32_t)atoi(et) : (g->info.weighted ? 24u : g->flags.directed ? 8u : 12u);
    struct Buf { void *p = nullptr; ~Buf() { if (p) gt_scratch_free(p); } } deg, hubflag, tailflag, hubpos, tailpos, key, key2, col, col2, tmp;
    for (Buf *b : {&deg, &hubflag, &tailflag, &hubpos, &tailpos}) LAY_HIP(gt_scratch_malloc(&b->p, (uint64_t)(nc + 1) * 4));
    LAY_HIP(hipMemsetAsync(hubflag.p, 0, (uint64_t)(nc + 1) * 4, s));
    LAY_HIP(hipMemsetAsync(tailflag.p, 0, (uint64_t)(nc + 1) * 4, s));
    k_col_degrees<<<grid_for(nc), TPB, 0, s>>>(g->JA, nc, thr, (uint32_t *)deg.p, (uint32_t *)hubflag.p, (uint32_t *)tailflag.p);
    {
        size_t tb = 0;
        LAY_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, (uint32_t *)hubflag.p, (uint32_t *)hubpos.p, nc + 1, s));
        LAY_HIP(gt_scratch_malloc(&tmp.p, tb ? tb : 1));
        LAY_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, (uint32_t *)hubflag.p, (uint32_t *)hubpos.p, nc + 1, s));
        LAY_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, (uint32_t *)tailflag.p, (uint32_t *)tailpos.p, nc + 1, s));
    }
    uint32_t nhub = 0;
    LAY_HIP(hipMemcpy(&nhub, (uint32_t *)hubpos.p + nc, 4, hipMemcpyDeviceToHost));
    const uint32_t ntail = nc - nhub;
    const uint32_t ndw = (nhub + W - 1) / W;
    const uint64_t xl = (uint64_t)ndw * W + ntail;
    GT_REQUIRE(xl < 0xFFFFFFF0ull, GT_ERR_UNSUPPORTED, "message vector exceeds 32-bit slot ids");
    LAY_HIP(hipMalloc((void **)&g->xslot, (uint64_t)nc * 4));
    LAY_HIP(hipMalloc((void **)&g->XV, std::max<uint64_t>(xl, 1) * 4));
    LAY_HIP(hipMalloc((void **)&g->xcol, std::max<uint64_t>(xl, 1) * 4));
    LAY_HIP(hipMalloc((void **)&g->R2X, (uint64_t)std::max(nr, 1u) * 4));
    if (nhub) {
        for (Buf *b : {&key, &key2, &col, &col2}) LAY_HIP(gt_scratch_malloc(&b->p, (uint64_t)nhub * 4));
        k_hub_list<<<grid_for(nc), TPB, 0, s>>>((const uint32_t *)deg.p, (const uint32_t *)hubflag.p, (const uint32_t *)hubpos.p, nc, (uint32_t *)key.p, (uint32_t *)col.p);
        hipcub::DoubleBuffer<uint32_t> dk((uint32_t *)key.p, (uint32_t *)key2.p), dc((uint32_t *)col.p, (uint32_t *)col2.p);
        size_t tb = 0;
        LAY_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dc, nhub, 0, 32, s));   // stable: equal degrees keep ascending column order
        Buf st; LAY_HIP(gt_scratch_malloc(&st.p, tb ? tb : 1));
        LAY_HIP(hipcub::DeviceRadixSort::SortPairs(st.p, tb, dk, dc, nhub, 0, 32, s));
        k_slots_hub<<<grid_for(nhub), TPB, 0, s>>>(dc.Current(), nhub, g->xslot);
        LAY_HIP(hipStreamSynchronize(s));
    }
    k_slots_tail<<<grid_for(nc), TPB, 0, s>>>((const uint32_t *)tailflag.p, (const uint32_t *)tailpos.p, nc, ndw * W, g->xslot);
    LAY_HIP(hipMemsetAsync(g->XV, 0xFF, std::max<uint64_t>(xl, 1) * 4, s));
    LAY_HIP(hipMemsetAsync(g->xcol, 0xFF, std::max<uint64_t>(xl, 1) * 4, s));
    k_slot_vertices<<<grid_for(nc), TPB, 0, s>>>(g->xslot, g->JC, nc, g->XV, g->xcol);
    if (nr) k_row_slots<<<grid_for(nr), TPB, 0, s>>>(g->xslot, g->R2C, nr, g->R2X);
    LAY_HIP(hipStreamSynchronize(s));
    LAY_HIP(hipGetLastError());
    g->x_len = (uint32_t)xl; g->ndw = ndw;
    if (getenv("GRAPHTAP_PB_STATS"))
        fprintf(stderr, "[pb] layout: hubs first, degree >= %u: %u hub columns in %u dense windows, %u tail columns, x has %u slots\n", thr, nhub, ndw, ntail, g->x_len);
    return GT_OK;
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
#define PB_SCAN_EXCL(in, out, n)                                                                    \
    do {                                                                                            \
        size_t tb_ = 0;                                                                             \
        PB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb_, (in), (out), (n), s));                \
        DevBuf st_; PB_ALLOC(st_, tb_);                                                             \
        PB_HIP(hipcub::DeviceScan::ExclusiveSum(st_.p, tb_, (in), (out), (n), s));                  \
        PB_HIP(hipStreamSynchronize(s));                                                            \
    } while (0)

static int pb_build_impl(gt_graph *g, bool wide, gt_pb **out);
int gt_pb_build(gt_graph *g) {
    if (!g->pb) { gt_pb *pb = nullptr; int st = pb_build_impl(g, false, &pb); if (st != GT_OK) return st; g->pb = pb; }
    // The WIDE build beside it, for the 4-byte-message PageRank (GT_SPMV_PB_F32MSG) on one rank: windows of 32 766 / 32 768 slots --
    // pairs of the narrow ones, the layout of x is shared -- hold twice the columns per (window, row) pair: ~12 % fewer value-stream
    // slots (4 B written + 4 B read + 2 B of row id each). 128 KiB of LDS: one phase-1 workgroup per CU; f64 messages (converge mode,
    // gt_spmv) keep the narrow build (an f64 window of that width would be 256 KiB). GRAPHTAP_PB_WIDE=0 / 1: never / for every graph
    // without an exchange layout.
    const char *ew = gt_cfg(g, "GRAPHTAP_PB_WIDE");
    // ... by default only from ~0.47 G stored entries (R-MAT-25): measured by scale (PageRank f32 messages, narrow -> wide). Workgroups dispatched
    // per chunk (profiles/r04/ab_wide_windows_by_scale.txt): R-MAT-22 458 -> 423 GTEPS, 23: 536 -> 547, 24: 662 -> 633, 25: 716 -> 698,
    // 26: 686 -> 710 -- phase 1 pays its +18 % everywhere, phase 2's -20 % outweighs it only on the largest graph. With the persistent
    // phase 1 (profiles/r04/ab_recheck_after_register_fix.txt): R-MAT-20 174 -> 157, 21: 281 -> 273, 22: 422 -> 418, 24: 630 -> 633,
    // 25: 715 -> 707 (ab_recheck_after_register_fix.txt, third block). With dynamic trips in the wide kernel and the rotating priority in
    // the narrow one (ab_wide_by_scale_final.txt): R-MAT-22 462 -> 425, 23: 532 -> 569, 24: 680 -> 672, 25: 722 -> 751, 26: wide.
    // (An A/B in between had the wide build win from R-MAT-21 up -- against a narrow kernel that a doubled register count had cut to
    // one workgroup per CU; see k_pb_scatter, "ONE call site".)
    const bool want = ew ? atoi(ew) != 0 : (g->spmv_variant == GT_SPMV_PB_F32MSG && g->info.nnz_local >= (7ull << 26));
    if (want && !g->pb_wide && !gt_has_exchange(g) && g->info.nnz_local) { gt_pb *pb = nullptr; int st = pb_build_impl(g, true, &pb); if (st != GT_OK) return st; g->pb_wide = pb; }
    return GT_OK;
}
static int pb_build_impl(gt_graph *g, bool wide, gt_pb **out) {
    const uint32_t nnz = (uint32_t)g->info.nnz_local, nc = g->info.nnzcols, nr = g->info.nnzrows;
    gt_pb *pb = new gt_pb();
    pb->wide = wide;
    pb->nnz = nnz;
    pb->nbins = std::max<uint32_t>(1, (nr + R - 1) / R);
    *out = nullptr;
    if (nnz == 0) {   // a tile-row without entries: nothing to stream, but the K (empty) parts of phase 2 exist like everywhere else
        const uint32_t KP = gt_has_exchange(g) ? std::max<uint32_t>(g->info.x_slices, 1) : 1;
        pb->work_part.assign(KP + 1, 0); pb->split_part.assign(KP + 1, 0);
        *out = pb; return GT_OK;
    }
    hipStream_t s = 0;
    int binbits = 1;
    while ((1u << binbits) < pb->nbins) binbits++;
    const uint32_t binmask = (1u << binbits) - 1;
    // windows of the message vector: g->ndw dense ones, sparse ones behind them (gt_layout_build). The columns the kernels
    // index are the ncols_total columns of JA (compressed ids; the needed columns on a graph with an exchange layout).
    const uint32_t ncols = g->ncols_total;
    WinGeom geom;
    const bool wgeom = wide;
    geom.wd = wgeom ? 2 * W : W; geom.ws = wgeom ? 2 * WS : WS;   // (the wide build: pairs of the layout's windows; an odd last dense window stays single)
    geom.ndw = wgeom ? (g->ndw + 1) / 2 : g->ndw; geom.dense_end = g->ndw * W; geom.x_len = g->x_len;
    geom.nwin = geom.ndw + (g->x_len > geom.dense_end ? (g->x_len - geom.dense_end + geom.ws - 1) / geom.ws : 0u);
    // row classes: source rows apart (see WinGeom) unless the graph has an exchange layout (K slices of chunks in column order)
    const bool classes = !gt_has_exchange(g) && nr > 0 && !(getenv("GRAPHTAP_PB_CLASSES") && atoi(getenv("GRAPHTAP_PB_CLASSES")) == 0);
    geom.ncls = classes ? 2u : 1u; geom.nvwin = geom.nwin * geom.ncls;
    const uint32_t nwin = geom.nvwin;   // "windows" below are virtual windows: (row class, window)
    (void)nc;
    uint32_t ch = ch_default(g, nnz, geom.nwin);
    if (wide && !gt_cfg(g, "GRAPHTAP_PB_CH") && ch >= (1u << 19)) ch <<= 1;   // a wide window costs twice the staging: twice the entries behind each
    DevBuf wcount, nsub, cbase, cutflag, cutidx, plan, srcbits_b;
    const uint32_t *srcbits = nullptr;
    if (classes) {
        PB_ALLOC(srcbits_b, ((uint64_t)nr / 32 + 4) * 4);
        k_source_bits<<<grid_for(nr), TPB, 0, s>>>(g->R2C, nr, srcbits_b.as<uint32_t>());
        srcbits = srcbits_b.as<uint32_t>();
    }
    PB_ALLOC(wcount, (uint64_t)(nwin + 1) * 4); PB_ALLOC(nsub, (uint64_t)(nwin + 1) * 4); PB_ALLOC(cbase, (uint64_t)(nwin + 1) * 4);
    PB_ALLOC(cutflag, (uint64_t)(nwin + 1) * 4); PB_ALLOC(cutidx, (uint64_t)(nwin + 1) * 4);
    PB_HIP(hipMemsetAsync(wcount.p, 0, (uint64_t)(nwin + 1) * 4, s));
    k_win_count<<<grid_for(ncols), TPB, 0, s>>>(g->JA, ncols, g->xslot, geom, wcount.as<uint32_t>());
    if (classes) {
        k_win_count_src<<<grid_for(nnz), TPB, 0, s>>>(g->JI, g->IA, nnz, g->xslot, geom, srcbits, wcount.as<uint32_t>() + geom.nwin);
        k_win_count_move<<<grid_for(geom.nwin), TPB, 0, s>>>(wcount.as<uint32_t>(), geom.nwin);
    }
    uint32_t nchunks = 0;
    for (;;) {  // chunk ids must fit above the bin bits of a 32-bit sort key
        PB_HIP(hipMemsetAsync(nsub.p, 0, (uint64_t)(nwin + 1) * 4, s));
        PB_HIP(hipMemsetAsync(cutflag.p, 0, (uint64_t)(nwin + 1) * 4, s));
        k_win_sizes<<<grid_for(nwin), TPB, 0, s>>>(wcount.as<uint32_t>(), nwin, ch, nsub.as<uint32_t>(), cutflag.as<uint32_t>());
        PB_SCAN_EXCL(cutflag.as<uint32_t>(), cutidx.as<uint32_t>(), nwin + 1);
        uint32_t ncut = 0;
        PB_HIP(hipMemcpy(&ncut, cutidx.as<uint32_t>() + nwin, 4, hipMemcpyDeviceToHost));
        if (plan.p) { gt_scratch_free(plan.p); plan.p = nullptr; }
        GT_REQUIRE((uint64_t)std::max(ncut, 1u) * pb->nbins < 0xFFFFFFFFull, GT_ERR_UNSUPPORTED, "%u heavy windows x %u row bins: the chunk plan exceeds 32-bit indexing", ncut, pb->nbins);
        PB_ALLOC(plan, (uint64_t)std::max(ncut, 1u) * pb->nbins * 4);
        if (ncut) {
            PB_HIP(hipMemsetAsync(plan.p, 0, (uint64_t)ncut * pb->nbins * 4, s));
            k_win_hist<<<grid_for(nnz), TPB, 0, s>>>(g->JI, g->IA, nnz, g->xslot, geom, srcbits, pb->nbins, cutflag.as<uint32_t>(), cutidx.as<uint32_t>(), plan.as<uint32_t>());
            k_win_plan<<<grid_for(nwin), TPB, 0, s>>>(wcount.as<uint32_t>(), nwin, ch, pb->nbins, cutflag.as<uint32_t>(), cutidx.as<uint32_t>(),
                                                     plan.as<uint32_t>(), nsub.as<uint32_t>());
        }
        PB_SCAN_EXCL(nsub.as<uint32_t>(), cbase.as<uint32_t>(), nwin + 1);
        PB_HIP(hipMemcpy(&nchunks, cbase.as<uint32_t>() + nwin, 4, hipMemcpyDeviceToHost));
        if ((uint64_t)nchunks < (1ull << (32 - binbits))) break;
        ch *= 2;
    }
    int chunkbits = 1;
    while ((1ull << chunkbits) < nchunks) chunkbits++;
    pb->nchunks = nchunks;
    {   // chunk ranges by (row class, kind of window)
        const uint32_t at[5] = {0, std::min(geom.ndw, geom.nwin), geom.nwin, std::min(geom.nwin + geom.ndw, geom.nvwin), geom.nvwin};
        for (int i = 0; i < 5; i++) PB_HIP(hipMemcpy(&pb->bound[i], cbase.as<uint32_t>() + std::min(at[i], geom.nvwin), 4, hipMemcpyDeviceToHost));
        pb->ndense = pb->bound[1];
        if (classes) {
            std::vector<uint32_t> hw(geom.nvwin);
            PB_HIP(hipMemcpy(hw.data(), wcount.p, (uint64_t)geom.nvwin * 4, hipMemcpyDeviceToHost));
            for (uint32_t q = geom.nwin; q < geom.nvwin; q++) pb->nnz_source += hw[q];
        }
    }
    PB_MALLOC(pb->cv0, (uint64_t)nchunks * 4); PB_MALLOC(pb->cv1, (uint64_t)nchunks * 4); PB_MALLOC(pb->ccol0, (uint64_t)nchunks * 4);
    k_fill_chunks<<<grid_for(nwin), TPB, 0, s>>>(nwin, nsub.as<uint32_t>(), cbase.as<uint32_t>(), geom, pb->ccol0);

    // v-order: entries sorted by (chunk, bin, row); the radix sort is stable, so inside a run rows ascend and,
    // for equal rows, the column-major input order (ascending compressed column) survives
    DevBuf key, key2, idx, idx2, rkey, sidb;
    PB_ALLOC(key, (uint64_t)nnz * 8); PB_ALLOC(key2, (uint64_t)nnz * 8); PB_ALLOC(idx, (uint64_t)nnz * 4); PB_ALLOC(idx2, (uint64_t)nnz * 4);
    k_keys<<<grid_for(nnz), TPB, 0, s>>>(g->JI, g->IA, nnz, g->xslot, geom, srcbits, cbase.as<uint32_t>(), cutflag.as<uint32_t>(), cutidx.as<uint32_t>(),
                                         plan.as<uint32_t>(), pb->nbins, binbits, key.as<uint64_t>(), idx.as<uint32_t>());
    hipcub::DoubleBuffer<uint64_t> dk(key.as<uint64_t>(), key2.as<uint64_t>());
    hipcub::DoubleBuffer<uint32_t> di(idx.as<uint32_t>(), idx2.as<uint32_t>());
    {
        size_t tb = 0;
        PB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, di, nnz, 0, RB + binbits + chunkbits, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceRadixSort::SortPairs(st.p, tb, dk, di, nnz, 0, RB + binbits + chunkbits, s));
        PB_HIP(hipStreamSynchronize(s));
    }
    const uint64_t *skey64 = dk.Current();
    uint32_t *sidx = di.Current(), *head = di.Alternate();
    PB_ALLOC(rkey, (uint64_t)nnz * 4); PB_ALLOC(sidb, (uint64_t)nnz * 4);
    uint32_t *skey = rkey.as<uint32_t>(), *sid = sidb.as<uint32_t>();
    k_run_keys<<<grid_for(nnz), TPB, 0, s>>>(skey64, nnz, skey);
    k_heads<<<grid_for(nnz), TPB, 0, s>>>(skey, nnz, head);
    {
        size_t tb = 0;
        PB_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, head, sid, nnz, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceScan::InclusiveSum(st.p, tb, head, sid, nnz, s));
    }
    uint32_t nrun = 0;
    PB_HIP(hipMemcpy(&nrun, sid + (nnz - 1), 4, hipMemcpyDeviceToHost));
    DevBuf vstart, runkey, len, lenpad, runbin, runbin_s, order, iota, lenpad_s, kscan, pvstart, pkstart;
    for (DevBuf *b : {&vstart, &runkey, &len, &lenpad, &runbin, &runbin_s, &order, &iota, &lenpad_s, &kscan, &pvstart, &pkstart})
        PB_ALLOC(*b, (uint64_t)(nrun + 1) * 4);
    k_runs<<<grid_for(nnz), TPB, 0, s>>>(skey, sid, nnz, vstart.as<uint32_t>(), runkey.as<uint32_t>());
    PB_HIP(hipMemsetAsync(lenpad.as<uint32_t>() + nrun, 0, 4, s));
    k_run_lens<<<grid_for(nrun), TPB, 0, s>>>(vstart.as<uint32_t>(), runkey.as<uint32_t>(), nrun, nnz, binmask, len.as<uint32_t>(),
                                              lenpad.as<uint32_t>(), runbin.as<uint32_t>());
    PB_SCAN_EXCL(lenpad.as<uint32_t>(), pvstart.as<uint32_t>(), nrun + 1);
    // every chunk's padded v-range becomes a multiple of 256 entries (extra pads after its last run), so that a
    // 256-entry group of phase 1 never straddles two chunks
    k_align_chunks<<<grid_for(nrun), TPB, 0, s>>>(runkey.as<uint32_t>(), nrun, binbits, pvstart.as<uint32_t>(), lenpad.as<uint32_t>());
    PB_SCAN_EXCL(lenpad.as<uint32_t>(), pvstart.as<uint32_t>(), nrun + 1);   // pvstart[nrun] = padded total
    uint32_t np = 0;
    PB_HIP(hipMemcpy(&np, pvstart.as<uint32_t>() + nrun, 4, hipMemcpyDeviceToHost));
    {
        uint64_t chk = (uint64_t)nnz + 3ull * nrun + 255ull * nchunks;
        if (chk >= 0xFFFFFFF0ull) { gt_set_error("pb build: padded entry count exceeds 32 bits"); gt_pb_free(pb); return GT_ERR_UNSUPPORTED; }
    }
    pb->np = np;
    k_chunk_ranges<<<grid_for(nchunks), TPB, 0, s>>>(runkey.as<uint32_t>(), nrun, binbits, pvstart.as<uint32_t>(), nchunks, pb->cv0, pb->cv1);
    const int chunk_shift = RB + binbits;   // chunk id of a sorted entry = key >> chunk_shift
    const bool stats = getenv("GRAPHTAP_PB_STATS") != nullptr;
    if (stats) {  // entry-weighted histogram of run lengths
        std::vector<uint32_t> hl(nrun);
        PB_HIP(hipMemcpy(hl.data(), len.p, (uint64_t)nrun * 4, hipMemcpyDeviceToHost));
        uint64_t hist[33] = {0}, cnt[33] = {0};
        for (uint32_t l : hl) { int b = 0; while ((1u << (b + 1)) <= l) b++; hist[b] += l; cnt[b]++; }
        fprintf(stderr, "[pb] %s build (windows of %u / %u slots)\n", wide ? "WIDE" : "narrow", geom.wd, geom.ws);
        fprintf(stderr, "[pb] nnz=%u padded=%u (+%.2f%%) nbins=%u windows=%u (%u dense) chunks=%u (regular rows: %u dense + %u sparse, source rows: %u + %u) runs=%u mean run=%.1f\n", nnz, np,
                100.0 * (np - nnz) / nnz, pb->nbins, geom.nwin, geom.ndw, nchunks, pb->bound[1] - pb->bound[0], pb->bound[2] - pb->bound[1],
                pb->bound[3] - pb->bound[2], pb->bound[4] - pb->bound[3], nrun, (double)nnz / nrun);
        fprintf(stderr, "[pb] entries of source rows (class 1): %llu of %u (%.2f%%): PageRank under TCSC_CF leaves their chunks out of every iteration but the last\n",
                (unsigned long long)pb->nnz_source, nnz, 100.0 * pb->nnz_source / nnz);
        for (int b = 0; b < 33; b++) if (cnt[b]) fprintf(stderr, "[pb] run length [%u,%u): %10llu runs, %5.2f%% of entries\n", 1u << b, 1u << (b + 1), (unsigned long long)cnt[b], 100.0 * hist[b] / nnz);
    }
    // outputs: E marks the last entry of each output in the padded v-order, X = exclusive scan of E
    DevBuf Eb, Xb, noutpad;
    PB_ALLOC(Eb, ((uint64_t)np + 1) * 4); PB_ALLOC(Xb, ((uint64_t)np + 1) * 4); PB_ALLOC(noutpad, (uint64_t)(nrun + 1) * 4);
    PB_HIP(hipMemsetAsync(Eb.p, 0, ((uint64_t)np + 1) * 4, s));
    k_chunk_defaults<<<nchunks, TPB, 0, s>>>(pb->cv0, pb->cv1, pb->ccol0, geom.dense_end, Eb.as<uint32_t>(), nullptr, (uint16_t)geom.wd);   // sparse: pads are outputs too
    k_group_ends<<<grid_for(nnz), TPB, 0, s>>>(skey64, sid, nnz, vstart.as<uint32_t>(), len.as<uint32_t>(), pvstart.as<uint32_t>(), pb->ccol0, geom.dense_end,
                                               chunk_shift, Eb.as<uint32_t>());
    PB_SCAN_EXCL(Eb.as<uint32_t>(), Xb.as<uint32_t>(), (uint64_t)np + 1);
    // k-slots of a run: a multiple of four (16-byte loads of phase 2); GRAPHTAP_PB_KALIGN = log2 of a coarser alignment (experiment:
    // runs of different chunks -- different workgroups, mostly different XCDs -- then never share a cache line of the value stream)
    const uint32_t kalign = gt_cfg(g, "GRAPHTAP_PB_KALIGN") ? 1u << std::max(2, std::min(8, atoi(gt_cfg(g, "GRAPHTAP_PB_KALIGN")))) : 4u;
    k_run_outputs<<<grid_for(nrun), TPB, 0, s>>>(pvstart.as<uint32_t>(), Xb.as<uint32_t>(), nrun, noutpad.as<uint32_t>(), kalign);
    // k-order: runs by (bin, chunk) -- stable sort of the (chunk, bin)-ordered run list by bin
    k_iota<<<grid_for(nrun), TPB, 0, s>>>(iota.as<uint32_t>(), nrun);
    {
        size_t tb = 0;
        PB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, runbin.as<const uint32_t>(), runbin_s.as<uint32_t>(),
                                                  iota.as<const uint32_t>(), order.as<uint32_t>(), nrun, 0, binbits, s));
        DevBuf st; PB_ALLOC(st, tb);
        PB_HIP(hipcub::DeviceRadixSort::SortPairs(st.p, tb, runbin.as<const uint32_t>(), runbin_s.as<uint32_t>(),
                                                  iota.as<const uint32_t>(), order.as<uint32_t>(), nrun, 0, binbits, s));
    }
    k_gather_u32<<<grid_for(nrun), TPB, 0, s>>>(order.as<uint32_t>(), noutpad.as<uint32_t>(), nrun, lenpad_s.as<uint32_t>());
    PB_HIP(hipMemsetAsync(lenpad_s.as<uint32_t>() + nrun, 0, 4, s));
    PB_SCAN_EXCL(lenpad_s.as<uint32_t>(), kscan.as<uint32_t>(), nrun + 1);   // kscan[nrun] = padded outputs
    uint32_t nout = 0;
    PB_HIP(hipMemcpy(&nout, kscan.as<uint32_t>() + nrun, 4, hipMemcpyDeviceToHost));
    pb->nout = nout;
    if (stats) {   // how many (run, row) groups there are at all: the limit of any in-run aggregation
        DevBuf cntb; PB_ALLOC(cntb, 8); PB_HIP(hipMemsetAsync(cntb.p, 0, 8, s));
        k_count_run_rows<<<grid_for(nnz), TPB, 0, s>>>(skey64, nnz, cntb.as<unsigned long long>());
        unsigned long long uq = 0; PB_HIP(hipMemcpy(&uq, cntb.p, 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "[pb] distinct (run,row) groups: %llu of %u entries (factor %.3f)\n", uq, nnz, (double)nnz / uq);
        fprintf(stderr, "[pb] value-stream slots: %u for %u entries (factor %.3f)\n", nout, nnz, (double)nnz / nout);
        for (uint32_t mask : {7u, 63u, 255u}) {
            DevBuf cb; PB_ALLOC(cb, 8); PB_HIP(hipMemsetAsync(cb.p, 0, 8, s));
            k_count_ends<<<grid_for(nnz), TPB, 0, s>>>(skey64, sid, nnz, vstart.as<uint32_t>(), len.as<uint32_t>(), pvstart.as<uint32_t>(), pb->ccol0, geom.dense_end, chunk_shift, mask, cb.as<unsigned long long>());
            unsigned long long c = 0; PB_HIP(hipMemcpy(&c, cb.p, 8, hipMemcpyDeviceToHost));
            fprintf(stderr, "[pb] dense chunks, stretches of up to %u consecutive entries: %llu outputs\n", mask + 1, c);
        }
    }
    if (stats && !wide) {
        // Where the value-stream slots are: by WINDOW RANK (dense windows are in descending column-degree order, so the window index
        // is the rank) and by how heavy the row bin is. VERDICT round 3 asked for this table before any scheme that keeps the
        // heaviest windows' partial sums off HBM: such a scheme saves 10 B per slot of the windows it fuses (4 written + 4 read
        // + 2 of LROW) and has to stage every fused window once per row bin it meets (W x 4 B from L2 into LDS).
        std::vector<uint32_t> hk(nrun), hl(nrun), ho(nrun), hc(nchunks);
        PB_HIP(hipMemcpy(hk.data(), runkey.p, (uint64_t)nrun * 4, hipMemcpyDeviceToHost));
        PB_HIP(hipMemcpy(hl.data(), len.p, (uint64_t)nrun * 4, hipMemcpyDeviceToHost));
        PB_HIP(hipMemcpy(ho.data(), noutpad.p, (uint64_t)nrun * 4, hipMemcpyDeviceToHost));
        PB_HIP(hipMemcpy(hc.data(), pb->ccol0, (uint64_t)nchunks * 4, hipMemcpyDeviceToHost));
        std::vector<uint64_t> we(geom.nwin, 0), ws(geom.nwin, 0), wruns(geom.nwin, 0), be(pb->nbins, 0), bs(pb->nbins, 0);
        for (uint32_t r = 0; r < nrun; r++) {
            const uint32_t c = hk[r] >> binbits, b = hk[r] & binmask, q = win_of(geom, hc[c]);
            we[q] += hl[r]; ws[q] += ho[r]; wruns[q]++; be[b] += hl[r]; bs[b] += ho[r];
        }
        // bins by slot count, heaviest first: class 0 = the heaviest 1 %, 1 = the next 9 %, 2 = the rest
        std::vector<uint32_t> border(pb->nbins);
        for (uint32_t b = 0; b < pb->nbins; b++) border[b] = b;
        std::sort(border.begin(), border.end(), [&](uint32_t a, uint32_t b) { return bs[a] > bs[b]; });
        std::vector<uint8_t> bclass(pb->nbins, 2);
        for (uint32_t i = 0; i < pb->nbins; i++) bclass[border[i]] = i < (pb->nbins + 99) / 100 ? 0 : i < (pb->nbins + 9) / 10 ? 1 : 2;
        const uint32_t edges_[] = {0, 1, 4, 16, 64, 256, geom.ndw};   // window-rank classes among the dense windows, then the sparse ones
        struct Cls { const char *name; uint32_t lo, hi; } cls[8];
        int ncl = 0;
        static char names[8][40];
        for (int i = 0; i + 1 < 7; i++) {
            const uint32_t lo = std::min(edges_[i], geom.ndw), hi = std::min(edges_[i + 1], geom.ndw);
            if (hi <= lo) continue;
            snprintf(names[ncl], sizeof(names[ncl]), "dense windows %u..%u", lo, hi - 1);
            cls[ncl] = Cls{names[ncl], lo, hi}; ncl++;
        }
        snprintf(names[ncl], sizeof(names[ncl]), "sparse windows (%u)", geom.nwin - geom.ndw);
        cls[ncl] = Cls{names[ncl], geom.ndw, geom.nwin}; ncl++;
        std::vector<uint64_t> ce(ncl * 3, 0), cs(ncl * 3, 0);
        for (uint32_t r = 0; r < nrun; r++) {
            const uint32_t c = hk[r] >> binbits, b = hk[r] & binmask, q = win_of(geom, hc[c]);
            for (int k = 0; k < ncl; k++) if (q >= cls[k].lo && q < cls[k].hi) { ce[k * 3 + bclass[b]] += hl[r]; cs[k * 3 + bclass[b]] += ho[r]; }
        }
        fprintf(stderr, "[pb] slots by window rank (value-stream round trip = 10 B per slot; fusing a window into phase 2 = one staging of %u B per row bin = %.1f MB of L2->LDS per window)\n",
                W * 4, (double)pb->nbins * W * 4 / 1e6);
        fprintf(stderr, "[pb] %-28s %8s %12s %12s %8s %10s %12s | slots in the heaviest 1%% / next 9%% / other bins\n", "class", "windows", "entries", "slots", "e/slot", "saved MB", "staged MB");
        uint64_t te = 0, ts = 0;
        for (int k = 0; k < ncl; k++) {
            uint64_t e = 0, sl = 0;
            for (uint32_t q = cls[k].lo; q < cls[k].hi; q++) { e += we[q]; sl += ws[q]; }
            te += e; ts += sl;
            fprintf(stderr, "[pb] %-28s %8u %12llu %12llu %8.2f %10.1f %12.1f | %llu / %llu / %llu\n", cls[k].name, cls[k].hi - cls[k].lo, (unsigned long long)e, (unsigned long long)sl,
                    sl ? (double)e / sl : 0.0, sl * 10.0 / 1e6, (double)(cls[k].hi - cls[k].lo) * pb->nbins * W * 4 / 1e6,
                    (unsigned long long)cs[k * 3], (unsigned long long)cs[k * 3 + 1], (unsigned long long)cs[k * 3 + 2]);
        }
        fprintf(stderr, "[pb] %-28s %8u %12llu %12llu\n", "all", geom.nwin, (unsigned long long)te, (unsigned long long)ts);
        for (uint32_t q = 0; q < std::min(geom.ndw, 8u); q++)
            fprintf(stderr, "[pb] window %u: %llu entries, %llu slots (%.1f per slot), %llu runs\n", q, (unsigned long long)we[q], (unsigned long long)ws[q], ws[q] ? (double)we[q] / ws[q] : 0.0, (unsigned long long)wruns[q]);
        uint64_t b1 = 0, b10 = 0, ball = 0;
        for (uint32_t i = 0; i < pb->nbins; i++) { ball += bs[border[i]]; if (i < (pb->nbins + 99) / 100) b1 += bs[border[i]]; if (i < (pb->nbins + 9) / 10) b10 += bs[border[i]]; }
        fprintf(stderr, "[pb] row bins: %u; the heaviest 1 %% hold %.1f %% of the slots, the heaviest 10 %% %.1f %% (rows are in natural order: every bin mixes hub and tail rows)\n",
                pb->nbins, 100.0 * b1 / std::max<uint64_t>(ball, 1), 100.0 * b10 / std::max<uint64_t>(ball, 1));
    }
    k_scatter_u32<<<grid_for(nrun), TPB, 0, s>>>(order.as<uint32_t>(), kscan.as<uint32_t>(), nrun, pkstart.as<uint32_t>());
    DevBuf binoff; PB_ALLOC(binoff, (uint64_t)(pb->nbins + 1) * 4);
    k_bin_offsets<<<grid_for(pb->nbins + 1), TPB, 0, s>>>(runbin_s.as<uint32_t>(), kscan.as<uint32_t>(), nrun, nout, pb->nbins, binoff.as<uint32_t>());

    // static streams: dense v-order pads read LDS slot PADCOL (the neutral message), sparse ones column 0 of their window;
    // every k-order pad targets the dummy accumulator row R
    const uint64_t ngroups = ((uint64_t)np + 255) / 256;
    PB_MALLOC(pb->LCOL, (uint64_t)np * 2); PB_MALLOC(pb->LROW, (uint64_t)std::max(nout, 4u) * 2);
    PB_MALLOC(pb->G, ngroups * sizeof(GroupRec));
    PB_MALLOC(pb->KSTART, (uint64_t)(nrun + 64) * 4);
    k_chunk_defaults<<<nchunks, TPB, 0, s>>>(pb->cv0, pb->cv1, pb->ccol0, geom.dense_end, nullptr, pb->LCOL, (uint16_t)geom.wd);
    k_fill_t<uint16_t><<<grid_for(std::max(nout, 4u)), TPB, 0, s>>>(pb->LROW, std::max(nout, 4u), (uint16_t)R);
    if (g->A) {   // weights travel in the narrowest type that holds the largest one (the reference's converter draws 1..128)
        DevBuf mx; PB_ALLOC(mx, 4); PB_HIP(hipMemsetAsync(mx.p, 0, 4, s));
        k_max_u32<<<grid_for(nnz), TPB, 0, s>>>(g->A, nnz, mx.as<uint32_t>());
        uint32_t wmax = 0; PB_HIP(hipMemcpy(&wmax, mx.p, 4, hipMemcpyDeviceToHost));
        pb->wt_bytes = wmax < 256 ? 1 : wmax < 65536 ? 2 : 4;
        PB_MALLOC(pb->WT, (uint64_t)np * pb->wt_bytes); PB_HIP(hipMemsetAsync(pb->WT, 0, (uint64_t)np * pb->wt_bytes, s));
    }
    k_static_streams<<<grid_for(nnz), TPB, 0, s>>>(skey64, sidx, sid, nnz, binbits, pb->ccol0, g->JI, g->xslot, g->A, vstart.as<uint32_t>(),
                                                   pvstart.as<uint32_t>(), pkstart.as<uint32_t>(), Eb.as<uint32_t>(), Xb.as<uint32_t>(), geom.dense_end,
                                                   pb->LCOL, pb->LROW, pb->WT, pb->wt_bytes, wide ? (uint16_t)0 : HEAD);
    if (wide) k_group_table_wide<<<grid_for(ngroups), TPB, 0, s>>>(pvstart.as<uint32_t>(), pkstart.as<uint32_t>(), Xb.as<uint32_t>(), nrun, np, (GroupRec *)pb->G);
    else k_group_table<<<grid_for(ngroups), TPB, 0, s>>>(pvstart.as<uint32_t>(), pkstart.as<uint32_t>(), Xb.as<uint32_t>(), nrun, np, (GroupRec *)pb->G);
    PB_HIP(hipMemsetAsync(pb->KSTART, 0, (uint64_t)(nrun + 64) * 4, s));
    k_kstart<<<grid_for(nrun), TPB, 0, s>>>(pvstart.as<uint32_t>(), pkstart.as<uint32_t>(), Xb.as<uint32_t>(), nrun, pb->KSTART);
    // phase-2 work list (host: nbins is small)
    std::vector<uint32_t> hoff(pb->nbins + 1);
    PB_HIP(hipMemcpyAsync(hoff.data(), binoff.p, (uint64_t)(pb->nbins + 1) * 4, hipMemcpyDeviceToHost, s));
    PB_HIP(hipStreamSynchronize(s));
    PB_HIP(hipGetLastError());
    std::vector<BinWork> work;
    // slots per phase-2 workgroup: 2^19, but a small graph must still give every CU work (R-MAT 22 has 123 row bins for
    // 256 CUs): at least ~200 workgroups (more, smaller ones lose the fused applicator: R-MAT 22 -7 % at 2^16)
    uint64_t epw = EPW;
    while (epw > (1u << 14) && (uint64_t)nout / epw < 192) epw >>= 1;
    if (getenv("GRAPHTAP_PB_EPW")) epw = 1ull << atoi(getenv("GRAPHTAP_PB_EPW"));
    for (uint32_t b = 0; b < pb->nbins; b++) {
        uint64_t n = hoff[b + 1] - hoff[b];
        if (!n) continue;
        uint32_t parts = (uint32_t)((n + epw - 1) / epw);
        const uint64_t per = ((n + parts - 1) / parts + 255) & ~255ull;   // equal parts (whole 256-slot blocks)
        for (uint32_t i = 0; i < parts; i++) {
            uint64_t a = std::min<uint64_t>(hoff[b] + (uint64_t)i * per, hoff[b + 1]), e = std::min<uint64_t>(a + per, hoff[b + 1]);
            if (a == e) continue;
            work.push_back(BinWork{b, (uint32_t)a, (uint32_t)e, parts == 1 ? 1u : 0u, 0u, 0u, 0u, 0u});
        }
    }
    pb->nwork = (uint32_t)work.size();
    {
        std::vector<uint8_t> single(pb->nbins, 1);   // a bin without entries keeps y = 0: "single" with nothing to do... but
        std::vector<uint32_t> parts(pb->nbins, 0);   // nobody would apply its rows, so only bins with exactly one workgroup count
        for (const BinWork &w : work) parts[w.bin]++;
        for (uint32_t b = 0; b < pb->nbins; b++) {
            single[b] = parts[b] == 1;
            if (single[b]) pb->rows_single += std::min<uint32_t>(R, nr - b * R);
        }
        for (BinWork &w : work) w.single = single[w.bin];
        PB_MALLOC(pb->bin_single, pb->nbins);
        PB_HIP(hipMemcpy(pb->bin_single, single.data(), pb->nbins, hipMemcpyHostToDevice));
        std::vector<uint32_t> split;
        for (uint32_t b = 0; b < pb->nbins; b++) if (!single[b]) split.push_back(b);   // incl. bins without entries: nobody else applies their rows
        pb->nsplit = (uint32_t)split.size();
        PB_MALLOC(pb->split_bins, (uint64_t)std::max<size_t>(split.size(), 1) * 4);
        if (!split.empty()) PB_HIP(hipMemcpy(pb->split_bins, split.data(), split.size() * 4, hipMemcpyHostToDevice));
        if (stats) fprintf(stderr, "[pb] phase 2: %u workgroups for %u bins, %u of %u rows in single-workgroup bins\n", pb->nwork, pb->nbins, pb->rows_single, nr);
    }
    // largest first: the workgroups of a launch are dispatched in order and one fits per CU (128 KiB of LDS), so the
    // heavy (hub) bins must not end up in the last of the ~7 rounds
    if (!getenv("GRAPHTAP_PB_BIN_ORDER"))
        std::stable_sort(work.begin(), work.end(), [](const BinWork &a, const BinWork &b) { return a.k1 - a.k0 > b.k1 - b.k0; });
    {   // parts of the work list (see gt_pb::work_part): one part on graphs without slices
        const uint32_t KP = gt_has_exchange(g) ? std::max<uint32_t>(g->info.x_slices, 1) : 1;
        std::vector<uint32_t> part_of_bin(pb->nbins, 0);
        if (KP > 1 && nr) {
            std::vector<uint32_t> r2c(nr);
            PB_HIP(hipMemcpy(r2c.data(), g->R2C, (uint64_t)nr * 4, hipMemcpyDeviceToHost));
            const uint32_t T = std::max<uint32_t>(g->info.slice_width, 1);
            uint32_t last = 0;
            for (uint32_t b = 0; b < pb->nbins; b++) {
                uint32_t part = last;   // a bin of source rows only (no columns): with its predecessor
                for (uint32_t r = b * R; r < std::min<uint64_t>((uint64_t)(b + 1) * R, nr); r++)
                    if (r2c[r] != 0xFFFFFFFFu) { part = std::min(r2c[r] / T, KP - 1); break; }
                part_of_bin[b] = last = std::max(part, last);   // monotone: slice k is complete once the parts <= k are
            }
        }
        std::stable_sort(work.begin(), work.end(), [&](const BinWork &a, const BinWork &b) { return part_of_bin[a.bin] < part_of_bin[b.bin]; });
        pb->work_part.assign(KP + 1, (uint32_t)work.size()); pb->split_part.assign(KP + 1, pb->nsplit);
        pb->work_part[0] = 0; pb->split_part[0] = 0;
        for (uint32_t k = 1; k < KP; k++) {
            pb->work_part[k] = (uint32_t)(std::lower_bound(work.begin(), work.end(), k, [&](const BinWork &a, uint32_t kk) { return part_of_bin[a.bin] < kk; }) - work.begin());
            uint32_t i = 0;
            std::vector<uint32_t> hs(pb->nsplit);
            if (pb->nsplit) PB_HIP(hipMemcpy(hs.data(), pb->split_bins, (uint64_t)pb->nsplit * 4, hipMemcpyDeviceToHost));
            while (i < pb->nsplit && part_of_bin[hs[i]] < k) i++;   // (the split list ascends by bin, parts ascend with the bins)
            pb->split_part[k] = i;
        }
        if (stats && KP > 1) { fprintf(stderr, "[pb] phase 2 in %u parts (workgroups):", KP); for (uint32_t k = 0; k < KP; k++) fprintf(stderr, " %u", pb->work_part[k + 1] - pb->work_part[k]); fprintf(stderr, "\n"); }
    }
    PB_MALLOC(pb->work, work.size() * sizeof(BinWork));
    PB_HIP(hipMemcpy(pb->work, work.data(), work.size() * sizeof(BinWork), hipMemcpyHostToDevice));
    if (pb->nwork) k_work_chunks<<<grid_for(pb->nwork), TPB, 0, s>>>(pb->work, pb->nwork, kscan.as<uint32_t>(), order.as<uint32_t>(), runkey.as<uint32_t>(), nrun, binbits);
    PB_MALLOC(pb->chunk_active, (uint64_t)nchunks * 4); PB_MALLOC(pb->active_prefix, (uint64_t)(nchunks + 1) * 4);
    {   // chunks are in window order and no window straddles a slice (slice k starts at recv_off[k], a multiple of W)
        const uint32_t K = g->info.x_slices;
        std::vector<uint32_t> hcol(nchunks);
        PB_HIP(hipMemcpy(hcol.data(), pb->ccol0, (uint64_t)nchunks * 4, hipMemcpyDeviceToHost));
        pb->slice_chunk.assign(K + 1, nchunks);
        pb->slice_chunk[0] = 0;
        for (uint32_t k = 1; k < K; k++)
            pb->slice_chunk[k] = (uint32_t)(std::lower_bound(hcol.begin(), hcol.end(), (uint32_t)g->recv_off[k]) - hcol.begin());
        // Launch order. Chunk sizes are skewed and a chunk costs time in proportion to its entries: in window order the 512
        // resident workgroups finish much later than an even split would (list-scheduling model); largest-first fixes that.
        // Dense chunks first, sparse chunks behind them (they run as two kernels).
        std::vector<uint32_t> a(nchunks), b(nchunks), ord(nchunks);
        PB_HIP(hipMemcpy(a.data(), pb->cv0, (uint64_t)nchunks * 4, hipMemcpyDeviceToHost));
        PB_HIP(hipMemcpy(b.data(), pb->cv1, (uint64_t)nchunks * 4, hipMemcpyDeviceToHost));
        for (uint32_t c = 0; c < nchunks; c++) ord[c] = c;
        if (!getenv("GRAPHTAP_PB_COLUMN_ORDER")) {
            auto bigger = [&](uint32_t x, uint32_t y) { return b[x] - a[x] > b[y] - a[y]; };
            if (K == 1) for (int i = 0; i < 4; i++) std::stable_sort(ord.begin() + pb->bound[i], ord.begin() + pb->bound[i + 1], bigger);
            else for (uint32_t k = 0; k < K; k++) std::stable_sort(ord.begin() + pb->slice_chunk[k], ord.begin() + pb->slice_chunk[k + 1], bigger);   // one class, all dense
        }
        PB_MALLOC(pb->launch_order, (uint64_t)std::max(nchunks, 1u) * 4);
        PB_HIP(hipMemcpy(pb->launch_order, ord.data(), (uint64_t)nchunks * 4, hipMemcpyHostToDevice));
        {   // what a phase-1 workgroup needs to know about position i of the launch, in one 16-byte load: {chunk, cv0, cv1, ccol0}
            std::vector<uint32_t> desc((size_t)std::max(nchunks, 1u) * 4, 0);
            for (uint32_t i = 0; i < nchunks; i++) { const uint32_t c = ord[i]; desc[4 * (size_t)i] = c; desc[4 * (size_t)i + 1] = a[c]; desc[4 * (size_t)i + 2] = b[c]; desc[4 * (size_t)i + 3] = hcol[c]; }
            PB_MALLOC(pb->ldesc, desc.size() * 4);
            PB_HIP(hipMemcpy(pb->ldesc, desc.data(), desc.size() * 4, hipMemcpyHostToDevice));
        }
        PB_MALLOC(pb->p1_queue, P1_QUEUES * 4); PB_HIP(hipMemset(pb->p1_queue, 0, P1_QUEUES * 4));
        { int dev = 0; hipDeviceProp_t prop; PB_HIP(hipGetDevice(&dev)); PB_HIP(hipGetDeviceProperties(&prop, dev)); pb->ncu = prop.multiProcessorCount; }
    }
    if (!wide) {   // window statistics for the min programs' hybrid pass (cheap: a few MB)
        pb->nwin = geom.nwin; pb->ndw_ = geom.ndw; pb->dense_end_ = geom.dense_end;
        PB_MALLOC(pb->win_entries, (uint64_t)geom.nwin * 4); PB_MALLOC(pb->win_act, (uint64_t)geom.nwin * 8); PB_MALLOC(pb->xdeg, (uint64_t)std::max(g->x_len, 1u) * 4);
        k_win_total<<<grid_for(geom.nwin), TPB, 0, s>>>(wcount.as<uint32_t>(), geom.nwin, geom.ncls, pb->win_entries);
        k_slot_degrees<<<grid_for(g->x_len), TPB, 0, s>>>(g->JA, g->xcol, g->x_len, pb->xdeg);
    }
    PB_HIP(hipStreamSynchronize(s));
    PB_HIP(hipGetLastError());
    *out = pb;
    return GT_OK;
}

// GRAPHTAP_WINDOW_ACTIVITY=1 (diagnostic; a host round trip per call): for the message vector a streaming pass of a min program
// is about to read, how the stored entries divide over windows by the fraction of them that belongs to ACTIVE columns. Printed to
// stderr; what a per-window choice between the streaming pass and the column-driven SpMSpV would have to work with.
int gt_pb_window_activity_report(const gt_graph *g, const void *x, hipStream_t s, uint32_t iteration) {
    gt_pb *pb = g->pb;
    if (!pb || !pb->win_entries || pb->nnz == 0) return GT_OK;
    WinGeom geom{}; geom.ndw = pb->ndw_; geom.dense_end = pb->dense_end_; geom.nwin = pb->nwin; geom.x_len = g->x_len; geom.ncls = 1; geom.nvwin = pb->nwin;
    GT_HIP(hipMemsetAsync(pb->win_act, 0, (uint64_t)pb->nwin * 8, s));
    k_window_activity<<<grid_for(g->x_len), 256, 0, s>>>((const uint32_t *)x, pb->xdeg, g->x_len, geom, pb->win_act);
    std::vector<unsigned long long> act(pb->nwin); std::vector<uint32_t> tot(pb->nwin);
    GT_HIP(hipMemcpyAsync(act.data(), pb->win_act, (uint64_t)pb->nwin * 8, hipMemcpyDeviceToHost, s));
    GT_HIP(hipMemcpyAsync(tot.data(), pb->win_entries, (uint64_t)pb->nwin * 4, hipMemcpyDeviceToHost, s));
    GT_HIP(hipStreamSynchronize(s));
    const double cut[] = {0.0, 1.0 / 64, 1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0 / 2, 1.01};
    unsigned long long te[8] = {0}, ta[8] = {0}, nw[8] = {0}, all_e = 0, all_a = 0;
    for (uint32_t q = 0; q < pb->nwin; q++) {
        if (!tot[q]) continue;
        const double f = (double)act[q] / tot[q];
        int b = 0;
        if (act[q]) { b = 1; while (b < 7 && f > cut[b]) b++; }
        te[b] += tot[q]; ta[b] += act[q]; nw[b]++; all_e += tot[q]; all_a += act[q];
    }
    fprintf(stderr, "[activity] iteration %u: %llu of %llu entries belong to active columns (%.1f %%); windows by active fraction:\n", iteration, all_a, all_e, 100.0 * all_a / std::max(all_e, 1ull));
    const char *names[] = {"none active", "<= 1/64", "<= 1/32", "<= 1/16", "<= 1/8", "<= 1/4", "<= 1/2", "> 1/2"};
    for (int b = 0; b < 8; b++) if (nw[b]) fprintf(stderr, "[activity]   %-12s %6llu windows, %12llu entries (%5.1f %% of all), %12llu of them active\n", names[b], nw[b], te[b], 100.0 * te[b] / std::max(all_e, 1ull), ta[b]);
    return GT_OK;
}

template <class T, class TV, class TX, bool WEIGHTED, bool IS_MIN, class WTy = uint32_t, bool WIDE = false>
static int pb_run(const gt_graph *g, gt_pb *pb, const TX *x, T *y, hipStream_t s, const void *owner, uint64_t epoch,
                  uint32_t slice_lo, uint32_t slice_hi, unsigned phases, const gt_pr_epilogue *epi, bool skip_source, uint32_t part_lo, uint32_t part_hi) {
    // Activity filtering needs VAL to belong to one program between two of its initialize() calls (see stage_window).
    const bool filter = IS_MIN && owner != nullptr && !getenv("GRAPHTAP_NO_ACTIVITY_FILTERING");
    if (phases & GT_PB_PREPARE) {
        if (filter && (pb->val_owner != owner || pb->val_epoch != epoch)) { pb->val_min = -1; pb->val_owner = owner; pb->val_epoch = epoch; }
        if (!filter) pb->val_owner = nullptr;
        if (pb->val_min != (IS_MIN ? 1 : 0)) {   // pad slots (and, with filtering, every slot) start at the semiring's neutral value
            k_fill_t<TV><<<grid_for(pb->nout), TPB, 0, s>>>((TV *)pb->VAL, pb->nout, IS_MIN ? (TV)GT_INF : (TV)0);
            pb->val_min = IS_MIN ? 1 : 0;
        }
    }
    static const bool phase_timing = getenv("GRAPHTAP_PB_PHASE_TIMING") != nullptr;
    const bool pt = phase_timing && phases == (GT_PB_PREPARE | GT_PB_PHASE1 | GT_PB_PHASE2);
    if (pt) {
        while (pb->pt_ev.size() < pb->pt_used + 3) { hipEvent_t e; GT_HIP(hipEventCreate(&e)); pb->pt_ev.push_back(e); }
        GT_HIP(hipEventRecord(pb->pt_ev[pb->pt_used], s));
    }
    const uint8_t *win_mode = nullptr;
    if constexpr (IS_MIN) {
        // HYBRID pass: in the middle iterations of SSSP / CC the frontier is millions of columns, yet whole hub windows hold only a
        // few active ones (R-MAT-26, SSSP iteration 4: 19 windows with 37 % + 19 % of all entries have none / at most 1/64 of them
        // in active columns; profiles/r04/window_activity_sssp_cc.txt). One pass over x counts the entries of the active columns
        // per window; the windows at or below 1/32 (bounded to nnz/64 entries in all) hand those columns to a column-driven SpMSpV
        // (atomicMin on y) and their chunks leave the streaming pass. No host round trip: every count stays on the device.
        // BUILT, MEASURED, OFF BY DEFAULT (GRAPHTAP_HYBRID=1 switches it on; profiles/r04/ab_hybrid_pass.txt, two rounds on one box): the
        // pass leaves out what the table promised -- R-MAT-26 SSSP: 406 M of the 4.2 G entries its four streaming passes touch, 1.06 M
        // entries through the column kernels -- and buys 1 %: SSSP R-MAT-26 10.02 -> 9.92 / 9.59 -> 9.47 ms, CC stand-in 6.06 -> 5.89 /
        // 6.16 -> 6.14, CC R-MAT-26 7.67 -> 7.50 / 7.54 -> 7.82, SSSP R-MAT-24 2.82 -> 2.92 (its four extra enqueues per pass cost more
        // than the 47 % of iteration 4 it skips). The hub windows it removes are the CHEAP part of a min pass (12 entries per
        // value-stream slot: little store traffic), and phase 2 still streams every slot of a bin that any chunk fed.
        const char *eh = gt_cfg((const gt_program *)owner, "GRAPHTAP_HYBRID");   // (`owner` of a min program's SpMV is the program)
        const bool hybrid_on = eh && atoi(eh) != 0;   // (read per call: the tests run both ways in one process)
        if (hybrid_on && filter && pb->win_mode && pb->hy_ncand && g->info.x_slices == 1 && !gt_has_exchange(g) &&
            phases == (GT_PB_PREPARE | GT_PB_PHASE1 | GT_PB_PHASE2)) {
            WinGeom geom{}; geom.ndw = pb->ndw_; geom.dense_end = pb->dense_end_; geom.nwin = pb->nwin; geom.x_len = g->x_len; geom.ncls = 1; geom.nvwin = pb->nwin;
            const char *ef = getenv("GRAPHTAP_HYBRID_F");   // (tests: 1 = every candidate window with an active column goes through the column-driven kernels)
            const unsigned long long F = ef ? std::max(1, atoi(ef)) : 32;
            GT_HIP(hipMemsetAsync(pb->hy_cnt, 0, 2 * sizeof(unsigned int), s));
            k_hybrid_windows<<<pb->hy_ncand, 1024, 0, s>>>(pb->hy_cand, (const uint32_t *)x, g->xcol, pb->xdeg, g->x_len, geom, pb->win_entries, F, pb->win_mode, pb->hy_cnt,
                                                            pb->hy_cap, pb->hy_col, pb->hy_val, pb->hy_deg, pb->hy_stat);
            k_hybrid_cols<WEIGHTED><<<std::min<uint32_t>(2048u, std::max<uint32_t>(pb->hy_cap / 32, 1u)), 256, 0, s>>>(pb->hy_col, pb->hy_val, pb->hy_deg, pb->hy_cnt, pb->hy_cap, 1024u, g->JA, g->IA, g->A, (uint32_t *)y, pb->hy_long, pb->hy_cnt + 1);
            k_hybrid_long<WEIGHTED><<<dim3(64, 32), 256, 0, s>>>(pb->hy_long, pb->hy_cnt + 1, pb->hy_col, pb->hy_val, pb->hy_deg, g->JA, g->IA, g->A, (uint32_t *)y);
            win_mode = pb->win_mode;
        }
    }
    if (phases & GT_PB_PHASE1) {
        uint32_t *ca = filter ? pb->chunk_active : nullptr;
        auto scatter = [&](uint32_t c0, uint32_t c1) {
            if (c1 <= c0) return;
            // persistent: one workgroup per slot of the chip (the LDS window allows two per CU with 64 KiB, one with 128), each
            // drawing chunks from a counter of the ring (GRAPHTAP_PB_PERSIST=0: one workgroup per chunk, dispatched in launch order)
            // The plus semirings only: the min programs do not gain (BFS / CC R-MAT-26 +-1 %) or lose (SSSP R-MAT-24 3.09 -> 3.25 ms, R-MAT-26
            // 10.25 -> 10.45; profiles/r04/ab_recheck_after_register_fix.txt) -- their launches are mostly workgroups that leave at once
            // (windows without an active column), which the dispatcher retires faster than a draw and two barriers do.
            const char *pe = getenv("GRAPHTAP_PB_PERSIST");   // (read per launch: the tests run both forms in one process)
            const bool persist_on = !pe && GT_P1_PERSIST_DEFAULT && !IS_MIN;
            const bool persist_forced = pe && (atoi(pe) & 1) != 0;   // (A/B: bit 0 phase 1, bit 1 phase 2; 0 = neither)
            const uint32_t slots = (uint32_t)pb->ncu * ((WIDE || sizeof(TV) == 8) ? 1u : 2u);
            // Not on graphs with an exchange layout: the slices' launches run side by side with RCCL's kernels, which wait for a slot of
            // their own as long as persistent workgroups hold every CU (R-MAT-26 through the exchange at world size 1: 1.86 -> 1.97 ms
            // per SpMV, ab_recheck_after_register_fix.txt; with the registers doubled on top: last slice in after 0.75 -> 1.59 ms,
            // 1.96 -> 2.49 ms per step, bench_rmat26_forced_exchange_persistent_phase1_regression.json).
            constexpr bool can_persist = GT_P1_PERSIST_ALL != 0 || (!IS_MIN && (WIDE || sizeof(TV) == 8));   // (what k_pb_scatter was compiled with)
            const bool persist = can_persist && ((persist_on && !gt_has_exchange(g)) || persist_forced) && pb->p1_queue && slots && c1 - c0 > slots;
            uint32_t *q = persist ? pb->p1_queue + (pb->p1_seq++ % P1_QUEUES) : nullptr;
            k_pb_scatter<T, TV, TX, WEIGHTED, IS_MIN, WTy, WIDE><<<persist ? slots : c1 - c0, P1_THREADS, 0, s>>>(
                pb->cv0, pb->cv1, pb->ccol0, g->x_len, (const C4 *)pb->LCOL, (const WQ<WTy> *)pb->WT, pb->KSTART,
                (const GroupRec *)pb->G, x, (TV *)pb->VAL, ca, (const gt_u32x4 *)pb->ldesc, c0, g->ndw * W, win_mode, q, c1 - c0);
        };
        // one launch: [regular rows: dense, sparse][source rows: dense, sparse]; computation filtering (TCSC_CF) leaves the
        // source rows' chunks out of every iteration but the last
        if (g->info.x_slices == 1) scatter(pb->bound[0], skip_source ? pb->bound[2] : pb->bound[4]);
        else scatter(pb->slice_chunk[slice_lo], pb->slice_chunk[slice_hi]);   // exchange layout: identity x, every window dense
    }
    if (pt) GT_HIP(hipEventRecord(pb->pt_ev[pb->pt_used + 1], s));
    if (phases & GT_PB_PHASE2) {
        if (filter) k_active_prefix<<<1, 1024, 0, s>>>(pb->chunk_active, pb->nchunks, pb->active_prefix);
        const uint32_t *ap = filter ? pb->active_prefix : nullptr;
        // the parts [part_lo, part_hi) of the work list (all of it by default; gt_pb::work_part)
        const uint32_t np_ = (uint32_t)pb->work_part.size() - 1;
        const uint32_t w0 = pb->work_part[std::min(part_lo, np_)], w1 = pb->work_part[std::min(part_hi, np_)];
        const BinWork *wk = pb->work + w0;
        const uint32_t nw = w1 - w0;
        if (nw) {
            // persistent like phase 1: a slot per 128 KiB (f64) / 64 KiB (u32) of accumulators (GRAPHTAP_PB_PERSIST)
            // -- built, measured, OFF (GRAPHTAP_PB_PERSIST=2 / 3 switches it on): the counter closes phase 2's dispatch gaps too (busy integral
            // 91 -> 95 % of the launch) and its workgroups stream that much slower -- phase 2 runs at its mix's HBM rate either way
            const char *pe = getenv("GRAPHTAP_PB_PERSIST");
            const bool persist_forced = pe && (atoi(pe) & 2) != 0;
            const uint32_t slots = (uint32_t)pb->ncu * (sizeof(T) == 8 ? 1u : 2u);
            const bool persist = GT_P2_PERSIST != 0 && persist_forced && pb->p1_queue && slots && nw > slots;
            uint32_t *q = persist ? pb->p1_queue + (pb->p1_seq++ % P1_QUEUES) : nullptr;
            const uint32_t grid = persist ? slots : nw;
            if constexpr (std::is_same<T, double>::value) {
                if (epi && epi->x_f32)
                    k_pb_gather<T, TV, IS_MIN, 1><<<grid, P2_THREADS, 0, s>>>(wk, (const C4 *)pb->LROW, (const V4<TV> *)pb->VAL, g->info.nnzrows, y, ap, *epi, q, nw);
                else if (epi)
                    k_pb_gather<T, TV, IS_MIN, 2><<<grid, P2_THREADS, 0, s>>>(wk, (const C4 *)pb->LROW, (const V4<TV> *)pb->VAL, g->info.nnzrows, y, ap, *epi, q, nw);
                else
                    k_pb_gather<T, TV, IS_MIN, 0><<<grid, P2_THREADS, 0, s>>>(wk, (const C4 *)pb->LROW, (const V4<TV> *)pb->VAL, g->info.nnzrows, y, ap, gt_pr_epilogue{}, q, nw);
            } else {
                k_pb_gather<T, TV, IS_MIN, 0><<<grid, P2_THREADS, 0, s>>>(wk, (const C4 *)pb->LROW, (const V4<TV> *)pb->VAL, g->info.nnzrows, y, ap, gt_pr_epilogue{}, q, nw);
            }
        }
    }
    if (pt) { GT_HIP(hipEventRecord(pb->pt_ev[pb->pt_used + 2], s)); pb->pt_used += 3; }
    GT_HIP(hipGetLastError());
    return GT_OK;
}

// mean duration of phase 1 / phase 2 over the SpMVs recorded since the last reset (GRAPHTAP_PB_PHASE_TIMING=1)
extern "C" int gt_graph_phase_times(gt_graph *g, double *phase1_ms, double *phase2_ms, uint32_t *spmvs, int reset) {
    GT_REQUIRE(g && phase1_ms && phase2_ms && spmvs, GT_ERR_INVALID, "null argument");
    *phase1_ms = *phase2_ms = 0; *spmvs = 0;
    gt_pb *pb = (g->pb_wide && g->pb_wide->pt_used) ? g->pb_wide : g->pb;   // the build the recorded SpMVs ran on
    if (!pb) return GT_OK;
    GT_HIP(hipDeviceSynchronize());
    for (size_t i = 0; i + 2 < pb->pt_used; i += 3) {
        float a = 0, b = 0;
        GT_HIP(hipEventElapsedTime(&a, pb->pt_ev[i], pb->pt_ev[i + 1])); GT_HIP(hipEventElapsedTime(&b, pb->pt_ev[i + 1], pb->pt_ev[i + 2]));
        *phase1_ms += a; *phase2_ms += b; (*spmvs)++;
    }
    if (*spmvs) { *phase1_ms /= *spmvs; *phase2_ms /= *spmvs; }
    if (reset) pb->pt_used = 0;
    return GT_OK;
}

// The value stream is allocated (and touched once) where programs are initialized, never inside the iteration loop: a
// multi-GB hipMalloc stalls for seconds now and then on this pool, and the first streaming pass of a min program -- BFS's
// iteration 1, iteration 0 takes the SpMSpV -- used to pay for it inside execute().
static int pb_reserve_val(gt_pb *pb, uint32_t bytes_per_slot, hipStream_t s) {
    if (!pb || pb->nnz == 0 || pb->val_cap >= bytes_per_slot) return GT_OK;
    const auto t0 = std::chrono::steady_clock::now();
    if (pb->VAL) (void)hipFree(pb->VAL);   // (a long-lived buffer: not from the build's scratch pool)
    pb->VAL = nullptr; pb->val_cap = 0; pb->val_bytes = 0; pb->val_kind = 0;
    const uint64_t bytes = (uint64_t)std::max(pb->nout, 4u) * bytes_per_slot;
    if (hipMalloc(&pb->VAL, bytes) != hipSuccess) { pb->VAL = nullptr; gt_set_error("out of device memory for the value stream (%llu bytes)", (unsigned long long)bytes); return GT_ERR_HIP; }
    GT_HIP(hipMemsetAsync(pb->VAL, 0, bytes, s));   // first touch here, not in the first SpMV
    pb->val_cap = bytes_per_slot; pb->val_allocs++;
    if (getenv("GRAPHTAP_PB_STATS")) {
        GT_HIP(hipStreamSynchronize(s));
        fprintf(stderr, "[build] value stream%s: %.2f GB (%u B x %u slots) allocated and touched in %.1f ms\n", pb->wide ? " (wide build)" : "", bytes / 1e9, bytes_per_slot, pb->nout,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return GT_OK;
}
// both builds: the narrow one at the width asked for, the wide one (4-byte messages only) at 4
int gt_pb_reserve_val(const gt_graph *g, uint32_t bytes_per_slot, hipStream_t s) {
    int st = pb_reserve_val(g->pb, bytes_per_slot, s);
    if (st == GT_OK && g->pb_wide) st = pb_reserve_val(g->pb_wide, 4, s);
    return st;
}
uint32_t gt_pb_val_allocs(const gt_graph *g) { return (g->pb ? g->pb->val_allocs : 0) + (g->pb_wide ? g->pb_wide->val_allocs : 0); }

// `wide`: the build the SpMV in question ran on (gt_pb_uses_wide): the two builds cut their row bins into workgroups independently
static inline const gt_pb *pb_of(const gt_graph *g, bool wide) { return wide && g->pb_wide ? g->pb_wide : g->pb; }
// the SpMVs with 4-byte messages: PageRank's f32 ones, and the min semirings (u32) when the graph was built with GRAPHTAP_PB_WIDE=1
bool gt_pb_uses_wide(const gt_graph *g, int semiring, bool f32_messages) {
    return g->pb_wide != nullptr && ((semiring == GT_PLUS_F64 && f32_messages) || semiring == GT_MIN_U32 || semiring == GT_MINPLUS_U32);
}
const uint8_t *gt_pb_bin_single(const gt_graph *g, bool wide) { const gt_pb *pb = pb_of(g, wide); return pb ? pb->bin_single : nullptr; }
const uint32_t *gt_pb_split_bins(const gt_graph *g, uint32_t *n, bool wide) { const gt_pb *pb = pb_of(g, wide); *n = pb ? pb->nsplit : 0; return pb ? pb->split_bins : nullptr; }
// running totals of the hybrid passes of this graph: [0] passes, [1] entries handed to the column-driven kernels, [2] entries of the
// windows left out of the streaming pass, [3] such windows (diagnostic; tools/bench_apps.py)
extern "C" int gt_graph_hybrid_stats(const gt_graph *g, uint64_t *out4, int reset) {
    GT_REQUIRE(g && out4, GT_ERR_INVALID, "null argument");
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (!g->pb || !g->pb->hy_stat) return GT_OK;
    GT_HIP(hipDeviceSynchronize());
    GT_HIP(hipMemcpy(out4, g->pb->hy_stat, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) GT_HIP(hipMemset(g->pb->hy_stat, 0, 4 * sizeof(unsigned long long)));
    return GT_OK;
}
// launches of the persistent phase 1 on this graph so far, both builds (diagnostic: the tests make sure the form they compare did run)
extern "C" uint64_t gt_graph_persistent_launches(const gt_graph *g) {
    return g ? (uint64_t)(g->pb ? g->pb->p1_seq : 0) + (uint64_t)(g->pb_wide ? g->pb_wide->p1_seq : 0) : 0;
}
uint32_t gt_pb_parts(const gt_graph *g) { return g->pb && !g->pb->work_part.empty() ? (uint32_t)g->pb->work_part.size() - 1 : 1; }
const uint32_t *gt_pb_split_bins_part(const gt_graph *g, uint32_t k, uint32_t *n) {   // the split bins of part k of the phase-2 work list
    *n = 0;
    if (!g->pb || g->pb->split_part.size() < 2 || k + 1 >= g->pb->split_part.size()) return nullptr;
    *n = g->pb->split_part[k + 1] - g->pb->split_part[k];
    return g->pb->split_bins + g->pb->split_part[k];
}
uint32_t gt_pb_rows_single(const gt_graph *g, bool wide) { const gt_pb *pb = pb_of(g, wide); return pb ? pb->rows_single : 0; }
uint64_t gt_pb_source_entries(const gt_graph *g) { return g->pb ? g->pb->nnz_source : 0; }

// A min program's initialize() takes the value stream over: 4-byte slots, every one infinity(), owner and epoch recorded -- what
// the first streaming pass of its execute() would otherwise do first (a fill of 1.6 GB on R-MAT-26: 0.3 ms inside Execute time)
static int pb_claim_val_min(const gt_graph *g, gt_pb *pb, const void *owner, uint64_t epoch, hipStream_t s) {
    if (!pb || pb->nnz == 0 || !pb->VAL) return GT_OK;
    if (pb->val_bytes != 4 || pb->val_kind != 3) { pb->val_bytes = 4; pb->val_kind = 3; }
    if (!pb->wide && !pb->win_mode && pb->nwin && !gt_has_exchange(g)) {   // the buffers of the hybrid pass (pb_run): here, not inside the iteration loop
        // candidates: the windows that hold at least 1/256 of all entries each (GRAPHTAP_HYBRID_MIN_DIV; tests: a huge divisor = every window)
        const char *ed = getenv("GRAPHTAP_HYBRID_MIN_DIV");
        const uint64_t div = ed ? (uint64_t)std::max(1ll, atoll(ed)) : 256;
        std::vector<uint32_t> tot(pb->nwin), cand;
        GT_HIP(hipMemcpy(tot.data(), pb->win_entries, (uint64_t)pb->nwin * 4, hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < pb->nwin; q++) if (tot[q] && (uint64_t)tot[q] * div >= pb->nnz) cand.push_back(q);
        GT_HIP(hipMalloc((void **)&pb->win_mode, pb->nwin)); GT_HIP(hipMemsetAsync(pb->win_mode, 0, pb->nwin, s));   // everything else streams
        pb->hy_ncand = (uint32_t)cand.size();
        if (pb->hy_ncand) {
            pb->hy_cap = (uint32_t)std::min<uint64_t>((uint64_t)pb->hy_ncand * WS, 1u << 26);   // (a listed column is a slot of a candidate window)
            GT_HIP(hipMalloc((void **)&pb->hy_cand, (uint64_t)pb->hy_ncand * 4)); GT_HIP(hipMemcpy(pb->hy_cand, cand.data(), (uint64_t)pb->hy_ncand * 4, hipMemcpyHostToDevice));
            GT_HIP(hipMalloc((void **)&pb->hy_cnt, 2 * sizeof(unsigned int)));
            GT_HIP(hipMalloc((void **)&pb->hy_stat, 4 * sizeof(unsigned long long))); GT_HIP(hipMemsetAsync(pb->hy_stat, 0, 4 * sizeof(unsigned long long), s));
            for (uint32_t **q : {&pb->hy_col, &pb->hy_val, &pb->hy_deg, &pb->hy_long}) GT_HIP(hipMalloc((void **)q, (uint64_t)pb->hy_cap * 4));
        }
    }
    k_fill_t<uint32_t><<<grid_for(pb->nout), TPB, 0, s>>>((uint32_t *)pb->VAL, pb->nout, GT_INF);
    GT_HIP(hipGetLastError());
    pb->val_min = 1; pb->val_owner = owner; pb->val_epoch = epoch;
    return GT_OK;
}

int gt_pb_claim_val_min(const gt_graph *g, const void *owner, uint64_t epoch, hipStream_t s) {
    int st = pb_claim_val_min(g, g->pb, owner, epoch, s);
    if (st == GT_OK && g->pb_wide) st = pb_claim_val_min(g, g->pb_wide, owner, epoch, s);
    return st;
}

int gt_pb_spmv(const gt_graph *g, int semiring, const void *x, void *y, hipStream_t s, bool f32_messages, bool x_is_f32,
               const void *owner, uint64_t epoch, uint32_t slice_lo, uint32_t slice_hi, unsigned phases, const gt_pr_epilogue *epi, bool skip_source,
               uint32_t part_lo, uint32_t part_hi) {
    const bool wide = gt_pb_uses_wide(g, semiring, f32_messages);   // 4-byte PageRank messages on a graph that has the wide build
    gt_pb *pb = wide ? g->pb_wide : g->pb;
    GT_REQUIRE(pb, GT_ERR_STATE, "propagation-blocking structures were not built for this graph");
    if (pb->nnz == 0) return GT_OK;
    const uint32_t K = g->info.x_slices;
    if (slice_hi > K) slice_hi = K;
    if (phases == 0) phases = (slice_lo == 0 ? GT_PB_PREPARE : 0u) | GT_PB_PHASE1 | (slice_hi >= K ? GT_PB_PHASE2 : 0u);
    // one value stream per graph (SpMVs of one graph must not overlap in time); its element type follows the SpMV
    const uint32_t need = (semiring == GT_PLUS_F64 && !f32_messages) ? 8 : 4;
    const int kind = (semiring == GT_PLUS_F64) ? (f32_messages ? 1 : 2) : 3;
    if ((phases & GT_PB_PREPARE) && (pb->val_bytes != need || pb->val_kind != kind)) {
        // programs reserve the stream in initialize() (gt_pb_reserve_val): this allocates only for a bare gt_spmv
        { int st = pb_reserve_val(pb, need, s); if (st != GT_OK) return st; }
        pb->val_bytes = need; pb->val_kind = kind; pb->val_min = -1; pb->val_owner = nullptr;   // re-fill for the new element type
    }
    switch (semiring) {
        case GT_PLUS_F64:
            GT_REQUIRE(!x_is_f32 || f32_messages, GT_ERR_STATE, "f32 message vector with an f64-message SpMV variant");
            if (f32_messages && x_is_f32 && wide) return pb_run<double, float, float, false, false, uint32_t, true>(g, pb, (const float *)x, (double *)y, s, nullptr, 0, slice_lo, slice_hi, phases, epi, skip_source, part_lo, part_hi);
            if (f32_messages && wide) return pb_run<double, float, double, false, false, uint32_t, true>(g, pb, (const double *)x, (double *)y, s, nullptr, 0, slice_lo, slice_hi, phases, epi, skip_source, part_lo, part_hi);
            if (f32_messages && x_is_f32) return pb_run<double, float, float, false, false>(g, pb, (const float *)x, (double *)y, s, nullptr, 0, slice_lo, slice_hi, phases, epi, skip_source, part_lo, part_hi);
            if (f32_messages) return pb_run<double, float, double, false, false>(g, pb, (const double *)x, (double *)y, s, nullptr, 0, slice_lo, slice_hi, phases, epi, skip_source, part_lo, part_hi);
            return pb_run<double, double, double, false, false>(g, pb, (const double *)x, (double *)y, s, nullptr, 0, slice_lo, slice_hi, phases, epi, skip_source, part_lo, part_hi);
        case GT_PLUS_U32: return pb_run<uint32_t, uint32_t, uint32_t, false, false>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, nullptr, 0, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
        case GT_MIN_U32:
            if (wide) return pb_run<uint32_t, uint32_t, uint32_t, false, true, uint32_t, true>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, owner, epoch, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
            return pb_run<uint32_t, uint32_t, uint32_t, false, true>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, owner, epoch, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
        case GT_MINPLUS_U32:
            GT_REQUIRE(pb->WT, GT_ERR_INVALID, "min-plus SpMV needs a weighted graph");
            if (wide && pb->wt_bytes == 1) return pb_run<uint32_t, uint32_t, uint32_t, true, true, uint8_t, true>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, owner, epoch, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
            if (wide && pb->wt_bytes == 2) return pb_run<uint32_t, uint32_t, uint32_t, true, true, uint16_t, true>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, owner, epoch, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
            if (wide) return pb_run<uint32_t, uint32_t, uint32_t, true, true, uint32_t, true>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, owner, epoch, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
            if (pb->wt_bytes == 1) return pb_run<uint32_t, uint32_t, uint32_t, true, true, uint8_t>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, owner, epoch, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
            if (pb->wt_bytes == 2) return pb_run<uint32_t, uint32_t, uint32_t, true, true, uint16_t>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, owner, epoch, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
            return pb_run<uint32_t, uint32_t, uint32_t, true, true, uint32_t>(g, pb, (const uint32_t *)x, (uint32_t *)y, s, owner, epoch, slice_lo, slice_hi, phases, nullptr, false, part_lo, part_hi);
        default: gt_set_error("unknown semiring %d", semiring); return GT_ERR_INVALID;
    }
}
