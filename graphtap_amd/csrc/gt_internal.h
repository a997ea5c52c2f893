// gt_internal.h -- shared declarations of the engine's translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <chrono>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/graphtap_amd.h"

void gt_set_error(const char *fmt, ...);

#define GT_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            gt_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return GT_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define GT_REQUIRE(cond, status, ...)                                                       \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            gt_set_error(__VA_ARGS__);                                                      \
            return (status);                                                                \
        }                                                                                   \
    } while (0)

// hipMalloc / hipFree of the build scratch, with the time they take (printed by the [build] lines of GRAPHTAP_PB_STATS)
struct gt_alloc_clock { double malloc_ms = 0, free_ms = 0; uint64_t mallocs = 0, frees = 0, bytes = 0; };
inline gt_alloc_clock &gt_alloc_clock_ref() { static gt_alloc_clock c; return c; }
// The build scratch (ingest + propagation-blocking build: ~54 buffers, 70 GB in all at R-MAT-26, a few alive at a time) comes
// from a POOL: a freed block is kept and handed to the next request it fits (same size or up to 2x larger -- the builds ask for
// the same few sizes, nnz x 4 / 8, again and again), and the pool is emptied when the graph is built (gt_scratch_release).
// A multi-GB hipMalloc stalls for seconds now and then on this pool: the fewer of them, the smaller that exposure.
// Two invariants the pool relies on (both hold for every caller: gt_ingest, gt_layout_build, gt_pb_build, gt_tcsc_cf build):
//  * one build = one host thread from its first scratch allocation to gt_scratch_release (the pool is thread_local ON PURPOSE:
//    the loopback ranks of tests/ build their graphs side by side, one host thread each, and must not hand each other blocks);
//    a block is never freed on another thread than the one that took it;
//  * every kernel that touches a scratch block runs on the NULL stream, so a block handed out again (without hipFree's implicit
//    device sync) is reused strictly after its previous user's kernels in stream order.
struct gt_scratch_pool { struct Blk { void *p; uint64_t bytes; bool busy; }; std::vector<Blk> blocks; uint64_t reused = 0; };
inline gt_scratch_pool &gt_scratch_pool_ref() { static thread_local gt_scratch_pool pool; return pool; }
inline hipError_t gt_scratch_malloc(void **p, uint64_t bytes) {
    gt_scratch_pool &pool = gt_scratch_pool_ref();
    gt_scratch_pool::Blk *best = nullptr;
    for (auto &b : pool.blocks)
        if (!b.busy && b.bytes >= bytes && b.bytes <= 2 * bytes + 4096 && (!best || b.bytes < best->bytes)) best = &b;
    if (best) { best->busy = true; *p = best->p; pool.reused++; return hipSuccess; }
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {   // out of memory with idle blocks in the pool: give them back and try once more
        bool freed = false;
        for (auto it = pool.blocks.begin(); it != pool.blocks.end();) { if (!it->busy) { (void)hipFree(it->p); it = pool.blocks.erase(it); freed = true; } else ++it; }
        if (freed) { (void)hipGetLastError(); e = hipMalloc(p, bytes); }
    }
    gt_alloc_clock &c = gt_alloc_clock_ref();
    c.malloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); c.mallocs++; c.bytes += bytes;
    if (e == hipSuccess) pool.blocks.push_back({*p, bytes, true});
    return e;
}
inline void gt_scratch_free(void *p) {
    if (!p) return;
    gt_scratch_pool &pool = gt_scratch_pool_ref();
    for (auto &b : pool.blocks) if (b.p == p) { b.busy = false; return; }   // stays in the pool until gt_scratch_release
    (void)hipFree(p);
}
// end of a build: every idle block goes back to the driver (blocks still in use -- there should be none -- are left alone)
inline void gt_scratch_release() {
    gt_scratch_pool &pool = gt_scratch_pool_ref();
    const auto t0 = std::chrono::steady_clock::now();
    gt_alloc_clock &c = gt_alloc_clock_ref();
    for (auto it = pool.blocks.begin(); it != pool.blocks.end();) { if (!it->busy) { (void)hipFree(it->p); c.frees++; it = pool.blocks.erase(it); } else ++it; }
    c.free_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Handle-level configuration: (environment variable name, value) pairs set through gt_graph_options / gt_program_options; a site
// that used to call getenv("GRAPHTAP_X") asks the handle first (gt_cfg). Values live as long as the handle.
struct gt_overrides {
    std::vector<std::pair<std::string, std::string>> kv;
    void set(const char *name, const std::string &value) { for (auto &e : kv) if (e.first == name) { e.second = value; return; } kv.emplace_back(name, value); }
    const char *get(const char *name) const { for (const auto &e : kv) if (e.first == name) return e.second.c_str(); return nullptr; }
};

// Owned tile-row of the reference's p x p grid, in HBM.
struct gt_graph {
    gt_overrides cfg;
    gt_graph_info info{};
    gt_graph_flags flags{};
    // TCSC arrays of the tile-row (ds/compressed_column.hpp:287-296)
    uint32_t *JA = nullptr;   // [ncols_total + 1] over the tile-row's column space (see ncols_total)
    uint32_t *IA = nullptr;   // [nnz_local]
    uint32_t *A = nullptr;    // [nnz_local] or null
    uint32_t *JI = nullptr;   // [nnz_local] column id of every entry (edge-parallel kernels)
    uint32_t *JC = nullptr;   // [nnzcols]   owned segment: compressed col -> local vertex
    uint32_t *IR = nullptr;   // [nnzrows]   owned segment: compressed row -> local vertex
    // owned-segment filters (mat/matrix.hpp:861-1122): bit0 = row non-empty (I), bit1 = col non-empty (J)
    uint8_t *IJ = nullptr;    // [H]
    uint32_t *IV = nullptr;   // [H] local vertex -> compressed row
    uint32_t *JV = nullptr;   // [H] local vertex -> compressed col
    uint32_t *R2C = nullptr;  // [nnzrows] compressed row -> compressed col of the same vertex, or ~0u
    // Column space of the tile-row = index space of the message vector x the SpMV reads. One rank: the compressed
    // columns themselves (ncols_total = seg_stride). Several ranks: only the columns this tile-row has an entry in,
    // ordered [slice k][source segment s][ascending compressed column]; slice k occupies [recv_off[k], recv_off[k+1])
    // (starts are multiples of GT_PB_WINDOW) and its p blocks are exactly what the k-th all-to-all of an iteration
    // delivers (ingest.hip).
    uint32_t ncols_total = 0;
    uint32_t *loc2glob = nullptr;  // [ncols_total] local column -> s * seg_stride + j, ~0u for padding (several ranks)
    uint32_t *send_idx = nullptr;  // [send_elems] owned compressed column whose message is element i of the send buffer
    uint64_t send_elems = 0;
    std::vector<uint32_t> send_counts, recv_counts;  // [K][nranks] elements per block (multiples of 4)
    std::vector<uint64_t> send_off, recv_off;        // [K + 1] first element of slice k in the send buffer / in x
    // Balanced relabelling for multi-rank graphs (identity when nranks == 1): internal id = (vid * perm_a) & perm_mask,
    // vid = (internal * perm_ainv) & perm_mask. Segments are contiguous ranges of INTERNAL ids.
    uint32_t perm_a = 1, perm_ainv = 1, perm_mask = 0xFFFFFFFFu, nint = 0;  // nint = size of the internal id space
    // Layout of the message vector x (pb.hip, gt_layout_build). Identity (x_len = ncols_total, xslot = null) on graphs with an
    // exchange layout and under GRAPHTAP_PB_HUBS=0. Otherwise HUBS FIRST: the nhub columns of largest out-degree occupy the
    // first ndw windows of GT_PB_WINDOW slots (degree descending), the other columns follow in ascending compressed order
    // from slot ndw * GT_PB_WINDOW on, in windows of GT_PB_SPARSE_WINDOW slots. Same-row entries of the hub windows then
    // pre-aggregate several times better (R-MAT-26: 1.67 -> 2.3+ entries per (window, row) pair, tools/layout_stats.py).
    uint32_t x_len = 0;            // slots of x
    uint32_t ndw = 0;              // dense (aggregating, GT_PB_WINDOW wide) windows; the rest are sparse windows
    uint32_t *xslot = nullptr;     // [nnzcols] compressed column -> slot of x, or null = identity
    uint32_t *xcol = nullptr;      // [x_len]   slot -> compressed column (inverse of xslot), ~0u for an unused slot, or null = identity
    uint32_t *XV = nullptr;        // [x_len]   slot -> local vertex (what JC is for the identity layout), ~0u for an unused slot
    uint32_t *R2X = nullptr;       // [nnzrows] compressed row -> slot of the same vertex's column, ~0u if none (R2C through xslot)
    void *x_scratch = nullptr;     // [x_len] x 8 B: gt_spmv's copy of a caller's compressed-order x in slot order
    bool force_exchange = false;  // GRAPHTAP_FORCE_EXCHANGE: exchange layout on a single rank (rehearses the N-rank driver path)
    struct gt_pb *pb = nullptr;  // propagation-blocking structures (pb.hip)
    struct gt_pb *pb_wide = nullptr;   // the same with windows of twice the width, for SpMVs with 4-byte messages (pb.hip, gt_pb_build)
    struct gt_tcsc_cf *cf = nullptr;  // the tile in TCSC_CF form, built on first use (tcsc_cf.hip)
    int spmv_variant = 1;        // gt_spmv_variant
    uint64_t serial = 0;         // unique per built graph (a freed graph's address may come back): what a communicator remembers having checked
};

// the value of knob `name` for a graph / a program: the handle's own setting, else (programs) its graph's, else the environment
inline const char *gt_cfg(const gt_graph *g, const char *name) { const char *v = g ? g->cfg.get(name) : nullptr; return v ? v : getenv(name); }
// true when the message vector lives in the LOCAL column space filled by an exchange (several ranks, or forced)
inline bool gt_has_exchange(const gt_graph *g) { return g->loc2glob != nullptr; }
// messengers walk the slots of x: slot -> local vertex (~0u = unused slot), and the row -> slot map of PageRank's applicator
inline const uint32_t *gt_x_vertex(const gt_graph *g) { return g->XV ? g->XV : g->JC; }
inline uint32_t gt_x_owned(const gt_graph *g) { return g->XV ? g->x_len : g->info.nnzcols; }
inline const uint32_t *gt_row_slot(const gt_graph *g) { return g->R2X ? g->R2X : g->R2C; }

// Replaces Vertex_Program<> (vp:23-209): state in HBM as struct-of-arrays over the owned segment (engine.hip)
struct gt_program {
    gt_overrides cfg;
    double timeout_s = 0;   // gt_program_options::timeout_s (<= 0: GRAPHTAP_TIMEOUT_S / 300 s)
    gt_graph *g = nullptr;
    gt_program_params prm{};
    bool stationary = true;
    bool initialized = false;
    bool converged = false;
    bool check_sticky = false;   // vp:412-413: check_for_convergence is set by execute(0) and never cleared
    uint32_t iteration = 0;
    int semiring = 0;
    hipStream_t stream = 0;
    // V (vp:61) as struct-of-arrays over the owned segment, H entries each
    uint32_t *s0 = nullptr;  // degree | parent | distance | label
    uint32_t *s1 = nullptr;  // hops (BFS)
    double *rank = nullptr;  // PageRank
    uint8_t *C = nullptr;    // vp:161
    // messages / accumulators (vp:159-160)
    void *x_own = nullptr, *x = nullptr, *y = nullptr;
    // several ranks: messages of the owned segment's columns [nnzcols] and their per-destination packing (ingest.hip)
    void *xseg = nullptr, *send_own = nullptr, *send = nullptr;
    uint64_t x_elems = 0, y_elems = 0;
    uint32_t x_bytes = 4, y_bytes = 4;
    unsigned long long *d_active = nullptr;
    void *h_pinned = nullptr;   // 256 pinned host bytes: the few words an iteration reads back (gt_read_back)
    // PageRank working set in compressed-row space (dense, sequential): the V-space arrays above are
    // brought up to date lazily (pr_sync_state) when somebody looks at V
    double *rank_c = nullptr;   // [nnzrows]
    uint32_t *deg_c = nullptr;  // [nnzrows]
    uint8_t *C_c = nullptr;     // [nnzrows]
    bool v_stale = false;       // compressed state is newer than V
    bool x_fresh = false;       // the owned segment of x already holds the next iteration's messages
    bool y_clean = false;       // y was zeroed by the fused apply
    std::vector<hipEvent_t> ev;  // SpMV timing pairs
    size_t ev_used = 0;
    double ev_acc_ms = 0; uint32_t ev_acc_pairs = 0;   // pairs folded away when the 64 events were used up (engine.hip, timing_event)
    bool timing = false;
    uint32_t spmv_done = 0;     // complete SpMVs among the timed event pairs (a sliced SpMV records one pair per slice)
    uint64_t init_epoch = 0;    // bumped by every initialize(): scopes the activity filtering of the min programs
    bool x_f32 = false;         // PageRank under GT_SPMV_PB_F32MSG: the message vector itself is f32 (halves the exchange)
    bool f32_capable = false;   // ... which holds for fixed iteration counts; converge mode switches to f64 messages (gt_program_prepare)
    uint32_t x_alloc_bytes = 4; // bytes per element the message buffers were allocated for
    // sliced combine (several ranks): phase 1 of slice k runs on helper stream k % size so that the tail of one slice
    // overlaps the start of the next (and, in the pipelined driver, the exchange of the later slices)
    // gt_program_execute: PageRank's apply of this iteration is fused into phase 2 for the row bins one workgroup owns
    // sparse frontier (min programs): when the active columns hold few entries the SpMV runs as a frontier-driven SpMSpV
    // (kernels.hip) instead of the propagation-blocking pass
    uint32_t *fr_col = nullptr, *fr_val = nullptr, *fr_off = nullptr;   // [fr_cap] active columns, their messages, entry offsets
    void *fr_tmp = nullptr; size_t fr_tmp_bytes = 0;                    // scan scratch
    uint32_t fr_cap = 0;
    unsigned long long *d_frontier = nullptr;                           // [2] active columns, entries in them
    uint64_t last_active = ~0ull;                                       // vertices the previous apply() activated (converge mode), or ~0 if unknown
    uint32_t spmspv_allocs = 0;                                         // allocations gt_spmspv_reserve made since execute() began (0 after initialize reserved)
    uint32_t spmspv_iters = 0;                                          // iterations of the current execute() that took the sparse path
    // FRONTIER LISTS (min programs on one rank, converge mode): every apply() appends the vertices it changes to a list; while that
    // list is short (<= GT_FRONTIER_CAP) the next iteration never walks a full vector -- the messenger resets the previous
    // frontier's slots of x and writes the new ones, the SpMSpV takes its columns from the list and emits the rows it lowers
    // (de-duplicated through row_mark), and apply() visits those rows only (engine.hip, kernels.hip).
    uint32_t fl_cap = 0;                      // min(H, GT_FRONTIER_CAP): elements of each list
    uint32_t *fl_v[2] = {nullptr, nullptr};   // [fl_cap] vertex lists: fl_v[fl_cur] = vertices changed by the last apply
    uint32_t *fl_rows = nullptr;              // [fl_rows_cap] rows the SpMSpV of this iteration lowered
    uint8_t *row_mark = nullptr;              // [nnzrows] 1 = already in fl_rows (zero between iterations)
    unsigned int *d_fl = nullptr;             // [4] device counters: elements of fl_v[0], fl_v[1], fl_rows
    uint32_t fl_rows_cap = 0;
    int fl_cur = 0;
    bool fl_enabled = false, fl_cur_valid = false, fl_prev_valid = false, fl_rows_valid = false;
    uint32_t fl_cur_n = 0, fl_prev_n = 0;     // host copies of the two list lengths (valid lists only)
    uint32_t list_iters = 0;                  // iterations of the current execute() whose three phases all ran on lists
    uint32_t tail_iters = 0;                  // ... of them, inside the persistent tail kernel (kernels.hip, gt_tail_try)
    void *d_tail = nullptr;                   // the tail kernel's result record
    // BOTTOM-UP BFS steps (symmetric graphs): once fewer rows are unreached than vertices are active, the unreached rows look
    // their parent up (minimum id among the neighbours on the current level -- the same value the push sweep leaves in y)
    uint32_t *bu_rows = nullptr;              // [nnzrows] rows of unreached vertices
    uint32_t *bu_bits = nullptr;              // bitmap over the rows: on the current level (same allocation, behind bu_rows)
    uint32_t *bu_long = nullptr;              // [nnzrows] positions in bu_rows of the rows whose first probes found nothing (same allocation)
    uint32_t *bu_first = nullptr;             // [4 nnzrows] the first four entries of every row's column (same allocation, 16-byte aligned; initialize)
    // Row bitmaps kept up to date by BFS's own apply kernels (engine.hip): rows reached so far, rows reached by the LAST apply (= the
    // current level). With them a bottom-up step needs no collecting pass over the rows (0.22-0.30 ms per step on R-MAT-26). Valid
    // from initialize() until something else than those kernels changes the state (the one-launch tail): then the collecting pass.
    uint32_t *bu_reached = nullptr, *bu_next = nullptr;   // same allocation; bu_bits is the level bitmap, bu_next the one a bottom-up step writes (then swapped)
    uint32_t bu_words = 0;                    // words per bitmap
    bool bu_maps_valid = false, bu_step_emitted = false;   // emitted: the bottom-up step of this iteration already wrote the next level
    uint64_t bfs_settled = 0;                 // rows reached so far (host estimate from the active counts)
    bool root_here = false; uint32_t root_local = 0;   // BFS / SSSP: the root's slot in this rank's segment, if it lives here (init_common)
    uint32_t bottom_up_iters = 0;
    // a bottom-up step reads no messages: scatter_gather() defers the messenger when such a step is likely, combine runs it
    // after all if the step is declined, and x is marked stale (the next messenger rewrites all of it) if it was not needed
    bool x_deferred = false, x_stale = false;
    bool rowless_reset = false;   // the messages of vertices without a row were reset once (first apply that wrote messages itself)
    bool pack_deferred = false;   // scatter_gather leaves the per-destination packing to gt_program_pack_slice (the C++ multi-rank driver)
    bool p2_by_parts = false;     // this iteration's phase 2 runs part by part (gt_program_phase2_part): combine's last slice leaves it out
    // TCSC_CF computation filtering: the driver told us which iteration is the last (execute / gt_program_fuse_apply), so the
    // SpMVs before it may leave the source rows' entries out (vp:1264-1317)
    bool cf_hint = false;
    uint32_t cf_filtered = 0;   // SpMVs of the current execute() that left the source rows' entries out
    bool fuse_armed = false, fused = false;   // armed before combine; fused = the combine of this iteration did it
    uint32_t fuse_iters = 0;
    bool fuse_count = false;
    // PageRank, fixed iteration count, the loop owned by the library (gt_program_execute / gt_dist_execute): rank (pr.h:43-47) does
    // not depend on the previous rank, so an iteration that is neither the last nor the one before it only has to produce the
    // next messages -- the stores of rank / C it would make are overwritten before anybody can read them. 0 = full applicator,
    // 1 = writes rank (the last iteration compares against it), 2 = messages only. Set per iteration by gt_pr_state_mode().
    int pr_state = 0;
    std::vector<hipStream_t> slice_streams;
    std::vector<hipEvent_t> slice_in, slice_done;   // per slice: "inputs ready" (recorded on `stream`), "phase 1 done"
    std::vector<hipStream_t> part_streams;          // phase 2 part by part: a stream per part, priority descending (gt_program_phase2_part)
    std::vector<hipEvent_t> part_done;              // ... "part k and the apply of its split bins are done: the messages of slice k are final"
    hipEvent_t p2_go = nullptr;                     // ... "phase 1 is complete" (recorded on `stream`)
};

inline const char *gt_cfg(const gt_program *p, const char *name) { const char *v = p ? p->cfg.get(name) : nullptr; return v ? v : gt_cfg(p ? p->g : nullptr, name); }
#define GT_FRONTIER_CAP (1u << 24)   // longest frontier kept as a list
#ifndef GT_PB_ROW_BIN_BITS
#define GT_PB_ROW_BIN_BITS 14   // log2 rows per phase-2 row bin (pb.hip)
#endif
#ifndef GT_PB_WINDOW
#define GT_PB_WINDOW 16383u   // columns per DENSE phase-1 window (pb.hip): LDS slot GT_PB_WINDOW holds the neutral message that pad
                              // entries read, and a column offset (or the pad) must fit 14 bits; slice widths are multiples of it
#endif
#define GT_PB_SPARSE_WINDOW 16384u   // columns per SPARSE phase-1 window (low-degree columns: nothing to pre-aggregate)

// pb.hip
int gt_layout_build(gt_graph *g);   // x layout (hubs first); before gt_pb_build and before any program exists
int gt_pb_build(gt_graph *g);
void gt_pb_free(struct gt_pb *pb);
// slices [slice_lo, slice_hi) of phase 1; phase 2 runs when slice_hi == x_slices. `phases` = 0 does what the slice range
// implies (prepare with slice 0, phase 1, phase 2 with the last slice); the engine's sliced path passes the stages
// one by one so that the phase-1 launches of different slices can run on different streams.
enum { GT_PB_PREPARE = 1u, GT_PB_PHASE1 = 2u, GT_PB_PHASE2 = 4u };
// PageRank's applicator of iteration t + messenger of iteration t+1 (pr.h:31-33, 43-47) fused into phase 2: a row bin
// that ONE phase-2 workgroup owns has its complete sums in LDS at flush time, so that workgroup applies its rows
// directly and y is never touched for them; the rows of split bins go through y and k_pr_apply_msg as before
// (gt_pb_bin_single tells the two apart).
struct gt_pr_epilogue {
    double *rank_c; const uint32_t *deg_c; uint8_t *C_c; const uint32_t *R2X;   // R2X: row -> slot of x (gt_row_slot), ~0u for a source row
    void *x; int x_f32;
    double alpha, tol; int cf, last;
    unsigned long long *d_active;
    int state;   // gt_program::pr_state: 0 = full, 1 = no read of the old rank and no changed flag, 2 = messages only
};
// which of the three the iteration about to run may use (GRAPHTAP_PR_LEAN_STATE=0: always the full one)
static inline int gt_pr_state_mode(const gt_program *p, uint32_t iters, bool check) {
    const char *e = gt_cfg(p, "GRAPHTAP_PR_LEAN_STATE");   // read per call: the tests run both ways in one process
    const bool lean = !(e && atoi(e) == 0);
    if (!lean || p->prm.kind != GT_PR || check || iters == 0 || p->iteration + 1 >= iters) return 0;
    return p->iteration + 2 == iters ? 1 : 2;
}
// `wide`: of the wide build (the one an SpMV with 4-byte PageRank messages runs on when the graph has it: gt_pb_uses_wide)
bool gt_pb_uses_wide(const gt_graph *g, int semiring, bool f32_messages);
const uint8_t *gt_pb_bin_single(const gt_graph *g, bool wide = false);   // [row bins] 1 = one phase-2 workgroup owns the bin
const uint32_t *gt_pb_split_bins(const gt_graph *g, uint32_t *n, bool wide = false);   // the bins that are NOT single (device list): the only rows the apply kernel visits after a fused combine
uint32_t gt_pb_rows_single(const gt_graph *g, bool wide = false);        // rows of those bins
int gt_pb_spmv(const gt_graph *g, int semiring, const void *x, void *y, hipStream_t s, bool f32_messages, bool x_is_f32,
               const void *owner, uint64_t epoch, uint32_t slice_lo, uint32_t slice_hi, unsigned phases = 0,
               const gt_pr_epilogue *epi = nullptr, bool skip_source = false, uint32_t part_lo = 0, uint32_t part_hi = 0xFFFFFFFFu);
int gt_pb_window_activity_report(const gt_graph *g, const void *x, hipStream_t s, uint32_t iteration);   // GRAPHTAP_WINDOW_ACTIVITY=1 (diagnostic)
uint32_t gt_pb_parts(const gt_graph *g);   // parts of the phase-2 work list (x_slices on a graph with an exchange layout, else 1; pb.hip, gt_pb::work_part)
const uint32_t *gt_pb_split_bins_part(const gt_graph *g, uint32_t k, uint32_t *n);   // the split bins of part k
int gt_pb_reserve_val(const gt_graph *g, uint32_t bytes_per_slot, hipStream_t s);   // allocates + touches VAL (initialize time)
int gt_pb_claim_val_min(const gt_graph *g, const void *owner, uint64_t epoch, hipStream_t s);   // initialize() of BFS / SSSP / CC: VAL all infinity(), owned by that program
uint32_t gt_pb_val_allocs(const gt_graph *g);   // how many times VAL was (re)allocated so far
uint64_t gt_pb_source_entries(const gt_graph *g);   // entries in chunks of source rows (left out by PageRank/TCSC_CF until the last iteration)

// `d` != null: the DISTRIBUTED build -- `edges_dev` holds this rank's share of the records (any share); the records travel to the
// owners of their rows and the global pieces of the build (column flags, what every peer needs of my columns, the entry count)
// come from collectives over `d` (Matrix::distribute, mat/matrix.hpp:693-810)
struct gt_dist;
int gt_ingest(gt_graph *g, const void *edges_dev, uint64_t m, int weighted, gt_dist *d = nullptr);
// collectives of the distributed build (dist.hip; RCCL or the loopback transport)
int gt_dist_exchange_bytes(gt_dist *d, const void *send, const uint64_t *send_off, const uint64_t *send_bytes, void *recv, const uint64_t *recv_off,
                           const uint64_t *recv_bytes, hipStream_t s);
int gt_dist_all_reduce_max_u8(gt_dist *d, uint8_t *buf, uint64_t n, hipStream_t s);
int gt_dist_all_reduce_sum_u64_host(gt_dist *d, uint64_t *v, uint32_t count);
int gt_dist_rank(const gt_dist *d);
int gt_dist_nranks(const gt_dist *d);

// tcsc_cf.hip
int gt_tcsc_cf_build(gt_graph *g);
void gt_tcsc_cf_free(struct gt_tcsc_cf *c);
int gt_tcsc_cf_arrays(gt_graph *g, gt_tile_cf_arrays *a);
// the pair-list SpMV of vp:1243-1317; x in slot order (g->xslot), accumulates into y
int gt_tcsc_cf_spmv(gt_graph *g, const double *x, double *y, bool first, bool running, bool last, hipStream_t s);

// Original vertex id of state slot / internal id u. One rank (identity map): u itself, also for the padding slot
// u = nrows, exactly like the reference's get_vid (vp:1805-1808). Several ranks: ~0u when u is not a vertex.
// A slot holds a real vertex iff gt_vid_of(...) < nrows.
struct gt_vidmap { uint32_t ainv, mask, nint, nrows; };
__host__ __device__ inline uint32_t gt_vid_of(const gt_vidmap &m, uint64_t u) {
    if (m.mask == 0xFFFFFFFFu) return (uint32_t)u;
    if (u >= m.nint) return 0xFFFFFFFFu;
    const uint32_t v = ((uint32_t)u * m.ainv) & m.mask;
    return v < m.nrows ? v : 0xFFFFFFFFu;
}
inline gt_vidmap gt_vidmap_of(const gt_graph *g) { return gt_vidmap{g->perm_ainv, g->perm_mask, g->nint, g->info.nrows}; }

// kernels.hip
// Frontier-driven SpMSpV of the min semirings (the reference's sparse path, vp:754-784 and 1475-1489): y[r] = min(y[r], x[c] (+ w))
// over the entries of the ACTIVE columns only. Counts first; runs only if they hold at most nnz / 1024 entries (or GRAPHTAP_SPMSPV
// forces it); *done tells whether the SpMV is complete.
int gt_spmspv_try(gt_program *p, hipStream_t s, bool *done);
// the messenger over the frontier lists: resets the slots of the previous frontier, writes the messages of the current one
int gt_frontier_messages(gt_program *p, hipStream_t s);
int gt_kernels_preload(hipStream_t s);
int gt_bu_first_neighbours(const gt_graph *g, uint32_t *FN, hipStream_t s);   // fills gt_program::bu_first
int gt_bu_maps_init(gt_program *p, hipStream_t s);   // BFS initialize(): both row bitmaps = the root's row
// A few words from the device, once per iteration (the active count, a frontier's entry count): through the program's pinned
// buffer and a spin on the stream instead of a pageable copy + hipStreamSynchronize (engine.hip)
extern "C" int gt_read_back(gt_program *p, void *dst, const void *src_dev, size_t bytes, hipStream_t s);   // (inside engine.hip's extern "C" block; not part of the ABI header)
// Every wait of an iteration loop goes through this: a spin on the stream with a deadline (GRAPHTAP_TIMEOUT_S, else
// GRAPHTAP_DIST_TIMEOUT_S, default 300 s; read per call). GT_ERR_TIMEOUT + a message when it passes (engine.hip)
extern "C" int gt_stream_wait_deadline(hipStream_t s, const char *what, double limit_s = 0);   // limit_s <= 0: gt_wait_limit_s()
extern "C" double gt_wait_limit_s(void);
bool gt_frontier_list_worth(const gt_program *p, uint64_t n);   // kernels.hip
bool gt_bfs_bottom_up_likely(const gt_program *p);   // host-side part of the bottom-up test (kernels.hip)
bool gt_list_spmspv_likely(const gt_program *p);   // the frontier is a short list and the messenger would be a full pass (kernels.hip)
bool gt_cc_first_likely(const gt_program *p);        // CC's iteration 0 will read first entries instead of sweeping (kernels.hip)
extern "C" int gt_min_messenger(gt_program *p);      // the messenger of BFS / SSSP / CC, now (engine.hip, inside its extern "C" block; not part of the ABI header)   // loads the code object of kernels.hip (called by initialize)
int gt_spmspv_reserve(gt_program *p, uint32_t nact);
int gt_tail_try(gt_program *p, hipStream_t s, bool *converged, uint32_t *iterations_run);
// the SpMSpV over a frontier a driver put into fr_col / fr_val / fr_off (= entry counts) itself (several ranks: the (index, value)
// pairs its peers sent, dist.hip): y lowered with atomics, the rows it lowered left in fl_rows for the row-list apply
int gt_spmspv_run_frontier(gt_program *p, uint32_t nact, hipStream_t s);
// apply() in two halves for a driver that reads the active count itself, together with other words (dist.hip)
extern "C" int gt_program_pack_slice(gt_program *p, uint32_t k);
extern "C" int gt_program_pack_slice_on(gt_program *p, uint32_t k, hipStream_t s);   // ... on a stream of the caller's (the communication stream: off the compute stream's critical path)
extern "C" bool gt_program_parts_begin(gt_program *p);                               // phase 2 part by part (engine.hip): the pipelined multi-rank PageRank loop
extern "C" int gt_program_phase2_part(gt_program *p, uint32_t k, uint32_t num_iterations, int concurrent);
extern "C" int gt_program_apply_begin(gt_program *p, uint32_t num_iterations);
extern "C" int gt_program_apply_end(gt_program *p, uint64_t active_local);
int gt_launch_spmv_edge(const gt_graph *g, int semiring, const void *x, void *y, hipStream_t s);
// owner/epoch: the program (and its initialize() count) issuing the SpMV, or null for a stand-alone gt_spmv; lets the
// min programs skip chunks without an active column (activity filtering)
int gt_launch_spmv(const gt_graph *g, int semiring, const void *x, void *y, hipStream_t s, bool x_is_f32 = false,
                   const void *owner = nullptr, uint64_t epoch = 0, uint32_t slice_lo = 0, uint32_t slice_hi = 0xFFFFFFFFu,
                   unsigned phases = 0, const gt_pr_epilogue *epi = nullptr, bool skip_source = false, bool f64_messages = false,
                   uint32_t part_lo = 0, uint32_t part_hi = 0xFFFFFFFFu);
