/*
 * graphtap_amd.h -- C ABI of the MI355X-native SpMV vertex-program engine.
 *
 * Drop-in boundary for GraphTap's src/mat + src/vp hot path. The reference has
 * no FFI of its own (its operator API is the C++ template Vertex_Program<>,
 * /root/reference/src/vp/vertex_program.hpp:23-209, with virtual per-edge
 * hooks that cannot run on a device); the seam is therefore one level up: a
 * graph handle (replaces Graph<>::load + Matrix + Compressed_column) and a
 * program handle whose hooks are op-codes (replaces Vertex_Program<>::
 * initialize / execute / V). Each entry point cites what it replaces.
 *
 * Conventions: plain pointers and sizes, no C++/torch types. Every function
 * returns GT_OK (0) or a negative gt_status and never calls exit() (the
 * reference prints to stderr and exits, src/mat/graph.hpp:143-144);
 * gt_last_error() gives the message of the calling thread's last failure.
 * Handles are not re-entrant: one caller per handle. All device work of a
 * handle is ordered on the HIP stream given to gt_set_stream() (default: the
 * null stream); calls return without waiting for the device unless they copy
 * to the host or say otherwise.
 *
 * include/graphtap_amd.hpp layers Vertex_Program-shaped C++ classes over this
 * ABI; graphtap_amd/ (Python, ctypes) does the same for tests and bench.py.
 */
#ifndef GRAPHTAP_AMD_H
#define GRAPHTAP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GT_ABI_VERSION 3 /* 2: gt_exec_stats grew (round 2: five fields; round 3: allocs_in_execute), gt_dist_* and gt_spmv_cf added; 3 (round 4): GT_ERR_TIMEOUT, gt_diag_hbm_ceiling, gt_graph_options / gt_program_options */
#define GT_INF 2147483647u /* apps/bfs.h:12 */

typedef enum gt_status {
    GT_OK = 0,
    GT_ERR_INVALID = -1,    /* bad argument / vertex id out of range */
    GT_ERR_HIP = -2,        /* a HIP runtime call failed */
    GT_ERR_NO_DEVICE = -3,  /* no gfx950 device visible: the product has no CPU fallback */
    GT_ERR_UNSUPPORTED = -4,
    GT_ERR_STATE = -5,      /* call order violated */
    GT_ERR_TIMEOUT = -6     /* a wait inside an iteration loop passed its deadline (GRAPHTAP_TIMEOUT_S, default 300 s): the
                               library returns with a message instead of spinning for ever; the handles may only be freed */
} gt_status;

/* program kinds = the reference's five apps (src/apps/{deg,pr,bfs,sssp,cc}.h) */
typedef enum gt_kind { GT_DEG = 0, GT_PR = 1, GT_BFS = 2, GT_SSSP = 3, GT_CC = 4 } gt_kind;
/* Ordering_type, vertex_program.hpp:17-21 */
typedef enum gt_order { GT_ROW = 0, GT_COL = 1 } gt_order;
/* Compression_type subset used by the apps (TCSC / TCSC_CF), ds/compressed_column.hpp */
typedef enum gt_compression { GT_TCSC = 0, GT_TCSC_CF = 1 } gt_compression;
/* generalized SpMV semirings of the five programs (K1/K2/K5 of SURVEY 2.2) */
typedef enum gt_semiring {
    GT_PLUS_F64 = 0,     /* y[r] += x[c]            PageRank, pr.h:39-41          */
    GT_PLUS_U32 = 1,     /* y[r] += x[c]            Degree,   deg.h:43-45         */
    GT_MIN_U32 = 2,      /* y[r] = min(y, x[c])     BFS / CC, bfs.h:61-63         */
    GT_MINPLUS_U32 = 3   /* y[r] = min(y, x[c]+w)   SSSP,     sssp.h:48-51        */
} gt_semiring;

/* implementations of the generalized SpMV over a tile-row (same results, different HBM traffic) */
typedef enum gt_spmv_variant {
    GT_SPMV_EDGE = 0, /* one lane per stored entry, device atomics on y (correctness baseline)      */
    GT_SPMV_PB = 1,   /* propagation blocking: LDS-staged messages + LDS row-bin accumulators; f64 messages everywhere (default) */
    GT_SPMV_PB_F32MSG = 2 /* same, a PageRank PROGRAM's messages rounded to f32 in flight in fixed-count runs (sums, ranks and y
                             stay f64; max relative rank error 5e-8 against the fp64 oracle, tolerance 1e-6); converge mode
                             (gt_program_prepare) and the bare gt_spmv keep f64 messages; halves the value stream; integer
                             semirings are unaffected. Opt-in (GRAPHTAP_SPMV=pb_f32msg; bench.py's default): as the default it flipped the
                             sixth printed decimal of a rank between two layouts, and the mains print the reference's lines */
} gt_spmv_variant;

typedef struct gt_graph gt_graph;     /* replaces Graph<> + Matrix<> + tile compressors */
typedef struct gt_program gt_program; /* replaces Vertex_Program<>                       */

/* Arguments 4-8 of Graph::load (mat/graph.hpp:41-43). */
typedef struct gt_graph_flags {
    int32_t directed, transpose, self_loops, acyclic, parallel_edges;
} gt_graph_flags;

typedef struct gt_graph_info {
    uint32_t num_vertices;  /* N as passed                                       */
    uint32_t nrows;         /* N + 1                 (mat/graph.hpp:89-90)        */
    uint32_t tile_height;   /* H = nrows/p + 1       (mat/matrix.hpp:193)         */
    uint32_t rank, nranks;  /* this handle holds tile-row `rank` of the p x p grid */
    uint32_t nnzrows;       /* non-empty rows of the owned segment                */
    uint32_t nnzcols;       /* non-empty columns of the owned segment             */
    uint32_t seg_stride;    /* S = the largest nnzcols over all segments: the global column space [segment][S] that
                               Degree in _COL_ order accumulates (and all-reduces) in has nranks*S slots          */
    uint64_t nnz_local;     /* stored entries in this tile-row                    */
    uint64_t nnz_global;    /* stored entries over all tile-rows (TEPS denominator) */
    uint64_t nnzrows_global, nnzcols_global;
    int32_t weighted;
    uint32_t regular, source_rows, sink_cols; /* owned-segment class counts (matrix.hpp:1125-1144) */
    uint32_t x_slices;      /* K: the exchange of an iteration runs in K slices (1 on one rank; 4 at 2-4 ranks and 2
                               beyond by default, environment GRAPHTAP_X_SLICES at build time)                        */
    uint32_t slice_width;   /* T = ceil(S / K): compressed column j of any segment travels in slice j / T  */
    uint32_t ncols_local;   /* elements of the message vector x this handle's SpMV reads: S on one rank; on several
                               ranks only the columns the tile-row has an entry in, see gt_graph_exchange_plan */
    uint32_t send_elems;    /* elements of the send buffer (0 on one rank)                                 */
} gt_graph_info;

/* Device pointers to the owned tile-row in TCSC form (ds/compressed_column.hpp:287-296).
   Column ids index the message vector x (ncols_local of them): the compressed columns on one rank, the tile-row's
   needed columns in exchange order on several (gt_graph_exchange_plan). */
typedef struct gt_tile_arrays {
    const uint32_t *JA; /* [ncols_local + 1] column pointers                    */
    const uint32_t *IA; /* [nnz_local] compressed row ids                       */
    const uint32_t *A;  /* [nnz_local] weights, or NULL                         */
    const uint32_t *JC; /* [nnzcols]  compressed col -> segment-local vertex id */
    const uint32_t *IR; /* [nnzrows]  compressed row -> segment-local vertex id */
    const uint32_t *L2G;/* [ncols_local] several ranks: column -> s * seg_stride + j (segment s, compressed
                           column j), UINT32_MAX for padding; NULL on one rank   */
} gt_tile_arrays;

/* The same tile in TCSC_CF form (TCSC_CF_BASE, ds/compressed_column.hpp:419-470), device pointers; one rank only. JA, JC
 * and IR are those of gt_tile_arrays. Inside every column of IA (and A) the entries of SOURCE rows -- rows whose vertex has
 * no column -- sit in the column's tail, in exactly the order the reference's swaps leave them and the regular rows in
 * (:671-708). The four pair lists (:744-1113) are indexed by gt_cf_list: pair q of list k is the range
 * [JA[k][2q], JA[k][2q+1]) of IA, JC[k][q] its compressed column. GT_CF_SRC_R_SNK_C is NC entries long but only its leading
 * pairs are filled, the rest are zero (empty ranges), as in the reference (:1040-1062). */
typedef enum gt_cf_list {
    GT_CF_REG_R_REG_C = 0, /* regular rows of regular columns: every iteration           (vp:1264-1281) */
    GT_CF_REG_R_SNK_C = 1, /* regular rows of sink columns:    iteration 0 only          (vp:1246-1262) */
    GT_CF_SRC_R_REG_C = 2, /* source rows of regular columns:  the last iteration only   (vp:1282-1297) */
    GT_CF_SRC_R_SNK_C = 3  /* source rows of sink columns:     the last iteration only   (vp:1298-1313) */
} gt_cf_list;
typedef struct gt_tile_cf_arrays {
    const uint32_t *IA;             /* [nnz_local]                                         */
    const uint32_t *A;              /* [nnz_local] weights moved with their entries, or NULL */
    const uint32_t *JA_REG_R_NNZ_C; /* [2 * nnzcols] regular rows of every column (:713-742) */
    uint32_t NC[4];
    const uint32_t *JA[4];          /* [2 * NC[k]] */
    const uint32_t *JC[4];          /* [NC[k]]     */
} gt_tile_cf_arrays;

/* The exchange of the message vector between ranks, replacing the reference's MPI_Ibcast of every x segment down its
 * column group (vp:843-862, 970-1013). A tile-row only reads the columns it has an entry in (47 % of all non-empty
 * columns at 8 ranks on R-MAT-26), so instead of all-gathering whole segments every rank SENDS to rank d just the
 * messages tile-row d needs: K all-to-alls per iteration (slice k of every segment in the k-th).
 *   send buffer (gt_program_send): slice k starts at send_offset[k]; its nranks blocks, destination-major, hold
 *     send_counts[k*nranks + d] elements for rank d;
 *   message vector x (gt_program_x): slice k starts at recv_offset[k]; its nranks blocks, source-major, hold
 *     recv_counts[k*nranks + s] elements from rank s.
 * Counts are multiples of 4 elements; recv_counts[k][s] on rank r equals send_counts[k][r] on rank s. The arrays
 * live in the graph handle (host memory) until gt_graph_free. */
typedef struct gt_exchange_plan {
    uint32_t nranks, x_slices;
    const uint64_t *send_offset; /* [x_slices + 1] */
    const uint64_t *recv_offset; /* [x_slices + 1] */
    const uint32_t *send_counts; /* [x_slices * nranks] */
    const uint32_t *recv_counts; /* [x_slices * nranks] */
} gt_exchange_plan;

typedef struct gt_program_params {
    int32_t kind;        /* gt_kind                                               */
    int32_t order;       /* gt_order: GT_COL only for GT_DEG (apps/pr.cpp:40)      */
    int32_t compression; /* gt_compression: only changes PageRank's converge mode  */
    uint32_t root;       /* BFS / SSSP root (bfs.cpp:46)                           */
    double alpha;        /* pr.h:13 (0.15)                                         */
    double tol;          /* pr.h:12 (1e-5)                                         */
} gt_program_params;

typedef struct gt_exec_stats {
    uint32_t iterations;   /* value of Vertex_Program::iteration after the call   */
    uint32_t converged;
    double seconds;        /* wall time of the iteration loop, device drained ("Execute time", vp:416-437) */
    double spmv_ms;        /* sum of SpMV kernel durations (HIP events on the handle's stream) */
    uint32_t spmv_launches;
    uint32_t fused_apply_rows; /* PageRank: rows whose applicator ran inside each SpMV launch pair (the row bins one
                                  phase-2 workgroup owns); 0 when nothing was fused */
    /* the reference's -DTIMING record (vp:2134-2152), host wall time of the three phases summed over the
     * iterations of this call; phases are enqueued asynchronously, so they only add up to `seconds` when the
     * library is asked to drain the stream after each phase (environment GRAPHTAP_TIMING=1) */
    double scatter_gather_ms, combine_ms, apply_ms;
    uint32_t spmspv_iterations; /* min programs: iterations whose frontier was small enough for the frontier-driven SpMSpV
                                   (the reference's sparse path, vp:754-784, 1475-1489) instead of the streaming SpMV */
    uint32_t phase_samples;     /* iterations behind the three phase sums (each phase is timed once per iteration)  */
    /* sums of squares (ms^2) of the per-iteration phase times, for the reference's "sum: avg +/- std_dev" record
     * (stats(), vp:2183-2190); like the sums only meaningful with GRAPHTAP_TIMING=1 */
    double scatter_gather_sq, combine_sq, apply_sq;
    uint32_t cf_filtered_iterations; /* PageRank under GT_TCSC_CF: SpMVs that left the entries of source rows out (computation
                                        filtering, compressed_column.hpp:671-708, vp:1264-1317: all but the last iteration) */
    uint32_t list_iterations;        /* BFS / SSSP / CC on one rank: iterations whose three phases all ran on frontier lists (messenger over
                                        the changed vertices, SpMSpV over their columns, apply over the rows it lowered) */
    uint32_t allocs_in_execute;      /* device allocations made inside the iteration loop of this call (0: everything the loop needs
                                        is reserved by initialize(); a bare gt_spmv on a graph no program touched may allocate) */
    uint32_t reserved0;
} gt_exec_stats;

/* state fields for gt_program_copy_state */
typedef enum gt_field {
    GT_F_DEGREE = 0,  /* u32  Deg_State::degree / PR_State::degree (deg.h:21-25) */
    GT_F_RANK = 1,    /* f64  PR_State::rank (pr.h:15-19)                        */
    GT_F_PARENT = 2,  /* u32  BFS_State::parent (bfs.h:23-31)                    */
    GT_F_HOPS = 3,    /* u32  BFS_State::hops                                    */
    GT_F_DISTANCE = 4,/* u32  SSSP_State::distance (sssp.h:21-25)                */
    GT_F_LABEL = 5,   /* u32  CC_State::label (cc.h:21-25)                       */
    GT_F_ACTIVE = 6   /* u8   Vertex_Program::C (vp:161)                         */
} gt_field;

/* ---- library ---------------------------------------------------------- */
int gt_abi_version(void);
const char *gt_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only host; never fails). */
int gt_device_count(void);
/* Binds the calling thread to HIP device `device` (one process per GPU: LOCAL_RANK). */
int gt_set_device(int device);

/* ---- graph: replaces Graph::load (mat/graph.hpp:105-148) ---------------
 * edges: m records <u4 row, u4 col[, u4 weight]> exactly as in the reference's
 * binary files (ds/triple.hpp:10-13), in host memory or (edges_on_device != 0)
 * already in HBM. The per-record flag handling, per-tile sort/dedupe, the
 * non-empty row/column filters and the TCSC build all run on the device
 * (replaces parread_binary :308-372, Matrix::init_tiles matrix.hpp:538-560,
 * init_filtering :813-858 and TCSC_BASE::populate compressed_column.hpp:371-417).
 * rank/nranks: the handle keeps tile-row `rank` of an nranks x nranks grid (1-D row
 * partition, a=1 b=p of SURVEY 8e) laid over a hashed internal id space when
 * nranks > 1 (see gt_graph_vertex_ids); every rank passes the SAME full edge list. The caller may free `edges` after the call returns. */
int gt_graph_build(gt_graph **out, const void *edges, uint64_t m, int edges_on_device, int weighted,
                   uint32_t num_vertices, const gt_graph_flags *flags, int rank, int nranks);
/* The same graph from a DISTRIBUTED edge list (Matrix::distribute, mat/matrix.hpp:693-810: the reference's ranks read the file in
 * parallel and shuffle the triples to their owners): every rank of `dist` passes a share of the records -- any split of the
 * list -- and the build sends each record to the owner(s) of its row(s); the global pieces (non-empty columns, what each peer
 * needs of a rank's columns) come from collectives. Collective over the communicator; rank and nranks are the communicator's. */
typedef struct gt_dist gt_dist;
int gt_graph_build_distributed(gt_graph **out, gt_dist *dist, const void *edges_share, uint64_t m_share, int edges_on_device, int weighted,
                               uint32_t num_vertices, const gt_graph_flags *flags);
/* ---- handle-level configuration (ABI 3) -----------------------------------
 * Everything the environment variables of README.md choose can be chosen PER HANDLE: two graphs (or two programs) of one process
 * may differ, and a caller need not touch its environment. A field left at its "unset" value (what gt_*_options_init writes) falls
 * back to the environment variable named beside it, then to the built-in default. `size` = sizeof the struct as the caller compiled
 * it (fields past it are unset). No counterpart in the reference, whose knobs are compile-time (-DHAS_WEIGHT, tiling.hpp). */
typedef struct gt_graph_options {
    uint32_t size;
    int32_t spmv_variant;      /* gt_spmv_variant; -1 unset: GRAPHTAP_SPMV (edge | pb | pb_f32msg), default GT_SPMV_PB */
    int32_t force_exchange;    /* 1 / 0; -1 unset: GRAPHTAP_FORCE_EXCHANGE (exchange layout on one rank: rehearses the N-rank driver) */
    uint32_t x_slices;         /* K slices of the exchange; 0 unset: GRAPHTAP_X_SLICES, default 4 at 2-4 ranks, 2 beyond */
    int32_t hubs_first;        /* 1 / 0; -1 unset: GRAPHTAP_PB_HUBS, default 1 (single rank) */
    uint32_t hub_min_degree;   /* entries from which a column is a hub; 0 unset: GRAPHTAP_PB_HUB_DEG, default 8 / 12 / 24 */
    uint32_t exchange_hub_min; /* the same inside a block of the exchange layout; 0 unset: GRAPHTAP_EXCHANGE_HUB_MIN, default 8 */
    uint32_t chunk_log2;       /* log2 entries per phase-1 chunk; 0 unset: GRAPHTAP_PB_CH, default by size (<= 19) */
    int32_t wide_windows;      /* 1 always / 0 never: the WIDE propagation-blocking build beside the narrow one (gt_graph_has_wide_build);
                                  -1 unset: GRAPHTAP_PB_WIDE, default for GT_SPMV_PB_F32MSG graphs of ~0.47 G entries (R-MAT-25) and more on one rank */
    uint32_t reserved[7];
} gt_graph_options;
typedef struct gt_program_options {
    uint32_t size;
    int32_t frontier_lists;    /* 1 / 0; -1 unset: GRAPHTAP_FRONTIER_LISTS, default 1 */
    int32_t spmspv;            /* 1 always / 0 never; -1 unset: GRAPHTAP_SPMSPV, default by size */
    int32_t tail_kernel;       /* 1 / 0; -1 unset: GRAPHTAP_TAIL_KERNEL, default 1 */
    int32_t bfs_bottom_up;     /* 1 always / 0 never; -1 unset: GRAPHTAP_BFS_BOTTOM_UP, default by size */
    int32_t cc_first;          /* 1 / 0; -1 unset: GRAPHTAP_CC_FIRST, default 1 */
    int32_t fuse_apply;        /* 1 / 0; -1 unset: GRAPHTAP_FUSE_APPLY, default 1 */
    int32_t lean_state;        /* 1 / 0; -1 unset: GRAPHTAP_PR_LEAN_STATE, default 1 */
    int32_t hybrid;            /* 1 / 0; -1 unset: GRAPHTAP_HYBRID, default 0 */
    uint32_t spmspv_fraction;  /* the SpMSpV takes frontiers of up to nnz / this entries; 0 unset: GRAPHTAP_SPMSPV_FRACTION, default 32 */
    uint32_t tail_list_max;    /* longest list the one-launch tail takes; 0 unset: GRAPHTAP_TAIL_LIST, default 4096 */
    uint32_t tail_entries_max; /* most entries of its columns; 0 unset: GRAPHTAP_TAIL_ENTRIES, default 2^17 */
    uint32_t reserved0;
    double timeout_s;          /* deadline of every wait of execute(); <= 0 unset: GRAPHTAP_TIMEOUT_S, default 300 */
    uint32_t reserved[8];
} gt_program_options;
void gt_graph_options_init(gt_graph_options *o);
void gt_program_options_init(gt_program_options *o);
/* gt_graph_build / gt_graph_build_distributed (dist != NULL) with options; opt == NULL is the plain call */
int gt_graph_build_opt(gt_graph **out, gt_dist *dist, const void *edges, uint64_t m, int edges_on_device, int weighted,
                       uint32_t num_vertices, const gt_graph_flags *flags, int rank, int nranks, const gt_graph_options *opt);
int gt_graph_info_get(const gt_graph *g, gt_graph_info *info);
/* 1 when the graph also carries the WIDE propagation-blocking build (windows of twice the width, used by SpMVs with 4-byte PageRank
 * messages; built for GT_SPMV_PB_F32MSG graphs of ~0.47 G entries (R-MAT-25) and more on one rank, or on request: GRAPHTAP_PB_WIDE), else 0 */
int gt_graph_has_wide_build(const gt_graph *g);
/* Picks the SpMV implementation used by gt_spmv and by every program of this graph (default
 * GT_SPMV_PB; GT_SPMV_PB_F32MSG / GT_SPMV_EDGE when the environment has GRAPHTAP_SPMV=pb_f32msg / edge at build time). */
int gt_graph_select_spmv(gt_graph *g, int variant);
int gt_graph_tile(const gt_graph *g, gt_tile_arrays *arrays);
/* Builds the TCSC_CF form on first use (Matrix::init_tcsc_cf, mat/matrix.hpp:1370-1403); the arrays live until gt_graph_free. */
int gt_graph_tile_cf(gt_graph *g, gt_tile_cf_arrays *arrays);
int gt_graph_exchange_plan(const gt_graph *g, gt_exchange_plan *plan);
/* Original vertex id of the first `count` (<= tile_height) state slots of this handle, UINT32_MAX for a padding
 * slot. On one rank slot i is vertex i (the reference's layout, vp:1805-1808). On several ranks the owned
 * segment is a contiguous range of a hashed internal id space (load balance under degree skew: tile-row 0 of 8
 * would hold 44 % of R-MAT-26), so callers that assemble a global V use this map; results per vertex are the
 * same for every rank count. */
int gt_graph_vertex_ids(const gt_graph *g, uint32_t *host_out, uint64_t count);
/* Graph::free (mat/graph.hpp:76-81). Programs borrow the graph: free them first. */
/* Diagnostics: with GRAPHTAP_PB_PHASE_TIMING=1 in the environment every whole propagation-blocking SpMV records HIP events
 * around its two phases; this returns their mean durations since the last reset (zeros when nothing was recorded). */
int gt_graph_phase_times(gt_graph *g, double *phase1_ms, double *phase2_ms, uint32_t *spmvs, int reset);
int gt_graph_free(gt_graph *g);

/* ---- programs: replaces Vertex_Program<> ------------------------------ */
/* ctor, vertex_program.hpp:214-330 (stationary / gather_depends_on_apply /
 * apply_depends_on_iter are implied by `kind` exactly as the apps set them). */
int gt_program_create(gt_program **out, gt_graph *g, const gt_program_params *params);
/* Options of one program (see gt_program_options): before or after initialize(); they take effect from the next execute(). */
int gt_program_set_options(gt_program *p, const gt_program_options *opt);
/* initialize(), vp:443-464 */
int gt_program_initialize(gt_program *p);
/* initialize(other), vp:466-501: PageRank takes Deg's degrees where the row is non-empty */
int gt_program_initialize_from(gt_program *p, const gt_program *other);
/* execute(n), vp:408-441; iters == 0 runs until converged. Single-rank graphs only
 * (nranks == 1); multi-rank runs go through gt_dist_execute (RCCL, below) or drive the three phases themselves. */
int gt_program_execute(gt_program *p, uint32_t iters, gt_exec_stats *stats);
/* Work of every later call on this handle is ordered on `hip_stream` (a hipStream_t). */
int gt_program_set_stream(gt_program *p, void *hip_stream);
/* SpMV kernel timing with HIP events recorded on the handle's stream around every SpMV launch of
 * gt_program_combine (the "-DTIMING" combine record of the reference, vp:2134-2152, at kernel
 * granularity). gt_program_timing waits for the stream, returns the sum of the durations and the
 * number of launches since the last reset, and optionally resets. gt_program_execute fills the
 * same numbers into gt_exec_stats by itself. */
int gt_program_enable_timing(gt_program *p, int on);
int gt_program_timing(gt_program *p, double *spmv_ms, uint32_t *launches, int reset);

/* phase level, for the multi-GPU driver (graphtap_amd/dist.py) ------------
 * x is the message vector the local SpMV reads: ncols_local elements (f64 for PageRank -- f32 when the graph's SpMV
 * variant is GT_SPMV_PB_F32MSG at program creation -- u32 otherwise; gt_program_x reports the element width). On one
 * rank it is written by scatter_gather directly. On several ranks scatter_gather fills the SEND buffer
 * (gt_program_send, send_elems elements of the same type) and the driver moves it into every rank's x with the K
 * all-to-alls of gt_graph_exchange_plan. The engine owns default buffers; a caller that exchanges through its own
 * allocations (torch tensors handed to RCCL) installs them with gt_program_set_x / gt_program_set_send.
 * One rank, BFS on a symmetric graph: an iteration that is likely to run bottom-up (it reads vertex states, not messages)
 * does not write x in scatter_gather -- combine does if it takes the push sweep after all; GRAPHTAP_BFS_BOTTOM_UP=0 for
 * callers that want x after every scatter_gather. */
/* First call of every execute(iters) (Vertex_Program::execute, vp:408-413): initializes the program if nobody did, makes
 * converge mode sticky for iters == 0, and fixes the message width of the run. PageRank on a GT_SPMV_PB_F32MSG graph keeps f32
 * messages for fixed iteration counts and switches to f64 ones in converge mode, where the reference's iteration count has to
 * be met on every layout (f32 rounding noise of a hub's rank exceeds pr.h:13's absolute tolerance). gt_program_execute and
 * gt_dist_execute call it themselves; phase-level drivers call it BEFORE gt_program_x / gt_program_send, whose element
 * width it may change (installed caller buffers must be taken back first: gt_program_set_x(p, NULL)). */
int gt_program_prepare(gt_program *p, uint32_t iters);
int gt_program_x(gt_program *p, void **dev_ptr, uint64_t *elems, uint32_t *elem_bytes);
int gt_program_set_x(gt_program *p, void *dev_ptr);
int gt_program_send(gt_program *p, void **dev_ptr, uint64_t *elems, uint32_t *elem_bytes);
int gt_program_set_send(gt_program *p, void *dev_ptr);
/* scatter_gather(), vp:639-758 without the broadcast: the messages of the OWNED segment's columns (into x on one
 * rank, packed per destination into the send buffer on several) */
int gt_program_scatter_gather(gt_program *p);
/* combine(), vp:1017-1113 for the local tile-row (all column segments of x must be current) */
int gt_program_combine(gt_program *p);
/* The same in K = x_slices steps, so that the exchange of slice k+1 can overlap the work on slice k: step k
 * needs only slice k of x (all segments); the accumulators are complete after step K-1. Steps must be issued in
 * order 0..K-1. */
int gt_program_combine_slice(gt_program *p, uint32_t k);
/* Optional, PageRank only: promises that gt_program_apply(p, num_iterations, active != NULL) follows the next complete
 * combine at once (nothing reads y in between). Phase 2 of that combine then applies the rows of the row bins one
 * workgroup owns while their sums are still in LDS; the apply call finishes the rest. Results are the same with or
 * without the call (environment GRAPHTAP_FUSE_APPLY=0 turns it into a no-op); gt_program_execute does this itself. */
int gt_program_fuse_apply(gt_program *p, uint32_t num_iterations, int want_active);
/* apply(), vp:1610-1802, then iteration++. active (nullable) receives the number of
 * owned vertices whose applicator returned true -- the local term of has_converged(),
 * vp:1885-1923 (regular rows only under GT_TCSC_CF); reading it waits for the stream. */
int gt_program_apply(gt_program *p, uint32_t num_iterations, uint64_t *active);
/* the `converged` tail of execute(), vp:425-428 (only PageRank under GT_TCSC_CF is affected) */
int gt_program_finish_converged(gt_program *p);
int gt_program_iteration(const gt_program *p, uint32_t *iteration);
/* In GT_COL order (Deg in apps/pr.cpp) the accumulators live in the GLOBAL column space [segment][seg_stride]
 * (nranks*seg_stride elements) and every rank holds a partial sum; the driver sums them across ranks (the
 * reference's row-group reduce M6, vp:1083-1111) before apply. */
int gt_program_y(gt_program *p, void **dev_ptr, uint64_t *elems, uint32_t *elem_bytes);

/* V, vp:61: copies `count` (<= tile_height) entries of the owned segment to the host */
int gt_program_copy_state(gt_program *p, int field, void *host_out, uint64_t count);
/* checksum(), vp:1927-1960, local part: sum of get_state() and count over owned vertices with
 * state != infinity() and vid < nrows. The reference's accumulator is uint64_t and is `+=`'d with
 * the state, so a double state truncates at EVERY add ("Value checksum: 70" for a true 317.02). */
int gt_program_checksum(gt_program *p, uint64_t *value_sum, uint64_t *reachable);
/* free(), vp:335-405 */
int gt_program_free(gt_program *p);

/* ---- several GPUs, one process (or, for rehearsal, one host thread) per rank: the iteration loop in C++ over RCCL ------
 * Replaces the communication inside Vertex_Program::execute as driven by the reference's C++ mains
 * (src/apps/pr.cpp:15-60): the MPI_Ibcast of x segments (vp:843-862, 970-1013) becomes K rounds of grouped
 * ncclSend / ncclRecv of the NEEDED columns (gt_graph_exchange_plan) on a communication stream, slice k+1 in flight
 * while phase 1 of slice k runs; the row-group reduce (vp:1083-1111) is gone (tile-rows make y complete locally; Degree
 * in GT_COL order is one ncclAllReduce); has_converged (vp:1918) and the checksums (vp:1940,1956) are ncclAllReduce of
 * 8-byte words. graphtap_amd/dist.py is the same loop over torch.distributed. */
#define GT_DIST_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* rank 0: a fresh RCCL unique id, to be handed to every rank by the launcher (file, pipe, MPI_Bcast ...) */
int gt_dist_unique_id(void *id_out);
/* ncclCommInitRank on the calling thread's current device (gt_set_device first); collective over all ranks */
int gt_dist_create(gt_dist **out, const void *unique_id, int rank, int nranks);
/* the same over an ncclComm_t the caller already owns (it stays the caller's) */
int gt_dist_create_from_comm(gt_dist **out, void *nccl_comm, int rank, int nranks);
/* rehearsal transport: the nranks ranks of ONE process on ONE GPU (RCCL refuses two ranks on one device); out[nranks]
 * handles, one per rank; every rank's gt_dist_execute must run on its own host thread */
int gt_dist_create_loopback(gt_dist **out, int nranks);
int gt_dist_free(gt_dist *d);
/* execute(n), vp:408-441, over all ranks: every rank calls it with its own program (same kind, graphs built from the
 * same edge list with rank / nranks of this communicator). iters == 0 runs until converged. */
int gt_dist_execute(gt_dist *d, gt_program *p, uint32_t iters, gt_exec_stats *stats);
/* Bytes this rank's exchanges of x moved since the last reset (to other ranks and to itself), what the same exchanges
 * would have moved as dense blocks, and their number. The min programs ship a (slice, peer) block whose active
 * messages (!= infinity()) are fewer than half of its elements as (index, value) pairs -- the reference's sparse broadcast,
 * vp:970-1013 with the count header of vp:766-773 -- after a count exchange; GRAPHTAP_SPARSE_EXCHANGE=0 keeps every block dense. */
int gt_dist_exchange_stats(gt_dist *d, uint64_t *bytes_sent, uint64_t *bytes_dense, uint64_t *exchanges, int reset);
/* Diagnostics of the last gt_dist_execute that was given a stats pointer. Per iteration (the first 64), in milliseconds by HIP events:
 * [0] messages written and packed, [1] first slice of the exchange landed / [2] last slice landed (after the send buffer was ready;
 * RCCL transport only, -1 otherwise), [3] SpMV span (send buffer ready -> accumulators complete: exchange waits that were not
 * hidden are inside), [4] apply, [5] rest of the iteration (next messages, the all-reduce, the host round trip; converge mode),
 * [6] mode: 0 dense blocks, 1 pairs scattered into x, 2 SpMSpV straight from the pairs. `out` holds 7 doubles per iteration. */
int gt_dist_iteration_times(gt_dist *d, double *out, uint32_t max_iterations, uint32_t *iterations);
/* Ranks of the transport (ncclCommCount for RCCL; the loopback context's size), and counters of the last converge-mode run of a min
 * program: iterations in which every rank's frontier travelled as a list, iterations this rank ran the SpMSpV straight from the
 * pairs, host round trips (one per iteration while the frontiers are lists). Any pointer may be null. */
int gt_dist_info(gt_dist *d, int32_t *transport_ranks, uint64_t *list_iterations, uint64_t *pair_spmspv_iterations, uint64_t *host_round_trips);

/* sum over ranks of `count` (<= 64) host words, in place: checksum() (vp:1940, 1956), nnz_global, display() */
int gt_dist_all_reduce_u64(gt_dist *d, uint64_t *host_values, uint32_t count);

/* ---- kernel-level seam: spmv_stationary / spmv_nonstationary -----------
 * (vp:1116-1327, 1438-1506) y = A (x) x over the handle's tile-row, device pointers.
 * x: ncols_local elements, y: nnzrows elements, both of the semiring's type;
 * y is accumulated into (the caller zero/INF-fills it). Columns whose message is
 * GT_INF are skipped under the two min semirings (vp:1492). */
int gt_spmv(const gt_graph *g, int semiring, const void *x_dev, void *y_dev, void *hip_stream);
/* spmv_stationary over the TCSC_CF pair lists in _ROW_ order (vp:1243-1317), plus-times over f64: regular rows of sink
 * columns when `first_iteration`, regular rows of regular columns when `running` (the program has not converged), source
 * rows when `last_iteration`. x: nnzcols doubles in compressed-column order, y: nnzrows doubles, accumulated into. This is
 * what a PageRank created with GT_TCSC_CF runs on the GT_SPMV_EDGE variant; the propagation-blocking variants apply the
 * same filter to their own streams (gt_exec_stats.cf_filtered_iterations). */
int gt_spmv_cf(gt_graph *g, const void *x_dev, void *y_dev, int first_iteration, int running, int last_iteration, void *hip_stream);

/* ---- synthetic input (no generator in the reference) --------
 * Writes records [first, first+count) of the counter-based R-MAT stream
 * (graphtap_amd/rmat.py is the bit-identical host version) to device memory. */
int gt_rmat_generate(void *dev_out, int scale, uint64_t seed, int weighted, uint64_t first, uint64_t count,
                     void *hip_stream);

/* ---- small device helpers so ctypes callers need no HIP binding -------- */
int gt_malloc(void **dev_ptr, uint64_t bytes);
int gt_free(void *dev_ptr);
int gt_memcpy_h2d(void *dev_dst, const void *host_src, uint64_t bytes);
int gt_memcpy_d2h(void *host_dst, const void *dev_src, uint64_t bytes);
int gt_memset(void *dev_ptr, int value, uint64_t bytes);
int gt_device_synchronize(void);

/* ---- diagnostics ------------------------------------------------------- */
/* This box's HBM streaming rate for one access mix, in GB/s, measured now (graphtap_amd/csrc/diag.hip; persistent grid, 16 B per
 * lane, contiguous span per workgroup, best of three launches over two buffers of `bytes` bytes each, 64 MiB..32 GiB):
 * mode 0 read only (non-temporal loads), 1 write only, 2 copy 1:1, 3 mix of 3 reads : 2 writes (phase 1 of the SpMV),
 * 4 mix 16:1 with nt loads (phase 2 with the lean applicator), 5 mix 7:1 with nt loads (phase 2, full applicator).
 * bench.py reports them beside roofline.achieved; no counterpart in the reference (it has no device). */
int gt_diag_hbm_ceiling(int mode, uint64_t bytes, double *gbps);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHTAP_AMD_H */
