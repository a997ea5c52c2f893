/*
 * gt_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the GraphTap reference's algorithm for the hot
 * path (ingest flags -> TCSC -> scatter_gather / combine / apply loop) at
 * np = 1, written from the reference's semantics, not from its text. Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (graphtap_amd/, include/) never does.
 *
 * PARITY IS PINNED: tests/test_oracle_golden.py checks every function below
 * against (a) full vertex-state vectors dumped from the unmodified reference
 * built by oracle/ref/Makefile (tests/golden/*.npz, made by
 * tests/golden/make_golden.py) and (b) the known-answer lines of the
 * reference's own checksum().
 *
 * Citations are file:line under /root/reference/src.
 *
 * Known, documented deviation (results unaffected): with weights the
 * reference sorts a tile by (col, weight) with the unstable std::sort and
 * removes only *adjacent* (row,col) duplicates (ds/triple.hpp:83-92,
 * mat/matrix.hpp:545-555), so which parallel edges survive depends on
 * libstdc++'s introsort. The minimum-weight copy of every (row,col) always
 * survives, and min-plus is idempotent, so labels are identical; we keep
 * exactly one entry per (row,col) with the minimum weight. Only `nnz` (a
 * TEPS denominator) differs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define GTO_INF 2147483647u /* apps/bfs.h:12, sssp.h:12, cc.h:12 */

typedef struct gto_graph {
    uint32_t N;        /* num_vertices argument */
    uint32_t n;        /* nrows = ncols = N + 1        (mat/graph.hpp:89-90) */
    uint32_t H;        /* tile_height = n / 1 + 1      (mat/matrix.hpp:193)  */
    uint64_t nnz;      /* stored entries                                      */
    uint32_t nnzrows, nnzcols;
    int weighted;
    uint32_t *JA;      /* [nnzcols+1] column pointers over compressed cols    */
    uint32_t *IA;      /* [nnz] compressed row ids                            */
    uint32_t *A;       /* [nnz] weights or NULL                               */
    uint32_t *JC;      /* [nnzcols] compressed col -> vertex (colgrp_nnz_columns) */
    uint32_t *IR;      /* [nnzrows] compressed row -> vertex (rowgrp_nnz_rows)    */
    uint8_t  *I, *J;   /* [H] non-empty row / column flags (matrix.hpp:861-1122) */
    uint32_t *IV, *JV; /* [H] vertex -> compressed row / col id               */
} gto_graph;

/* ------------------------------------------------------------------ sort */
/* LSD radix sort of 64-bit keys with a 32-bit payload (stable). */
static void radix_sort_kv(uint64_t *k, uint32_t *v, uint64_t n, int key_bits) {
    uint64_t *k2 = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint32_t *v2 = v ? (uint32_t *)malloc(n * sizeof(uint32_t)) : NULL;
    uint64_t *cnt = (uint64_t *)malloc(65536 * sizeof(uint64_t));
    for (int shift = 0; shift < key_bits; shift += 16) {
        memset(cnt, 0, 65536 * sizeof(uint64_t));
        for (uint64_t i = 0; i < n; i++) cnt[(k[i] >> shift) & 0xFFFF]++;
        uint64_t s = 0;
        for (int b = 0; b < 65536; b++) { uint64_t c = cnt[b]; cnt[b] = s; s += c; }
        for (uint64_t i = 0; i < n; i++) {
            uint64_t p = cnt[(k[i] >> shift) & 0xFFFF]++;
            k2[p] = k[i];
            if (v) v2[p] = v[i];
        }
        uint64_t *tk = k; k = k2; k2 = tk;
        if (v) { uint32_t *tv = v; v = v2; v2 = tv; }
    }
    /* key_bits is always a multiple of 32 here => even number of passes =>
       results are back in the caller's arrays */
    free(k2); free(v2); free(cnt);
}

/* ---------------------------------------------------------------- ingest */
void gto_graph_free(gto_graph *g) {
    if (!g) return;
    free(g->JA); free(g->IA); free(g->A); free(g->JC); free(g->IR);
    free(g->I); free(g->J); free(g->IV); free(g->JV);
    free(g);
}

/*
 * rec: m records of <u4 a, u4 b[, u4 w]> as in the reference's binary edge
 * files (Triple layout, ds/triple.hpp:10-13, 40-46: row first, then col).
 * Per-record handling follows mat/graph.hpp:329-356 in this exact order:
 * drop self loops unless self_loops; acyclic swap; transpose swap; insert;
 * insert the mirrored copy when !directed. Then per tile (one tile here):
 * sort column-major and, unless parallel_edges, drop duplicate (row,col)
 * (mat/matrix.hpp:546-555); build the non-empty row/column filters
 * (matrix.hpp:861-1122) and the TCSC arrays (ds/compressed_column.hpp:371-417).
 */
gto_graph *gto_graph_build(const uint32_t *rec, uint64_t m, int weighted, uint32_t N,
                           int directed, int transpose, int self_loops, int acyclic,
                           int parallel_edges) {
    gto_graph *g = (gto_graph *)calloc(1, sizeof(gto_graph));
    g->N = N; g->n = N + 1; g->H = g->n + 1; g->weighted = weighted;
    const int stride = weighted ? 3 : 2;
    uint64_t cap = directed ? m : 2 * m;
    uint64_t *key = (uint64_t *)malloc((cap ? cap : 1) * sizeof(uint64_t));
    uint32_t *w = weighted ? (uint32_t *)malloc((cap ? cap : 1) * sizeof(uint32_t)) : NULL;
    uint64_t k = 0;
    for (uint64_t e = 0; e < m; e++) {
        uint32_t row = rec[e * stride], col = rec[e * stride + 1];
        uint32_t wt = weighted ? rec[e * stride + 2] : 0;
        if (row == col && !self_loops) continue;
        if (acyclic && col < row) { uint32_t t = row; row = col; col = t; }
        if (transpose) { uint32_t t = row; row = col; col = t; }
        /* ids beyond the grid overflow silently in the reference
           (matrix.hpp:218-220); the oracle rejects them instead */
        if (row >= g->n || col >= g->n) { free(key); free(w); gto_graph_free(g); return NULL; }
        key[k] = ((uint64_t)col << 32) | row; if (w) w[k] = wt; k++;
        if (!directed) { key[k] = ((uint64_t)row << 32) | col; if (w) w[k] = wt; k++; }
    }
    /* column-major order: (col, row) -- ColSort without weights
       (ds/triple.hpp:94); with weights see the header note: (col,row,weight) */
    if (weighted) {
        /* sort by weight first (stable), then by (col,row) => (col,row,weight) */
        uint64_t *wk = (uint64_t *)malloc((k ? k : 1) * sizeof(uint64_t));
        uint32_t *idx = (uint32_t *)malloc((k ? k : 1) * sizeof(uint32_t));
        for (uint64_t i = 0; i < k; i++) { wk[i] = w[i]; idx[i] = (uint32_t)i; }
        radix_sort_kv(wk, idx, k, 32);
        uint64_t *key2 = (uint64_t *)malloc((k ? k : 1) * sizeof(uint64_t));
        uint32_t *w2 = (uint32_t *)malloc((k ? k : 1) * sizeof(uint32_t));
        for (uint64_t i = 0; i < k; i++) { key2[i] = key[idx[i]]; w2[i] = w[idx[i]]; }
        free(key); free(w); free(wk); free(idx);
        key = key2; w = w2;
        radix_sort_kv(key, w, k, 64);
    } else {
        radix_sort_kv(key, NULL, k, 64);
    }
    if (!parallel_edges) {
        uint64_t o = 0;
        for (uint64_t i = 0; i < k; i++) {
            if (o && key[o - 1] == key[i]) continue; /* first copy = min weight */
            key[o] = key[i]; if (w) w[o] = w[i]; o++;
        }
        k = o;
    }
    g->nnz = k;
    /* filters: I[r] = row r has an entry, J[c] = column c has an entry;
       IV / JV = exclusive prefix index (matrix.hpp:861-1122) */
    g->I = (uint8_t *)calloc(g->H, 1); g->J = (uint8_t *)calloc(g->H, 1);
    g->IV = (uint32_t *)calloc(g->H, sizeof(uint32_t));
    g->JV = (uint32_t *)calloc(g->H, sizeof(uint32_t));
    for (uint64_t i = 0; i < k; i++) { g->I[(uint32_t)key[i]] = 1; g->J[key[i] >> 32] = 1; }
    uint32_t nr = 0, nc = 0;
    for (uint32_t v = 0; v < g->H; v++) {
        g->IV[v] = nr; if (g->I[v]) nr++;
        g->JV[v] = nc; if (g->J[v]) nc++;
    }
    g->nnzrows = nr; g->nnzcols = nc;
    g->IR = (uint32_t *)malloc((nr ? nr : 1) * sizeof(uint32_t));
    g->JC = (uint32_t *)malloc((nc ? nc : 1) * sizeof(uint32_t));
    for (uint32_t v = 0, a = 0, b = 0; v < g->H; v++) {
        if (g->I[v]) g->IR[a++] = v;
        if (g->J[v]) g->JC[b++] = v;
    }
    /* TCSC populate (compressed_column.hpp:371-417) */
    g->JA = (uint32_t *)calloc((size_t)nc + 1, sizeof(uint32_t));
    g->IA = (uint32_t *)malloc((k ? k : 1) * sizeof(uint32_t));
    g->A = weighted ? (uint32_t *)malloc((k ? k : 1) * sizeof(uint32_t)) : NULL;
    for (uint64_t i = 0; i < k; i++) {
        uint32_t row = (uint32_t)key[i], col = (uint32_t)(key[i] >> 32);
        g->JA[g->JV[col] + 1]++;
        g->IA[i] = g->IV[row];
        if (weighted) g->A[i] = w[i];
    }
    for (uint32_t j = 0; j < nc; j++) g->JA[j + 1] += g->JA[j];
    free(key); free(w);
    return g;
}

/* field accessors for ctypes */
uint64_t gto_nnz(const gto_graph *g) { return g->nnz; }
uint32_t gto_nnzrows(const gto_graph *g) { return g->nnzrows; }
uint32_t gto_nnzcols(const gto_graph *g) { return g->nnzcols; }
uint32_t gto_height(const gto_graph *g) { return g->H; }
uint32_t gto_nrows(const gto_graph *g) { return g->n; }
const uint32_t *gto_JA(const gto_graph *g) { return g->JA; }
const uint32_t *gto_IA(const gto_graph *g) { return g->IA; }
const uint32_t *gto_A(const gto_graph *g) { return g->A; }
const uint32_t *gto_JC(const gto_graph *g) { return g->JC; }
const uint32_t *gto_IR(const gto_graph *g) { return g->IR; }

/* vertex classes of the TCSC_CF format (matrix.hpp:1125-1144):
   regular = row and column non-empty, source row = row only, sink col = col only */
void gto_class_counts(const gto_graph *g, uint32_t *regular, uint32_t *source_rows, uint32_t *sink_cols) {
    uint32_t r = 0, s = 0, t = 0;
    for (uint32_t v = 0; v < g->H; v++) {
        if (g->I[v] && g->J[v]) r++;
        if (g->I[v] && !g->J[v]) s++;
        if (!g->I[v] && g->J[v]) t++;
    }
    *regular = r; *source_rows = s; *sink_cols = t;
}

/* ------------------------------------------------------------- TCSC_CF */
/*
 * The tile in TCSC_CF_BASE form (ds/compressed_column.hpp:419-470), built
 * from the TCSC arrays the way populate() does it after its TCSC part
 * (:671-1120). A row is a SOURCE row when its vertex has no column
 * (source_rows_bitvector, mat/matrix.hpp:1125-1144); a column is REGULAR
 * when its vertex has a row too, a SINK column otherwise.
 *   IA (and A) : per column the source-row entries are swapped to the tail;
 *                the swap order is the reference's (:671-708), so the
 *                regular entries end up in the reference's order as well.
 *   lists 0..3 : REG_R_REG_C, REG_R_SNK_C, SRC_R_REG_C, SRC_R_SNK_C -- pairs
 *                [begin, end) into IA plus the compressed column of each.
 * Pinned against arrays dumped from the unmodified reference
 * (tests/golden/tcsc_cf.npz, tests/golden/make_tcsc_cf_golden.py).
 */
typedef struct gto_tcsc_cf {
    uint64_t nnz; uint32_t nnzcols;
    uint32_t *IA, *A;
    uint32_t *JA_REG_R_NNZ_C;           /* [2 * nnzcols], :713-742 */
    uint32_t NC[4]; uint32_t *JA[4], *JC[4];
} gto_tcsc_cf;

void gto_tcsc_cf_free(gto_tcsc_cf *c) {
    if (!c) return;
    free(c->IA); free(c->A); free(c->JA_REG_R_NNZ_C);
    for (int k = 0; k < 4; k++) { free(c->JA[k]); free(c->JC[k]); }
    free(c);
}

gto_tcsc_cf *gto_tcsc_cf_build(const gto_graph *g) {
    const uint32_t nc = g->nnzcols;
    const uint32_t *JA = g->JA;
    gto_tcsc_cf *c = (gto_tcsc_cf *)calloc(1, sizeof(gto_tcsc_cf));
    c->nnz = g->nnz; c->nnzcols = nc;
    c->IA = (uint32_t *)malloc((g->nnz ? g->nnz : 1) * sizeof(uint32_t));
    memcpy(c->IA, g->IA, g->nnz * sizeof(uint32_t));
    if (g->A) { c->A = (uint32_t *)malloc((g->nnz ? g->nnz : 1) * sizeof(uint32_t)); memcpy(c->A, g->A, g->nnz * sizeof(uint32_t)); }
    uint32_t *IA = c->IA, *A = c->A;
#define IS_SRC(i) (!g->J[g->IR[IA[i]]])
    uint32_t *nsrc = (uint32_t *)calloc((size_t)nc + 1, sizeof(uint32_t));
    uint32_t *r = (uint32_t *)malloc((g->nnz ? g->nnz : 1) * sizeof(uint32_t));
    /* :671-708 -- for every source entry of a column, front to back: walk from the column's end towards the front;
       the first entry of a regular row found is swapped with it; reaching the entry itself first ends the walk */
    for (uint32_t j = 0; j < nc; j++) {
        uint32_t n = 0, m = JA[j + 1] - JA[j];
        for (uint32_t i = JA[j]; i < JA[j + 1]; i++) if (IS_SRC(i)) r[n++] = i;
        nsrc[j] = n;
        if (n == 0 || m == n) continue;
        for (uint32_t p = 0; p < n; p++) {
            for (uint32_t q = JA[j + 1]; q-- > JA[j];) {
                if (!IS_SRC(q)) {
                    uint32_t t = IA[r[p]]; IA[r[p]] = IA[q]; IA[q] = t;
                    if (A) { t = A[r[p]]; A[r[p]] = A[q]; A[q] = t; }
                    break;
                }
                if (r[p] == q) break;
            }
        }
    }
#undef IS_SRC
    free(r);
    /* :713-742 regular rows of every non-empty column */
    c->JA_REG_R_NNZ_C = (uint32_t *)calloc(2 * (size_t)nc + 1, sizeof(uint32_t));
    for (uint32_t j = 0; j < nc; j++) { c->JA_REG_R_NNZ_C[2 * j] = JA[j]; c->JA_REG_R_NNZ_C[2 * j + 1] = JA[j + 1] - nsrc[j]; }
    /* the four lists walk the columns that have entries in this tile and are regular / sink (:744-1113) */
    for (int k = 0; k < 4; k++) {
        const int want_regular_col = (k == 0 || k == 2);
        /* sizes: lists 0 and 1 count the columns with at least one regular row (:744-775, :846-875); list 2 the columns
           with at least one source row (:952-975); list 3 counts SOURCE ENTRIES, not columns (:1040-1060: k++ per entry),
           so it is longer than what gets filled -- the rest stays zero */
        uint32_t count = 0;
        for (uint32_t j = 0; j < nc; j++) {
            if (JA[j] == JA[j + 1] || (g->I[g->JC[j]] != 0) != want_regular_col) continue;
            uint32_t m = JA[j + 1] - JA[j], n = nsrc[j];
            if (k < 2) count += (m != n);
            else if (k == 2) count += (n > 0);
            else count += n;
        }
        c->NC[k] = count;
        c->JA[k] = (uint32_t *)calloc(2 * (size_t)count + 1, sizeof(uint32_t));
        c->JC[k] = (uint32_t *)calloc((size_t)count + 1, sizeof(uint32_t));
        if (!count) continue;
        uint32_t o = 0;
        for (uint32_t j = 0; j < nc; j++) {
            if (JA[j] == JA[j + 1] || (g->I[g->JC[j]] != 0) != want_regular_col) continue;
            uint32_t m = JA[j + 1] - JA[j], n = nsrc[j];
            if (k < 2) {                 /* :790-843, :888-940 */
                if (m == n) continue;
                c->JA[k][2 * o] = JA[j]; c->JA[k][2 * o + 1] = JA[j + 1] - n;
            } else if (k == 2) {         /* :990-1035 */
                if (!n) continue;
                c->JA[k][2 * o] = JA[j + 1] - n; c->JA[k][2 * o + 1] = JA[j + 1];
            } else {                     /* :1075-1113: the pair starts at JA[j] + n, as the reference writes it */
                if (!n) continue;
                c->JA[k][2 * o] = JA[j] + n; c->JA[k][2 * o + 1] = JA[j + 1];
            }
            c->JC[k][o++] = j;
        }
    }
    free(nsrc);
    return c;
}

const uint32_t *gto_cf_IA(const gto_tcsc_cf *c) { return c->IA; }
const uint32_t *gto_cf_A(const gto_tcsc_cf *c) { return c->A; }
const uint32_t *gto_cf_nnz_pairs(const gto_tcsc_cf *c) { return c->JA_REG_R_NNZ_C; }
uint32_t gto_cf_count(const gto_tcsc_cf *c, int k) { return c->NC[k]; }
const uint32_t *gto_cf_pairs(const gto_tcsc_cf *c, int k) { return c->JA[k]; }
const uint32_t *gto_cf_cols(const gto_tcsc_cf *c, int k) { return c->JC[k]; }

/* spmv_stationary on a TCSC_CF tile in _ROW_ order, vp/vertex_program.hpp:1243-1317: regular rows from sink columns on
   iteration 0 only, regular rows from regular columns while the program runs, source rows on the last iteration. The
   caller passes the three conditions; the SRC_R_SNK_C list is walked over all NC pairs, zero-filled ones included
   (they are empty ranges), and only when NC_SRC_R_REG_C != 0 (:1298), as in the reference. */
void gto_spmv_cf_plus_f64(const gto_tcsc_cf *c, const double *x, double *y, int first, int running, int last) {
    const int use[4] = {running, first, last, last && c->NC[2] != 0};
    const int order[4] = {1, 0, 2, 3};
    for (int o = 0; o < 4; o++) {
        const int k = order[o];
        if (!use[k]) continue;
        for (uint32_t j = 0; j < c->NC[k]; j++) {
            const double xl = x[c->JC[k][j]];
            for (uint32_t i = c->JA[k][2 * j]; i < c->JA[k][2 * j + 1]; i++) y[c->IA[i]] += xl;
        }
    }
}

/* ------------------------------------------------------ kernel-level SpMV */
/* K1, vp/vertex_program.hpp:1162-1173: y[IA[i]] += x[j]   (plus-times, unweighted) */
void gto_spmv_plus_f64(const gto_graph *g, const double *x, double *y) {
    const uint32_t *JA = g->JA, *IA = g->IA;
    for (uint32_t j = 0; j < g->nnzcols; j++) {
        double xj = x[j];
        for (uint32_t i = JA[j]; i < JA[j + 1]; i++) y[IA[i]] += xj;
    }
}
/* K5, vertex_program.hpp:1490-1503: skip x[j]==INF; y = min(y, x[j] (+ w)) */
void gto_spmv_min_u32(const gto_graph *g, const uint32_t *x, uint32_t *y) {
    const uint32_t *JA = g->JA, *IA = g->IA, *A = g->A;
    for (uint32_t j = 0; j < g->nnzcols; j++) {
        uint32_t xj = x[j];
        if (xj == GTO_INF) continue;
        for (uint32_t i = JA[j]; i < JA[j + 1]; i++) {
            uint32_t t = A ? xj + A[i] : xj;
            if (t < y[IA[i]]) y[IA[i]] = t;
        }
    }
}

/* ----------------------------------------------------------- Deg program */
/*
 * apps/deg.h:27-53, one stationary iteration. order_col = 1 reproduces the
 * `_COL_` pass of apps/pr.cpp:40-42 (y[j] += x[IA[i]], vertex_program.hpp:
 * 1174-1184, with I/J roles swapped, :316-324); order_col = 0 the `_ROW_`
 * pass of apps/deg.cpp. degree_out has H entries; vertices whose
 * accumulator slot does not exist keep 0 (applicator(state) no-op, :1666).
 */
void gto_degree(const gto_graph *g, int order_col, uint32_t *degree_out) {
    memset(degree_out, 0, (size_t)g->H * sizeof(uint32_t));
    if (order_col) {
        for (uint32_t j = 0; j < g->nnzcols; j++) {
            double y = 0;
            for (uint32_t i = g->JA[j]; i < g->JA[j + 1]; i++) y += 1.0;
            degree_out[g->JC[j]] = (uint32_t)y;
        }
    } else {
        double *y = (double *)calloc(g->nnzrows ? g->nnzrows : 1, sizeof(double));
        for (uint32_t j = 0; j < g->nnzcols; j++)
            for (uint32_t i = g->JA[j]; i < g->JA[j + 1]; i++) y[g->IA[i]] += 1.0;
        for (uint32_t r = 0; r < g->nnzrows; r++) degree_out[g->IR[r]] = (uint32_t)y[r];
        free(y);
    }
}

/* -------------------------------------------------------------- PageRank */
/*
 * apps/pr.h:21-48 driven by vertex_program.hpp:408-441.
 *   degree_in : H entries from gto_degree (the `Deg` program's V)
 *   iters     : fixed count, or 0 = run until converged (:412-413)
 *   cf        : 1 = TCSC_CF semantics of apps/pr.cpp, 0 = TCSC of apps/pr1.cpp.
 *               With a fixed count both give identical values; in converge
 *               mode TCSC_CF (i) tests regular rows only (:1902-1916) and
 *               (ii) ends by applying source rows from a y that never
 *               received their edges, leaving them at exactly alpha
 *               (:1036-1041, 1282, 1683-1691; SURVEY 8a trap 5).
 * initialize(other) copies the degree only where the row is non-empty
 * (:476-483) -- pure sources message 0 and keep rank alpha (trap 1).
 * Returns iterations executed.
 */
uint32_t gto_pagerank(const gto_graph *g, const uint32_t *degree_in, uint32_t iters, int cf,
                      double alpha, double tol, double *rank_out, uint32_t *degree_out) {
    const uint32_t H = g->H, nr = g->nnzrows, nc = g->nnzcols;
    uint8_t *C = (uint8_t *)malloc(H);
    for (uint32_t v = 0; v < H; v++) {
        rank_out[v] = alpha;                       /* PR_State default, pr.h:16 */
        degree_out[v] = g->I[v] ? degree_in[v] : 0;
        C[v] = 1;                                  /* initializer returns true  */
    }
    double *x = (double *)malloc((nc ? nc : 1) * sizeof(double));
    double *y = (double *)malloc((nr ? nr : 1) * sizeof(double));
    uint32_t it = 0;
    for (;;) {
        /* scatter_gather_stationary :688-708 */
        for (uint32_t j = 0; j < nc; j++) {
            uint32_t v = g->JC[j];
            x[j] = degree_out[v] ? rank_out[v] / degree_out[v] : 0;
        }
        /* combine :1017-1035 (zero y, SpMV) */
        memset(y, 0, (size_t)nr * sizeof(double));
        gto_spmv_plus_f64(g, x, y);
        /* apply_stationary :1641-1693. Fixed count: TCSC_CF touches regular
           rows every iteration and source rows on the last one; values equal
           TCSC's because a source row is never messaged. */
        int last = (iters != 0) && (it + 1 == iters);
        for (uint32_t r = 0; r < nr; r++) {
            uint32_t v = g->IR[r];
            if (cf && !g->J[v] && !last) continue;     /* source row, not last */
            double tmp = rank_out[v];
            rank_out[v] = alpha + (1.0 - alpha) * y[r];
            C[v] = fabs(rank_out[v] - tmp) > tol;
        }
        if (!cf) for (uint32_t v = 0; v < H; v++) if (!g->I[v]) C[v] = 0; /* :1667 */
        it++;
        if (iters == 0) {
            /* has_converged :1885-1923 */
            int conv = 1;
            if (cf) { for (uint32_t v = 0; v < H; v++) if (g->I[v] && g->J[v] && C[v]) { conv = 0; break; } }
            else    { for (uint32_t v = 0; v < H; v++) if (C[v]) { conv = 0; break; } }
            if (conv) {
                /* extra combine();apply() :425-428. TCSC: SpMV re-runs but
                   apply is skipped -> no visible effect. TCSC_CF: combine is
                   a no-op, apply_stationary updates source rows from y whose
                   source-row slots are still 0 from the last zero-fill. */
                if (cf) for (uint32_t r = 0; r < nr; r++) {
                    uint32_t v = g->IR[r];
                    if (!g->J[v]) rank_out[v] = alpha + (1.0 - alpha) * 0.0;
                }
                break;
            }
        } else if (it >= iters) break;
    }
    free(C); free(x); free(y);
    return it;
}

/* ------------------------------------------- non-stationary engine (min) */
/*
 * Shared skeleton of BFS / SSSP / CC: vertex_program.hpp:711-758
 * (C-gated messenger), :1490-1503 (dense SpMV with INF skip; the sparse
 * variant :1475-1489 gives identical results, trap 4), :1696-1802 (apply;
 * iteration 0 walks all of V, later ones the non-empty rows; y is NOT reset
 * for these three programs, trap 3), :1885-1901 (all C == 0).
 * kind: 2 = BFS, 3 = SSSP, 4 = CC. s0/s1 are the state arrays (H entries):
 *   BFS : s0 = parent, s1 = hops     SSSP: s0 = distance     CC: s0 = label
 */
static uint32_t run_min_program(const gto_graph *g, int kind, uint32_t root,
                                uint32_t *s0, uint32_t *s1, uint32_t max_iters) {
    const uint32_t H = g->H, nr = g->nnzrows, nc = g->nnzcols;
    uint8_t *C = (uint8_t *)malloc(H);
    /* initializer: bfs.h:37-50, sssp.h:33-42, cc.h:33-36 */
    for (uint32_t v = 0; v < H; v++) {
        if (kind == 2) { s0[v] = (v == root) ? v : 0; s1[v] = (v == root) ? 0 : GTO_INF; C[v] = (v == root); }
        else if (kind == 3) { s0[v] = (v == root) ? 0 : GTO_INF; C[v] = (v == root); }
        else { s0[v] = v; C[v] = 1; }
    }
    uint32_t *x = (uint32_t *)malloc((nc ? nc : 1) * sizeof(uint32_t));
    uint32_t *y = (uint32_t *)malloc((nr ? nr : 1) * sizeof(uint32_t));
    for (uint32_t r = 0; r < nr; r++) y[r] = GTO_INF;          /* :625-635 */
    uint32_t it = 0;
    for (;;) {
        for (uint32_t j = 0; j < nc; j++) {
            uint32_t v = g->JC[j];
            /* messenger: bfs.h:52-54 (vid), sssp.h:44-46 (distance), cc.h:38-40 (label) */
            x[j] = C[v] ? (kind == 2 ? v : s0[v]) : GTO_INF;
        }
        gto_spmv_min_u32(g, x, y);
        if (it == 0) for (uint32_t v = 0; v < H; v++) if (!g->I[v]) C[v] = 0;  /* applicator(state) */
        for (uint32_t r = 0; r < nr; r++) {
            uint32_t v = g->IR[r], yv = y[r];
            if (kind == 2) {                     /* bfs.h:65-77 */
                if (s1[v] != GTO_INF) C[v] = 0;
                else if (yv != GTO_INF) { s1[v] = it + 1; s0[v] = yv; C[v] = 1; }
                else C[v] = 0;
            } else {                             /* sssp.h:57-65 (HAS_WEIGHT), cc.h:51-55 */
                uint32_t tmp = s0[v];
                s0[v] = (yv < s0[v]) ? yv : s0[v];
                C[v] = (tmp != s0[v]);
            }
        }
        it++;
        int conv = 1;
        for (uint32_t v = 0; v < H; v++) if (C[v]) { conv = 0; break; }
        if (conv) break;
        if (max_iters && it >= max_iters) break;
    }
    free(C); free(x); free(y);
    return it;
}

uint32_t gto_bfs(const gto_graph *g, uint32_t root, uint32_t *parent, uint32_t *hops) {
    return run_min_program(g, 2, root, parent, hops, 0);
}
uint32_t gto_sssp(const gto_graph *g, uint32_t root, uint32_t *distance) {
    return run_min_program(g, 3, root, distance, NULL, 0);
}
uint32_t gto_cc(const gto_graph *g, uint32_t *label) {
    return run_min_program(g, 4, 0, label, NULL, 0);
}

/* -------------------------------------------------------------- checksum */
/*
 * vertex_program.hpp:1927-1960. The accumulator is uint64_t and is `+=`'d
 * with the state value, so for a double state every add truncates (trap 8).
 * `infinity` is the program's infinity() (0 for Deg/PR, INF otherwise).
 */
void gto_checksum_u32(const uint32_t *state, uint32_t count, uint32_t nrows, uint32_t infinity,
                      uint64_t *value_sum, uint64_t *reachable) {
    uint64_t s = 0, c = 0;
    for (uint32_t i = 0; i < count; i++)
        if (state[i] != infinity && i < nrows) { s += state[i]; c++; }
    *value_sum = s; *reachable = c;
}
void gto_checksum_f64(const double *state, uint32_t count, uint32_t nrows,
                      uint64_t *value_sum, uint64_t *reachable) {
    uint64_t s = 0, c = 0;
    for (uint32_t i = 0; i < count; i++)
        if (state[i] != 0.0 && i < nrows) { s = (uint64_t)((double)s + state[i]); c++; }
    *value_sum = s; *reachable = c;
}
