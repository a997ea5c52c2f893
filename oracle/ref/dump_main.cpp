/*
 * dump_main.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Out-of-tree driver for the *unmodified* GraphTap reference headers under
 * /root/reference/src. It runs one of the reference's five vertex programs
 * exactly the way the reference's own mains configure them
 * (src/apps/{deg,pr,pr1,bfs,sssp,cc}.cpp) and writes every rank's full
 * vertex-state vector `V` (public member, src/vp/vertex_program.hpp:61) to
 * `<out>.<rank>.bin`, so that golden fixtures under tests/golden/ come from
 * the reference itself and not from a re-implementation.
 *
 * Select the program at compile time: -DAPP_DEG | -DAPP_PR | -DAPP_PR1 |
 * -DAPP_BFS | -DAPP_SSSP (+ -DHAS_WEIGHT) | -DAPP_CC | -DAPP_TCSC_CF
 * (+ -DHAS_WEIGHT).
 *
 * -DAPP_TCSC_CF (np = 1 only) runs no program: it loads the graph with the
 * flags of src/apps/pr.cpp and writes the tile's TCSC_CF_BASE arrays
 * (src/ds/compressed_column.hpp:419-470, all public members of the
 * compressor) to `<out>.cf.bin`:
 *   u32 magic 'GTCF', u32 weighted, u64 nnz, u32 nnzcols, u32 nnzrows,
 *   IA[nnz], (A[nnz],) JA[nnzcols+1], JC[nnzcols], IR[nnzrows],
 *   JA_REG_R_NNZ_C[2*nnzcols], then for REG_R_REG_C, REG_R_SNK_C,
 *   SRC_R_REG_C, SRC_R_SNK_C: u32 NC, JA_*[2*NC], JC_*[NC].
 * Graph::A and Matrix::tiles are private; they are read through the
 * explicit-instantiation rule ([temp.explicit]: access checks do not apply
 * to names in an explicit instantiation) -- the reference is not modified.
 *
 * File layout (little endian):
 *   u32 magic 'GTV1', u32 app, u32 rank, u32 nranks, u32 owned_segment,
 *   u32 tile_height, u32 nitems, u32 iterations,
 *   then nitems records: u32 a, u32 b, f64 c  (meaning depends on app):
 *     DEG : a=degree
 *     PR  : a=degree, c=rank
 *     BFS : a=parent, b=hops
 *     SSSP: a=distance
 *     CC  : a=label
 */
#include <iostream>
#include <cstdio>
#include <cstdint>
#include <unistd.h>

#include "mpi/env.hpp"
#include "mat/graph.hpp"

#if defined(APP_DEG)
#include "deg.h"
#define APP_ID 0
#elif defined(APP_PR) || defined(APP_PR1)
#include "pr.h"
#define APP_ID 1
#elif defined(APP_BFS)
#include "bfs.h"
#define APP_ID 2
#elif defined(APP_SSSP)
#include "sssp.h"
#define APP_ID 3
#elif defined(APP_CC)
#include "cc.h"
#define APP_ID 4
#elif defined(APP_TCSC_CF)
#ifdef HAS_WEIGHT
#include "sssp.h"
#else
#include "pr.h"
#endif
#define APP_ID 5
template <class Tag, typename Tag::type M>
struct Expose { friend typename Tag::type peek(Tag) { return M; } };
struct GraphA { typedef Matrix<wp, ip, fp>* Graph<wp, ip, fp>::*type; friend type peek(GraphA); };
template struct Expose<GraphA, &Graph<wp, ip, fp>::A>;
struct MatrixTiles { typedef std::vector<std::vector<struct Tile2D<wp, ip, fp>>> Matrix<wp, ip, fp>::*type; friend type peek(MatrixTiles); };
template struct Expose<MatrixTiles, &Matrix<wp, ip, fp>::tiles>;
#else
#error "pick an APP_*"
#endif

template <class P>
struct Peek : public P {
    using P::P;
    uint32_t seg() { return (uint32_t)this->owned_segment; }
    uint32_t height() { return (uint32_t)this->tile_height; }
};

struct Rec { uint32_t a, b; double c; };

template <class P, class F>
static void dump(const char* out, P& prog, F fill) {
    char path[4096];
    snprintf(path, sizeof(path), "%s.%d.bin", out, Env::rank);
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); Env::exit(1); }
    uint32_t hdr[8] = {0x31565447u, APP_ID, (uint32_t)Env::rank, (uint32_t)Env::nranks,
                       prog.seg(), prog.height(), (uint32_t)prog.V.size(), (uint32_t)prog.iteration};
    fwrite(hdr, sizeof(hdr), 1, f);
    for (size_t i = 0; i < prog.V.size(); i++) {
        Rec r = {0, 0, 0.0};
        fill(prog.V[i], r);
        fwrite(&r, sizeof(r), 1, f);
    }
    fclose(f);
}

int main(int argc, char** argv) {
    Env::init();
    if (argc < 4) {
        if (Env::is_master) fprintf(stderr, "usage: %s <edges.bin> <num_vertices> <out_prefix> [iters|root]\n", argv[0]);
        Env::exit(1);
    }
    std::string file_path = argv[1];
    ip num_vertices = std::atoi(argv[2]);
    const char* out = argv[3];
    ip arg = (argc > 4) ? (ip)std::atoi(argv[4]) : 0;
    Tiling_type TT = _2DT_;

#if defined(APP_DEG)
    Graph<wp, ip, ip> G;
    G.load(file_path, num_vertices, num_vertices, true, false, true, false, true, TT, _TCSC_);
    Peek<Deg_Program<wp, ip, ip>> V(G, true, false, false, _ROW_);
    V.execute(1);
    V.checksum();
    dump(out, V, [](Deg_State& s, Rec& r) { r.a = s.degree; });
    V.free();
    G.free();
#elif defined(APP_PR)
    Graph<wp, ip, fp> G;
    G.load(file_path, num_vertices, num_vertices, true, true, true, false, true, TT, _TCSC_CF_);
    Deg_Program<wp, ip, fp> V(G, true, false, false, _COL_);
    V.execute(1);
    V.checksum();
    Peek<PR_Program<wp, ip, fp>> VR(G, true, false, false, _ROW_);
    VR.initialize(V);
    V.free();
    VR.execute(arg);
    VR.checksum();
    dump(out, VR, [](PR_State& s, Rec& r) { r.a = s.degree; r.c = s.rank; });
    VR.free();
    G.free();
#elif defined(APP_PR1)
    Graph<wp, ip, fp> G;
    G.load(file_path, num_vertices, num_vertices, true, false, true, false, true, TT, _TCSC_);
    Deg_Program<wp, ip, fp> V(G, true, false, false, _ROW_);
    V.execute(1);
    V.checksum();
    G.free();
    Graph<wp, ip, fp> GR;
    GR.load(file_path, num_vertices, num_vertices, true, true, true, false, true, TT, _TCSC_);
    Peek<PR_Program<wp, ip, fp>> VR(GR, true, false, false, _ROW_);
    VR.initialize(V);
    V.free();
    VR.execute(arg);
    VR.checksum();
    dump(out, VR, [](PR_State& s, Rec& r) { r.a = s.degree; r.c = s.rank; });
    VR.free();
    GR.free();
#elif defined(APP_BFS)
    Graph<wp, ip, fp> G;
    G.load(file_path, num_vertices, num_vertices, false, false, false, false, false, TT, _TCSC_);
    Peek<BFS_Program<wp, ip, fp>> V(G, false, false, true, _ROW_);
    V.root = arg;
    V.execute();
    V.checksum();
    dump(out, V, [](BFS_State& s, Rec& r) { r.a = s.parent; r.b = s.hops; });
    V.free();
    G.free();
#elif defined(APP_SSSP)
    Graph<wp, ip, fp> G;
    G.load(file_path, num_vertices, num_vertices, true, true, false, false, false, TT, _TCSC_);
    Peek<SSSP_Program<wp, ip, fp>> V(G, false, true, false, _ROW_);
    V.root = arg;
    V.execute();
    V.checksum();
    dump(out, V, [](SSSP_State& s, Rec& r) { r.a = s.distance; });
    V.free();
    G.free();
#elif defined(APP_TCSC_CF)
    if (Env::nranks != 1) { fprintf(stderr, "dump_tcsc_cf: np = 1 only\n"); Env::exit(1); }
    Graph<wp, ip, fp> G;
    G.load(file_path, num_vertices, num_vertices, true, true, true, false, true, TT, _TCSC_CF_);
    {
        Matrix<wp, ip, fp>* M = G.*peek(GraphA());
        auto& tile = (M->*peek(MatrixTiles()))[0][0];
        auto* c = static_cast<TCSC_CF_BASE<wp, ip>*>(tile.compressor);
        char path[4096];
        snprintf(path, sizeof(path), "%s.cf.bin", out);
        FILE* f = fopen(path, "wb");
        if (!f) { perror(path); Env::exit(1); }
#ifdef HAS_WEIGHT
        uint32_t weighted = 1;
#else
        uint32_t weighted = 0;
#endif
        uint32_t magic = 0x46435447u, nc = c->nnzcols, nr = c->nnzrows;
        uint64_t nnz = c->nnz;
        fwrite(&magic, 4, 1, f); fwrite(&weighted, 4, 1, f); fwrite(&nnz, 8, 1, f); fwrite(&nc, 4, 1, f); fwrite(&nr, 4, 1, f);
        fwrite(c->IA, 4, nnz, f);
#ifdef HAS_WEIGHT
        fwrite(c->A, 4, nnz, f);
#endif
        fwrite(c->JA, 4, nc + 1, f); fwrite(c->JC, 4, nc, f); fwrite(c->IR, 4, nr, f);
        fwrite(c->JA_REG_R_NNZ_C, 4, 2 * (size_t)nc, f);
        auto list = [&](uint32_t n, ip* ja, ip* jc) { fwrite(&n, 4, 1, f); if (n) { fwrite(ja, 4, 2 * (size_t)n, f); fwrite(jc, 4, n, f); } };
        list(c->NC_REG_R_REG_C, c->JA_REG_R_REG_C, c->JC_REG_R_REG_C);
        list(c->NC_REG_R_SNK_C, c->JA_REG_R_SNK_C, c->JC_REG_R_SNK_C);
        list(c->NC_SRC_R_REG_C, c->JA_SRC_R_REG_C, c->JC_SRC_R_REG_C);
        list(c->NC_SRC_R_SNK_C, c->JA_SRC_R_SNK_C, c->JC_SRC_R_SNK_C);
        fclose(f);
    }
    G.free();
#elif defined(APP_CC)
    Graph<wp, ip, fp> G;
    G.load(file_path, num_vertices, num_vertices, false, false, true, false, false, TT, _TCSC_);
    Peek<CC_Program<wp, ip, fp>> V(G, false, true, false, _ROW_);
    V.execute();
    V.checksum();
    dump(out, V, [](CC_State& s, Rec& r) { r.a = s.label; });
    V.free();
    G.free();
#endif
    Env::finalize();
    return 0;
}
