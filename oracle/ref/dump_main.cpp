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
 * -DAPP_BFS | -DAPP_SSSP (+ -DHAS_WEIGHT) | -DAPP_CC.
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
