// sssp.cpp -- Single Source Shortest Path main, counterpart of /root/reference/src/apps/sssp.cpp
// (the reference builds it with -DHAS_WEIGHT: 12-byte records).
#include "common.hpp"

int main(int argc, char **argv) try {
    EndToEnd e2e("Single Source Shortest Path (SSSP)");
    if (argc != 3 && argc != 4 && argc != 5) return usage(argv[0], "<file_path> <num_vertices> <root>");
    std::string file_path = argv[1];
    uint32_t num_vertices = std::atoi(argv[2]);
    uint32_t root = (argc > 3) ? std::atoi(argv[3]) : 0;
    bool directed = true, transpose = false, self_loops = false, acyclic = false, parallel_edges = false;
    bool stationary = false;
    if (!stationary && directed) transpose = !transpose;
    gt::Graph G(/*weighted=*/true);
    G.load(file_path, num_vertices, num_vertices, directed, transpose, self_loops, acyclic, parallel_edges, gt::_2DT_, gt::_TCSC_);
    bool gather_depends_on_apply = true, apply_depends_on_iter = false;
    gt::SSSP_Program V(G, stationary, gather_depends_on_apply, apply_depends_on_iter, gt::_ROW_);
    V.root = root;
    V.execute();
    V.checksum();
    V.display();
    V.free();
    G.free();
    return 0;
} catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
}
