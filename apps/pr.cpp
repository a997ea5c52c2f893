// pr.cpp -- PageRank main, counterpart of /root/reference/src/apps/pr.cpp (same flags, same order of calls).
#include "common.hpp"

int main(int argc, char **argv) try {
    EndToEnd e2e("PageRank");
    if (argc != 3 && argc != 4) return usage(argv[0], "<file_path> <num_vertices> [<num_iterations=INF>]");
    std::string file_path = argv[1];
    uint32_t num_vertices = std::atoi(argv[2]);
    uint32_t num_iterations = (argc > 3) ? (uint32_t)atoi(argv[3]) : 0;
    bool directed = true, transpose = true, self_loops = true, acyclic = false, parallel_edges = true;
    gt::Graph G;
    G.load(file_path, num_vertices, num_vertices, directed, transpose, self_loops, acyclic, parallel_edges, gt::_2DT_, gt::_TCSC_CF_);
    bool stationary = true, gather_depends_on_apply = false, apply_depends_on_iter = false;
    gt::Deg_Program V(G, stationary, gather_depends_on_apply, apply_depends_on_iter, gt::_COL_);
    V.execute(1);
    V.checksum();
    gt::PR_Program VR(G, stationary, gather_depends_on_apply, apply_depends_on_iter, gt::_ROW_);
    VR.initialize(V);
    V.free();
    VR.execute(num_iterations);
    VR.checksum();
    VR.display();
    VR.free();
    G.free();
    return 0;
} catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
}
