// bfs.cpp -- Breadth First Search main, counterpart of /root/reference/src/apps/bfs.cpp.
#include "common.hpp"

int main(int argc, char **argv) try {
    EndToEnd e2e("Breadth First Search (BFS)");
    if (argc != 3 && argc != 4) return usage(argv[0], "<file_path> <num_vertices> <root>");
    std::string file_path = argv[1];
    uint32_t num_vertices = std::atoi(argv[2]);
    uint32_t root = (argc > 3) ? std::atoi(argv[3]) : 0;
    bool directed = false, transpose = false, self_loops = false, acyclic = false, parallel_edges = false;
    bool stationary = false;
    if (!stationary && directed) transpose = !transpose;  // engine requirement for non-stationary programs on directed graphs
    gt::Graph G;
    G.load(file_path, num_vertices, num_vertices, directed, transpose, self_loops, acyclic, parallel_edges, gt::_2DT_, gt::_TCSC_);
    bool gather_depends_on_apply = false, apply_depends_on_iter = true;
    gt::BFS_Program V(G, stationary, gather_depends_on_apply, apply_depends_on_iter, gt::_ROW_);
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
