// cc.cpp -- Connected Components main, counterpart of /root/reference/src/apps/cc.cpp.
#include "common.hpp"

int main(int argc, char **argv) try {
    EndToEnd e2e("Connected component");
    if (argc != 3 && argc != 4) return usage(argv[0], "<file_path> <num_vertices>");
    std::string file_path = argv[1];
    uint32_t num_vertices = std::atoi(argv[2]);
    bool directed = false, transpose = false, self_loops = true, acyclic = false, parallel_edges = false;
    bool stationary = false;
    if (!stationary && directed) transpose = !transpose;
    gt::Graph G;
    G.load(file_path, num_vertices, num_vertices, directed, transpose, self_loops, acyclic, parallel_edges, gt::_2DT_, gt::_TCSC_);
    bool gather_depends_on_apply = true, apply_depends_on_iter = false;
    gt::CC_Program V(G, stationary, gather_depends_on_apply, apply_depends_on_iter, gt::_ROW_);
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
