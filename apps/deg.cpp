// deg.cpp -- Degree main, counterpart of /root/reference/src/apps/deg.cpp.
#include "common.hpp"

int main(int argc, char **argv) try {
    EndToEnd e2e("Degree");
    if (argc != 3 && argc != 4) return usage(argv[0], "<file_path> <num_vertices>");
    std::string file_path = argv[1];
    uint32_t num_vertices = std::atoi(argv[2]);
    gt::Graph G;
    G.load(file_path, num_vertices, num_vertices, true, false, true, false, true, gt::_2DT_, gt::_TCSC_);
    gt::Deg_Program V(G, true, false, false, gt::_ROW_);
    V.execute(1);
    V.checksum();
    V.checksum1(GT_F_DEGREE);   // deg.cpp:43
    V.display();
    V.free();
    G.free();
    return 0;
} catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
}
