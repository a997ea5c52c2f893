// common.hpp -- shared by the application mains: same CLI, flags and printed lines as the
// reference's src/apps/*.cpp, on top of include/graphtap_amd.hpp.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../include/graphtap_amd.hpp"

struct EndToEnd {
    const char *name;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    // Env::init() first, like the reference's mains (src/apps/pr.cpp:16): with GRAPHTAP_NGPUS=N this call forks the N rank
    // processes before anything touches the GPU
    explicit EndToEnd(const char *n) : name(n) { gt::Env::init(); t0 = std::chrono::steady_clock::now(); }
    ~EndToEnd() {
        GT_MASTER_PRINTF("%s end-to-end time: %f seconds\n", name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        try { gt::Env::finalize(); } catch (...) {}
    }
};

static inline int usage(const char *argv0, const char *args) {
    GT_MASTER_PRINTF("\"Usage: %s %s\"\n", argv0, args);
    return 1;
}
