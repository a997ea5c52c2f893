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
    explicit EndToEnd(const char *n) : name(n) {}
    ~EndToEnd() { printf("%s end-to-end time: %f seconds\n", name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
};

static inline int usage(const char *argv0, const char *args) {
    printf("\"Usage: %s %s\"\n", argv0, args);
    return 1;
}
