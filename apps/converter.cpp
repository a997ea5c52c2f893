// converter.cpp -- edge-list converter (text <-> binary, with or without weights): the companion of the reference's
// src/misc/converter.cpp, same command line, same output files byte for byte and the same statistics lines.
//
//   converter <in> <in is 0(txt)|1(bin)> <in weighted 0|1> <out> <out 0(txt)|1(bin)> <out weighted 0|1> [<offset>]
//
// Records are <u4 row, u4 col[, u4 weight]> little-endian (ds/triple.hpp:10-13); text lines are "row col[ weight]",
// lines starting with '#' or '%' are comments (echoed, converter.cpp:117-125). An input without weights gets
// 1 + rand() % 128 per edge from the C library's default-seeded rand() (converter.cpp:81, 136) -- drawn for every edge
// whether or not the output keeps it, so a weighted output is reproducible. `offset` is added to both vertex ids.
// Deviation: empty text lines are skipped (the reference re-emits the previous edge's column for them).
// Host-only tool: it does not touch the GPU library.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Edge { uint32_t row, col, weight; };

struct Sink {   // buffered writer of either format
    FILE *f; bool binary, weighted;
    std::vector<char> buf;
    void put(const Edge &e) {
        if (binary) {
            const size_t n = weighted ? 12 : 8, at = buf.size();
            buf.resize(at + n);
            memcpy(&buf[at], &e, n);
        } else {
            char line[48];
            const int n = weighted ? snprintf(line, sizeof line, "%u %u %u\n", e.row, e.col, e.weight)
                                   : snprintf(line, sizeof line, "%u %u\n", e.row, e.col);
            buf.insert(buf.end(), line, line + n);
        }
        if (buf.size() >= (1u << 20)) flush();
    }
    void flush() { if (!buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
};

struct Stats { uint32_t comments = 0, max_vertex = 0; uint64_t edges = 0; };

void emit(Edge e, bool had_weight, uint32_t offset, Sink &out, Stats &st) {
    if (!had_weight) e.weight = 1u + (uint32_t)(std::rand() % (127 + 1));
    e.row += offset; e.col += offset;
    out.put(e);
    st.edges++;
    if (e.row > st.max_vertex) st.max_vertex = e.row;
    if (e.col > st.max_vertex) st.max_vertex = e.col;
}

int from_binary(FILE *in, bool weighted, uint32_t offset, Sink &out, Stats &st) {
    const size_t rec = weighted ? 3 : 2;
    std::vector<uint32_t> chunk(rec << 16);
    size_t got;
    while ((got = fread(chunk.data(), 4, chunk.size(), in)) > 0) {
        if (got % rec) { fprintf(stderr, "read() failure: the input is not a whole number of %zu-byte records\n", rec * 4); return 1; }
        for (size_t i = 0; i < got; i += rec) emit(Edge{chunk[i], chunk[i + 1], weighted ? chunk[i + 2] : 0u}, weighted, offset, out, st);
    }
    return 0;
}

int from_text(FILE *in, bool weighted, uint32_t offset, Sink &out, Stats &st) {
    char *line = nullptr; size_t cap = 0; ssize_t n;
    bool banner = false;
    while ((n = getline(&line, &cap, in)) >= 0) {
        while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        if (n == 0) continue;
        if (line[0] == '#' || line[0] == '%') {
            if (!banner) { banner = true; puts("########################################"); }
            puts(line);
            st.comments++;
            continue;
        }
        Edge e{0, 0, 0};
        const int want = weighted ? 3 : 2;
        const int have = weighted ? sscanf(line, "%u %u %u", &e.row, &e.col, &e.weight) : sscanf(line, "%u %u", &e.row, &e.col);
        if (have != want) { fprintf(stderr, "read() failure \"%s\"\n", line); free(line); return 1; }
        emit(e, weighted, offset, out, st);
    }
    free(line);
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc != 7 && argc != 8) {
        printf("Usage: %s <filepath_in> <filepath_in is [0(txt)|1(bin)]> <filepath_in is weighted [0|1]>  <filepath_out> "
               "<Want filepath_out be [0(txt)|1(bin)]> <Want filepath_out be weighted [0|1]> [<offsetted by ?>]\n", argv[0]);
        return 1;
    }
    const bool in_bin = atoi(argv[2]) == 1, in_w = atoi(argv[3]) == 1, out_bin = atoi(argv[5]) == 1, out_w = atoi(argv[6]) == 1;
    const uint32_t offset = argc == 8 ? (uint32_t)atoi(argv[7]) : 0u;
    FILE *in = fopen(argv[1], in_bin ? "rb" : "r");
    if (!in) { fprintf(stderr, "Unable to open input file\n"); return 1; }
    FILE *of = fopen(argv[4], out_bin ? "wb" : "w");
    if (!of) { fprintf(stderr, "Unable to open output file\n"); fclose(in); return 1; }
    Sink out{of, out_bin, out_w, {}};
    Stats st;
    const int rc = in_bin ? from_binary(in, in_w, offset, out, st) : from_text(in, in_w, offset, out, st);
    out.flush();
    fclose(of); fclose(in);
    if (rc) return rc;
    puts("########################################");
    puts("Read/write stats:");
    printf("%u line comments\n", st.comments);
    printf("%u vertices (excluding zero)\n", st.max_vertex);
    printf("%llu edges \n", (unsigned long long)st.edges);
    printf("%llu number of lines\n", (unsigned long long)(st.comments + st.edges));
    printf("Verify using \"wc -l %s\"\n", argv[1]);
    return 0;
}
