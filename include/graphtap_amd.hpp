// graphtap_amd.hpp -- header-only C++ shim over the C ABI (graphtap_amd.h) with the shape of the
// reference's Graph<> / Vertex_Program<> so that application mains read like src/apps/*.cpp:
//
//     gt::Graph G;
//     G.load(file_path, num_vertices, num_vertices, directed, transpose, self_loops, acyclic,
//            parallel_edges, gt::_2DT_, gt::_TCSC_CF_);
//     gt::Deg_Program V(G, stationary, gather_depends_on_apply, apply_depends_on_iter, gt::_COL_);
//     V.execute(1);
//     gt::PR_Program VR(G, stationary, gather_depends_on_apply, apply_depends_on_iter, gt::_ROW_);
//     VR.initialize(V); V.free(); VR.execute(num_iterations); VR.checksum(); VR.display(); VR.free(); G.free();
//
// Differences from the reference that a caller sees:
//  * the five programs are concrete classes; the virtual initializer/messenger/combiner/applicator
//    hooks (src/vp/vertex_program.hpp:32-45) are device kernels selected by the class;
//  * errors throw gt::Error instead of exit(1) (src/mat/graph.hpp:143-144);
//  * weights are a run-time property of the Graph object (the reference uses -DHAS_WEIGHT);
//  * V is struct-of-arrays copied out of HBM on demand (`V()`), not a std::vector of structs.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "graphtap_amd.h"

namespace gt {

enum Tiling_type { _2D_, _2DT_ };                          // src/mat/tiling.hpp:13-16
enum Compression_type { _CSC_, _DCSC_, _TCSC_, _TCSC_CF_ };
enum Ordering_type { _ROW_ = GT_ROW, _COL_ = GT_COL };     // src/vp/vertex_program.hpp:17-21

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void check(int status) {
    if (status != GT_OK) throw Error(std::string("graphtap_amd: ") + gt_last_error());
}

// Env, src/mpi/env.hpp:22-55, over RCCL instead of MPI: one PROCESS per GPU, rank r drives GPU r and owns tile-row r.
//   GRAPHTAP_NGPUS=N   Env::init() itself is the launcher: BEFORE anything touches HIP it forks N rank processes (the caller
//                      becomes a pure waiter and exits with the ranks' worst status), rank 0 draws the RCCL unique id and
//                      hands it over through a shared page;
//   RANK / WORLD_SIZE [/ LOCAL_RANK] + GRAPHTAP_UID_FILE   an outside launcher (mpirun, torchrun, slurm) started the ranks:
//                      rank 0 writes the unique id to that (shared) path, the others wait for it;
//   neither            one rank, the reference's `mpirun -np 1`.
// Only the master prints (Env::is_master, env.hpp:57), like the reference.
struct Env {
    static int &rank() { static int v = 0; return v; }
    static int &nranks() { static int v = 1; return v; }
    static gt_dist *&dist() { static gt_dist *d = nullptr; return d; }
    static bool is_master() { return rank() == 0; }
    static bool exchange() { const char *e = getenv("GRAPHTAP_FORCE_EXCHANGE"); return nranks() > 1 || (e && atoi(e) != 0); }

    static void init() {
        const char *ng = getenv("GRAPHTAP_NGPUS"), *ws = getenv("WORLD_SIZE"), *rk = getenv("RANK");
        unsigned char id[GT_DIST_UNIQUE_ID_BYTES];
        int local = 0;
        if (ng && atoi(ng) >= 1) {
            const int n = atoi(ng);
            struct Page { volatile int ready; unsigned char id[GT_DIST_UNIQUE_ID_BYTES]; };
            Page *pg = (Page *)mmap(nullptr, sizeof(Page), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
            if (pg == MAP_FAILED) throw Error("Env::init: mmap failed");
            pg->ready = 0;
            std::vector<pid_t> kids;
            int me = -1;
            fflush(stdout); fflush(stderr);
            for (int r = 0; r < n; r++) {
                pid_t c = fork();
                if (c < 0) throw Error("Env::init: fork failed");
                if (c == 0) { me = r; break; }
                kids.push_back(c);
            }
            if (me < 0) {   // the launcher itself: never initialises the GPU
                // the first rank that ends badly ends the run: its peers would wait for it in a collective for ever
                int worst = 0;
                for (size_t left = kids.size(); left > 0; left--) {
                    int st = 0;
                    const pid_t c = waitpid(-1, &st, 0);
                    if (c < 0) break;
                    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
                    if (code > worst) worst = code;
                    if (code != 0) {
                        fprintf(stderr, "graphtap launcher: a rank process ended with status %d: stopping the others\n", code);
                        for (pid_t k : kids) if (k != c) kill(k, SIGTERM);
                    }
                }
                _exit(worst);
            }
            rank() = me; nranks() = n; local = me;
            if (getenv("GRAPHTAP_SHARE_GPU")) local = 0;
            check(gt_set_device(local));
            if (me == 0) { check(gt_dist_unique_id((void *)pg->id)); __sync_synchronize(); pg->ready = 1; }
            else for (int tries = 0; !pg->ready; tries++) {   // (rank 0 gone before it published the id: the launcher ends us; the bound is for an outside kill of the launcher)
                if (tries > 120000) throw Error("Env::init: timed out waiting for rank 0's RCCL unique id");
                usleep(1000);
            }
            __sync_synchronize();
            memcpy(id, (const void *)pg->id, sizeof(id));
        } else if (ws && rk && atoi(ws) >= 1) {
            rank() = atoi(rk); nranks() = atoi(ws);
            local = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank();
            check(gt_set_device(local));
            const char *path = getenv("GRAPHTAP_UID_FILE");
            if (!path) throw Error("Env::init: RANK/WORLD_SIZE are set but GRAPHTAP_UID_FILE (shared path for the RCCL unique id) is not");
            if (rank() == 0) {
                check(gt_dist_unique_id(id));
                const std::string tmp = std::string(path) + ".tmp";
                FILE *f = fopen(tmp.c_str(), "wb");
                if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) throw Error("Env::init: cannot write GRAPHTAP_UID_FILE");
                fclose(f);
                if (rename(tmp.c_str(), path) != 0) throw Error("Env::init: cannot publish GRAPHTAP_UID_FILE");
            } else {
                for (int tries = 0;; tries++) {
                    FILE *f = fopen(path, "rb");
                    if (f) { const size_t got = fread(id, 1, sizeof(id), f); fclose(f); if (got == sizeof(id)) break; }
                    if (tries > 60000) throw Error("Env::init: timed out waiting for GRAPHTAP_UID_FILE");
                    usleep(1000);
                }
            }
        } else {
            if (!exchange()) return;            // plain single rank: no communicator at all
            check(gt_set_device(0));
            check(gt_dist_unique_id(id));       // one rank with the exchange layout forced on: RCCL with itself
        }
        check(gt_dist_create(&dist(), id, rank(), nranks()));
    }
    static void finalize() {
        if (dist()) { check(gt_dist_free(dist())); dist() = nullptr; }
        fflush(stdout);
    }
    // sum over ranks, in place (no-op on one rank without a communicator)
    static void all_reduce(uint64_t *v, uint32_t count) {
        if (!dist()) return;
        for (uint32_t i = 0; i < count; i += 64) check(gt_dist_all_reduce_u64(dist(), v + i, count - i < 64 ? count - i : 64));
    }
};
#define GT_MASTER_PRINTF(...) do { if (::gt::Env::is_master()) printf(__VA_ARGS__); } while (0)

class Graph {  // Graph<Weight, Integer_Type, Fractional_Type>, src/mat/graph.hpp:31-71
  public:
    explicit Graph(bool weighted = false) : weighted_(weighted) { gt_graph_options_init(&opt_); }
    // handle-level configuration instead of the GRAPHTAP_* environment variables (gt_graph_options, ABI 3): set before load()
    gt_graph_options &options() { has_opt_ = true; return opt_; }
    ~Graph() {}  // like the reference: free() is explicit (graph.hpp:73-81)

    void load(const std::string &filepath, uint32_t nrows, uint32_t ncols, bool directed = true, bool transpose = false,
              bool self_loops = true, bool acyclic = false, bool parallel_edges = true, Tiling_type = _2DT_,
              Compression_type compression_type = _CSC_) {
        auto t0 = std::chrono::steady_clock::now();
        if (nrows != ncols) throw Error("square matrices only");
        if (compression_type != _TCSC_ && compression_type != _TCSC_CF_) throw Error("only TCSC / TCSC_CF tiles exist in this engine");
        std::ifstream fin(filepath, std::ios::binary | std::ios::ate);
        if (!fin.is_open()) throw Error("Unable to open input file");  // graph.hpp:311-314
        const uint64_t bytes = (uint64_t)fin.tellg(), rec = weighted_ ? 12 : 8;
        // The reference tells text from binary with popen("file -b") (graph.hpp:119-145); here a file is text when its
        // first 4 KiB are printable ASCII / white space.
        std::vector<char> buf((size_t)std::min<uint64_t>(bytes, 4096));
        fin.seekg(0);
        fin.read(buf.data(), (std::streamsize)buf.size());
        bool text = bytes > 0;
        for (uint64_t i = 0; i < buf.size() && text; i++) {
            const unsigned char c = (unsigned char)buf[i];
            text = (c >= 32 && c < 127) || c == '\t' || c == '\n' || c == '\r';
        }
        if (!text && bytes % rec) throw Error("read() failure");                // graph.hpp:331-334
        if (!text && Env::dist() && !getenv("GRAPHTAP_REPLICATED_READ")) {
            // several ranks, binary file: every rank reads ITS 1/p of the records (the reference's parallel read, graph.hpp:308-372)
            // and the build shuffles them to the owners of their rows (Matrix::distribute, matrix.hpp:693-810)
            const uint64_t m = bytes / rec, lo = m * (uint64_t)Env::rank() / (uint64_t)Env::nranks(), hi = m * ((uint64_t)Env::rank() + 1) / (uint64_t)Env::nranks();
            std::vector<char> share((size_t)((hi - lo) * rec));
            fin.seekg((std::streamoff)(lo * rec));
            fin.read(share.data(), (std::streamsize)share.size());
            GT_MASTER_PRINTF("%s: Read %llu edges\n", filepath.c_str(), (unsigned long long)m);
            free();
            gt_graph_flags f{directed, transpose, self_loops, acyclic, parallel_edges};
            check(gt_graph_build_opt(&h_, Env::dist(), share.data(), hi - lo, 0, weighted_, nrows, &f, Env::rank(), Env::nranks(), has_opt_ ? &opt_ : nullptr));
            after_build(compression_type);
        } else {
            buf.resize((size_t)bytes);
            fin.seekg(0);
            fin.read(buf.data(), (std::streamsize)bytes);
            if (text) {
                std::vector<uint32_t> recs = parse_text(buf, weighted_ ? 3 : 2);
                const uint64_t m = recs.size() / (weighted_ ? 3 : 2);
                GT_MASTER_PRINTF("%s: Read %llu edges\n", filepath.c_str(), (unsigned long long)m);
                load_edges(recs.data(), m, nrows, directed, transpose, self_loops, acyclic, parallel_edges, compression_type);
            } else {
                GT_MASTER_PRINTF("%s: Read %llu edges\n", filepath.c_str(), (unsigned long long)(bytes / rec));
                load_edges(buf.data(), bytes / rec, nrows, directed, transpose, self_loops, acyclic, parallel_edges, compression_type);
            }
        }
        GT_MASTER_PRINTF("Ingress time: %f seconds\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    // ASCII edge list, parread_text (graph.hpp:195-304): leading '#' / '%' / empty lines are skipped, every other line is
    // "row col[ weight]" separated by single spaces (a different column count is "read() failure", :250-257), the list
    // ends at the first empty line. Comment lines further down are skipped too (the reference would reject them).
    static std::vector<uint32_t> parse_text(const std::vector<char> &buf, int columns) {
        std::vector<uint32_t> out;
        const char *p = buf.data(), *end = p + buf.size();
        bool started = false;
        while (p < end) {
            const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = eol ? eol : end;
            const char *stop = (le > p && le[-1] == '\r') ? le - 1 : le;
            if (stop == p) { if (started) break; }
            else if (*p == '#' || *p == '%') { /* comment */ }
            else {
                started = true;
                int spaces = 0;
                for (const char *q = p; q < stop; q++) spaces += (*q == ' ');
                if (spaces + 1 != columns) throw Error("read() failure \"" + std::string(p, stop) + "\"");
                const char *q = p;
                for (int c = 0; c < columns; c++) {
                    uint64_t v = 0; bool digit = false;
                    while (q < stop && *q >= '0' && *q <= '9') { v = v * 10 + (uint64_t)(*q - '0'); q++; digit = true; }
                    if (!digit || v > 0xFFFFFFFFull || (q < stop && *q != ' ')) throw Error("read() failure \"" + std::string(p, stop) + "\"");
                    out.push_back((uint32_t)v);
                    q++;
                }
            }
            p = eol ? eol + 1 : end;
        }
        return out;
    }
    void load_edges(const void *edges, uint64_t m, uint32_t num_vertices, bool directed, bool transpose, bool self_loops,
                    bool acyclic, bool parallel_edges, Compression_type compression_type, bool on_device = false) {
        free();
        gt_graph_flags f{directed, transpose, self_loops, acyclic, parallel_edges};
        // (an in-memory list: every rank passes all of it and keeps its tile-row; files go through the distributed build above)
        check(gt_graph_build_opt(&h_, nullptr, edges, m, on_device, weighted_, num_vertices, &f, Env::rank(), Env::nranks(), has_opt_ ? &opt_ : nullptr));
        after_build(compression_type);
    }
    void after_build(Compression_type compression_type) {
        check(gt_graph_info_get(h_, &info));
        compression = compression_type;
        nnz_global = info.nnz_local;
        Env::all_reduce(&nnz_global, 1);
    }
    void free() {
        if (h_) { check(gt_graph_free(h_)); h_ = nullptr; }
    }
    gt_graph *handle() const { return h_; }
    gt_graph_info info{};
    uint64_t nnz_global = 0;
    Compression_type compression = _TCSC_;

  private:
    bool weighted_;
    gt_graph_options opt_{}; bool has_opt_ = false;
    gt_graph *h_ = nullptr;
};

class Vertex_Program {  // src/vp/vertex_program.hpp:23-62
  public:
    Vertex_Program(Graph &G, int kind, bool stationary_, bool gather_depends_on_apply_, bool apply_depends_on_iter_, Ordering_type ot)
        : stationary(stationary_), gather_depends_on_apply(gather_depends_on_apply_), apply_depends_on_iter(apply_depends_on_iter_),
          G_(G), kind_(kind), order_(ot) {}
    virtual ~Vertex_Program() {}

    void initialize() { const auto t0 = clock_(); check(gt_program_initialize(handle())); init_done_(t0); }
    void initialize(Vertex_Program &other) { const auto t0 = clock_(); check(gt_program_initialize_from(handle(), other.handle())); init_done_(t0); }
    void execute(uint32_t num_iterations_ = 0) {  // vp:408-441
        num_iterations = num_iterations_;
        if (!already_initialized_) initialize();
        if (Env::exchange()) {   // several ranks: the loop with the exchange of x over RCCL (csrc/dist.hip)
            if (!Env::dist()) throw Error("several ranks need gt::Env::init() before the first program runs");
            check(gt_dist_execute(Env::dist(), handle(), num_iterations, &stats));
        } else check(gt_program_execute(handle(), num_iterations, &stats));
        const uint32_t first = iteration + 1;
        iteration = stats.iterations;
        for (uint32_t i = first; i <= iteration; i++) GT_MASTER_PRINTF("Iteration:  %u\n", i);  // vp:422
        GT_MASTER_PRINTF("Execute time: %f seconds\n", stats.seconds);                          // vp:437
        // -DTIMING bookkeeping of the reference: one sample per phase and iteration (vp:640-684, 1018-1054, 1611-1637),
        // the first execute()'s wall time (execute_time[0], vp:2143)
        tm_.sg += stats.scatter_gather_ms; tm_.cb += stats.combine_ms; tm_.ap += stats.apply_ms;
        tm_.sg2 += stats.scatter_gather_sq; tm_.cb2 += stats.combine_sq; tm_.ap2 += stats.apply_sq; tm_.n += stats.phase_samples;
        if (tm_.execute_ms < 0) tm_.execute_ms = stats.seconds * 1e3;
    }
    void checksum() {  // vp:1927-1960
        uint64_t sc[2] = {0, 0};
        check(gt_program_checksum(handle(), &sc[0], &sc[1]));
        Env::all_reduce(sc, 2);   // MPI_Allreduce, vp:1940, 1956
        GT_MASTER_PRINTF("Iterations: %u\nValue checksum: %llu\nReachable vertices: %llu\n", iteration, (unsigned long long)sc[0], (unsigned long long)sc[1]);
    }
    // checksum1(), vp:1963-2119: statistics of an integer state (used by apps/deg.cpp on the degrees)
    void checksum1(int field) {
        if (Env::nranks() > 1) throw Error("checksum1 is implemented for one rank");
        const uint32_t H = G_.info.tile_height, n = G_.info.nrows;
        std::vector<uint32_t> v = state_u32(field, H);
        uint64_t sum = 0; double sq = 0; uint32_t maxv = 0, maxi = 0;
        for (uint32_t i = 0; i < H && i < n; i++) { sum += v[i]; sq += (double)v[i] * v[i]; if (v[i] > maxv) { maxv = v[i]; maxi = i; } }
        std::vector<uint32_t> hist(maxv + 1, 0);
        for (uint32_t i = 0; i < H; i++) hist[i < n ? v[i] : 0]++;   // the reference counts padding slots as value 0
        uint32_t mode = 0; for (uint32_t k = 1; k <= maxv; k++) if (hist[k] > hist[mode]) mode = k;
        const double mean = (double)sum / n, sd = std::sqrt(sq / n - mean * mean);
        printf("Value checksum: %llu\n", (unsigned long long)sum);
        printf("Sum: mean +/- std_dev: %f: %f +/- %f\n", (double)sum, mean, sd);   // std::fixed is in effect in the reference
        printf("Mode & skew : %u & %f\n", mode, (mean - mode) / sd);
        printf("Max index : %u\nMax value : %u\n", maxi, maxv);
    }
    virtual std::string print_state(uint32_t i) = 0;
    void display(uint32_t count = 31) {  // vp:2124-2181
        // the reference prints the first states of rank 0's segment = vertices 0..30 at np = 1; on several ranks the
        // segments are ranges of a hashed id space, so the first `count` VERTICES are collected from their owners
        const uint32_t limit = Env::nranks() > 1 ? G_.info.nrows : G_.info.tile_height;
        count = count < limit ? count : limit;
        fetch(count);
        if (getenv("GRAPHTAP_TIMING") && tm_.n) {   // the reference's -DTIMING record, printed by display() (vp:2134-2152)
            auto line = [&](const char *name, double sum, double sq) {
                const double mean = sum / tm_.n, var = sq / tm_.n - mean * mean;
                GT_MASTER_PRINTF("%s time (sum: avg +/- std_dev): %f: %f +/- %f ms\n", name, sum, mean, std::sqrt(var > 0 ? var : 0));
            };
            auto triple = [&](double sum, double sq) {
                const double mean = sum / tm_.n, var = sq / tm_.n - mean * mean;
                GT_MASTER_PRINTF(" %f %f %f", sum, mean, std::sqrt(var > 0 ? var : 0));
            };
            GT_MASTER_PRINTF("Init           time: %f ms\n", tm_.init_ms);
            line("Scatter_gather", tm_.sg, tm_.sg2); line("Combine       ", tm_.cb, tm_.cb2); line("Apply         ", tm_.ap, tm_.ap2);
            GT_MASTER_PRINTF("Execute        time: %f ms\n", tm_.execute_ms);
            GT_MASTER_PRINTF("TIMING %f", tm_.init_ms); triple(tm_.sg, tm_.sg2); triple(tm_.cb, tm_.cb2); triple(tm_.ap, tm_.ap2);
            GT_MASTER_PRINTF(" %f\n", tm_.execute_ms);
        }
        for (uint32_t i = 0; i < count; i++) GT_MASTER_PRINTF("vertex[%u]:%s\n", i, print_state(i).c_str());
    }
    void free() {
        if (h_) { check(gt_program_free(h_)); h_ = nullptr; }
    }
    // handle-level configuration of this program (gt_program_options, ABI 3): e.g. { gt_program_options o; gt_program_options_init(&o); o.timeout_s = 30; V.set_options(o); }
    void set_options(const gt_program_options &o) { check(gt_program_set_options(handle(), &o)); }
    gt_program *handle() {
        if (!h_) {
            gt_program_params p{kind_, order_, G_.compression == _TCSC_CF_ ? GT_TCSC_CF : GT_TCSC, root, alpha, tol};
            check(gt_program_create(&h_, G_.handle(), &p));
        }
        return h_;
    }
    // state of the first `count` vertices (on one rank: the first `count` slots, which are the same thing)
    std::vector<uint32_t> state_u32(int field, uint32_t count) {
        if (Env::nranks() > 1) { std::vector<uint64_t> w = gathered(field, count, false); return std::vector<uint32_t>(w.begin(), w.end()); }
        std::vector<uint32_t> v(count);
        check(gt_program_copy_state(handle(), field, v.data(), count));
        return v;
    }
    std::vector<double> state_f64(int field, uint32_t count) {
        std::vector<double> v(count);
        if (Env::nranks() > 1) { std::vector<uint64_t> w = gathered(field, count, true); memcpy(v.data(), w.data(), (size_t)count * 8); return v; }
        check(gt_program_copy_state(handle(), field, v.data(), count));
        return v;
    }
    // several ranks: every vertex below `count` is owned by exactly one rank; the others contribute 0 to a sum of words
    std::vector<uint64_t> gathered(int field, uint32_t count, bool f64) {
        const uint32_t H = G_.info.tile_height;
        std::vector<uint32_t> vid(H);
        check(gt_graph_vertex_ids(G_.handle(), vid.data(), H));
        std::vector<uint64_t> w(count, 0);
        if (f64) { std::vector<double> v(H); check(gt_program_copy_state(handle(), field, v.data(), H)); for (uint32_t i = 0; i < H; i++) if (vid[i] < count) memcpy(&w[vid[i]], &v[i], 8); }
        else { std::vector<uint32_t> v(H); check(gt_program_copy_state(handle(), field, v.data(), H)); for (uint32_t i = 0; i < H; i++) if (vid[i] < count) w[vid[i]] = v[i]; }
        Env::all_reduce(w.data(), count);
        return w;
    }

    uint32_t num_iterations = 0, iteration = 0, root = 0;
    bool stationary, gather_depends_on_apply, apply_depends_on_iter;
    double alpha = 0.15, tol = 1e-5;  // pr.h:12-13
    gt_exec_stats stats{};

  protected:
    virtual void fetch(uint32_t count) = 0;
    static std::chrono::steady_clock::time_point clock_() { return std::chrono::steady_clock::now(); }
    void init_done_(std::chrono::steady_clock::time_point t0) {
        already_initialized_ = true;
        if (getenv("GRAPHTAP_TIMING")) check(gt_device_synchronize());
        if (tm_.init_ms < 0) tm_.init_ms = std::chrono::duration<double, std::milli>(clock_() - t0).count();   // init_time[0]
    }
    struct { double init_ms = -1, execute_ms = -1, sg = 0, cb = 0, ap = 0, sg2 = 0, cb2 = 0, ap2 = 0; uint64_t n = 0; } tm_;
    Graph &G_;
    int kind_;
    Ordering_type order_;
    gt_program *h_ = nullptr;
    bool already_initialized_ = false;
};

static inline std::string inf_str(uint32_t v) { return v == GT_INF ? "INF" : std::to_string(v); }

#define GT_PROGRAM(NAME, KIND)                                                                                          \
    NAME(Graph &G, bool stationary_ = false, bool gather_depends_on_apply_ = false, bool apply_depends_on_iter_ = false, \
         Ordering_type ot = _ROW_)                                                                                       \
        : Vertex_Program(G, KIND, stationary_, gather_depends_on_apply_, apply_depends_on_iter_, ot) {}

class Deg_Program : public Vertex_Program {  // src/apps/deg.h:27-53
  public:
    GT_PROGRAM(Deg_Program, GT_DEG)
    std::vector<uint32_t> degree;
    std::string print_state(uint32_t i) override { return "Degree=" + std::to_string(degree[i]); }
  protected:
    void fetch(uint32_t n) override { degree = state_u32(GT_F_DEGREE, n); }
};
class PR_Program : public Vertex_Program {  // src/apps/pr.h:21-48
  public:
    GT_PROGRAM(PR_Program, GT_PR)
    std::vector<uint32_t> degree;
    std::vector<double> rank;
    std::string print_state(uint32_t i) override { return "Rank=" + std::to_string(rank[i]) + ",Degree=" + std::to_string(degree[i]); }
  protected:
    void fetch(uint32_t n) override { degree = state_u32(GT_F_DEGREE, n); rank = state_f64(GT_F_RANK, n); }
};
class BFS_Program : public Vertex_Program {  // src/apps/bfs.h:33-82
  public:
    GT_PROGRAM(BFS_Program, GT_BFS)
    std::vector<uint32_t> parent, hops;
    std::string print_state(uint32_t i) override { return "Parent=" + std::to_string(parent[i]) + ",Hops=" + inf_str(hops[i]); }
  protected:
    void fetch(uint32_t n) override { parent = state_u32(GT_F_PARENT, n); hops = state_u32(GT_F_HOPS, n); }
};
class SSSP_Program : public Vertex_Program {  // src/apps/sssp.h:29-71
  public:
    GT_PROGRAM(SSSP_Program, GT_SSSP)
    std::vector<uint32_t> distance;
    std::string print_state(uint32_t i) override { return "Distance=" + inf_str(distance[i]); }
  protected:
    void fetch(uint32_t n) override { distance = state_u32(GT_F_DISTANCE, n); }
};
class CC_Program : public Vertex_Program {  // src/apps/cc.h:29-60
  public:
    GT_PROGRAM(CC_Program, GT_CC)
    std::vector<uint32_t> label;
    std::string print_state(uint32_t i) override { return "Label=" + std::to_string(label[i]); }
  protected:
    void fetch(uint32_t n) override { label = state_u32(GT_F_LABEL, n); }
};
#undef GT_PROGRAM

}  // namespace gt
