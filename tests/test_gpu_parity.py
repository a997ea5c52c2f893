"""GPU parity tests proper: the HIP path, called through the C ABI (graphtap_amd's ctypes mirror of
include/graphtap_amd.h), against (a) the golden vectors produced by the unmodified reference and
(b) the CPU oracle on the same seeded inputs. Integer programs bit-exact; PageRank within the
north-star tolerance of 1e-6 relative (fp64 atomics re-associate the sums).

Tests read like src/apps/*.cpp on purpose."""
import ctypes as C

import numpy as np
import pytest

from conftest import CASES, CF_CASES, load_case, load_cf_case

pytestmark = pytest.mark.gpu

PR_RTOL = 1e-6   # BASELINE.json north_star: "within 1e-6 relative for PageRank ranks"


@pytest.fixture(scope="module")
def gt():
    import graphtap_amd as gt
    gt._lib.require_gpu()          # fails loudly: there is no CPU fallback
    gt._lib.check(gt._lib.lib().gt_set_device(0))
    return gt


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def run_pr(gt, edges, nv, iters, cf=True):
    """apps/pr.cpp:23-54 (cf) / apps/pr1.cpp (plain TCSC)."""
    G = gt.Graph()
    G.load_edges(edges, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_ if cf else gt._TCSC_, rank=0, nranks=1)
    V = gt.Deg_Program(G, True, False, False, gt._COL_)
    V.execute(1)
    VR = gt.PR_Program(G, True, False, False, gt._ROW_)
    VR.initialize(V)
    V.free()
    VR.execute(iters)
    out = dict(VR.V, iterations=VR.iteration, checksum=VR.checksum(out=None), stats=VR.stats, nnz=G.info.nnz_local)
    VR.free(); G.free()
    return out


def run_min(gt, app, edges, nv, root=0):
    """apps/bfs.cpp, apps/sssp.cpp, apps/cc.cpp."""
    if app == "bfs":
        G = gt.Graph(); G.load_edges(edges, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = gt.BFS_Program(G, False, False, True, gt._ROW_); P.root = root
    elif app == "sssp":
        G = gt.Graph(weighted=True); G.load_edges(edges, nv, nv, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = gt.SSSP_Program(G, False, True, False, gt._ROW_); P.root = root
    else:
        G = gt.Graph(); G.load_edges(edges, nv, nv, False, False, True, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = gt.CC_Program(G, False, True, False, gt._ROW_)
    P.execute()
    out = dict(P.V, iterations=P.iteration, checksum=P.checksum(out=None), display=P.display(out=None), stats=P.stats)
    P.free(); G.free()
    return out


# ------------------------------------------------------------------------------- ingest
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("app", ["deg", "pr", "bfs", "sssp", "cc"])
def test_ingest_builds_the_oracles_tcsc(gt, O, name, app):
    c = load_case(name)
    weighted = (app == "sssp")
    e = c["wedges"] if weighted else c["edges"]
    f = O.APP_FLAGS[app]
    og = O.OracleGraph(e, c["num_vertices"], weighted=weighted, **f)
    G = gt.Graph(weighted=weighted)
    G.load_edges(e, c["num_vertices"], c["num_vertices"], f["directed"], f["transpose"], f["self_loops"], f["acyclic"],
                 f["parallel_edges"], gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    i = G.info
    assert (i.nnz_local, i.nnzrows, i.nnzcols, i.tile_height, i.nrows) == (og.nnz, og.nnzrows, og.nnzcols, og.H, og.nrows)
    assert i.seg_stride == max(og.nnzcols, 1)
    cc = og.class_counts()
    assert (i.regular, i.source_rows, i.sink_cols) == (cc["regular"], cc["source_rows"], cc["sink_cols"])
    t = G.tile_to_host()
    assert (t["JA"][:og.nnzcols + 1] == og.JA).all() and (t["IA"] == og.IA).all()
    assert (t["JC"] == og.JC).all() and (t["IR"] == og.IR).all()
    if weighted:
        assert (t["A"] == og.A).all()
    G.free()


def test_ingest_edge_cases(gt, O):
    # empty edge list
    G = gt.Graph(); G.load_edges(np.zeros((0, 2), np.uint32), 10, 10, True, True, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    assert (G.info.nnz_local, G.info.nnzrows, G.info.nnzcols, G.info.tile_height) == (0, 0, 0, 12)
    P = gt.CC_Program(G, False, True, False, gt._ROW_); P.execute()
    assert (P.V["label"] == np.arange(12)).all() and P.iteration == 1
    P.free(); G.free()
    # only self loops, dropped by the BFS flags -> empty graph, root reaches nothing else
    e = np.array([[3, 3], [4, 4]], np.uint32)
    G = gt.Graph(); G.load_edges(e, 8, 8, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    assert G.info.nnz_local == 0
    P = gt.BFS_Program(G, False, False, True, gt._ROW_); P.root = 3; P.execute()
    r = O.run_app("bfs", e, 8, root=3)
    assert (P.V["hops"] == r["hops"]).all() and (P.V["parent"] == r["parent"]).all() and P.iteration == r["iterations"]
    P.free(); G.free()
    # the maximum id N itself is a legal vertex (nrows = N + 1); N + 1 is not
    e = np.array([[0, 8], [8, 0]], np.uint32)
    G = gt.Graph(); G.load_edges(e, 8, 8, True, True, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    assert G.info.nnz_local == 2
    G.free()
    with pytest.raises(gt.GraphTapError, match="vertex id"):
        gt.Graph().load_edges(np.array([[0, 9]], np.uint32), 8, 8, True, True, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    # acyclic flag (graph.hpp:343-346)
    c = load_case("rmat8")
    og = O.OracleGraph(c["edges"], 256, directed=True, transpose=False, self_loops=False, acyclic=True, parallel_edges=False)
    G = gt.Graph(); G.load_edges(c["edges"], 256, 256, True, False, False, True, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    t = G.tile_to_host()
    assert G.info.nnz_local == og.nnz and (t["IA"] == og.IA).all() and (t["JA"][:og.nnzcols + 1] == og.JA).all()
    G.free()


def test_device_rmat_generator_matches_host(gt):
    from graphtap_amd.rmat import rmat_edges
    L = gt._lib.lib()
    for weighted in (False, True):
        host = rmat_edges(14, 16, seed=3, weighted=weighted, first=1000, count=100000)
        d = C.c_void_p(); gt._lib.check(L.gt_malloc(C.byref(d), host.nbytes))
        gt._lib.check(L.gt_rmat_generate(d, 14, 3, int(weighted), 1000, 100000, None))
        got = np.zeros_like(host); gt._lib.check(L.gt_memcpy_d2h(got.ctypes.data_as(C.c_void_p), d, host.nbytes))
        gt._lib.check(L.gt_free(d))
        assert (got == host).all()


# ------------------------------------------------------------------------------- kernel-level SpMV
@pytest.mark.parametrize("name", ["tiny", "rmat10", "rmat12"])
def test_spmv_kernels_match_oracle(gt, O, name):
    c = load_case(name)
    L = gt._lib.lib()
    rng = np.random.RandomState(11)

    def dev(a):
        d = C.c_void_p(); gt._lib.check(L.gt_malloc(C.byref(d), max(a.nbytes, 1)))
        gt._lib.check(L.gt_memcpy_h2d(d, a.ctypes.data_as(C.c_void_p), a.nbytes)); return d

    def host(d, like):
        o = np.zeros_like(like); gt._lib.check(L.gt_memcpy_d2h(o.ctypes.data_as(C.c_void_p), d, o.nbytes)); return o

    f = O.APP_FLAGS["pr"]
    og = O.OracleGraph(c["edges"], c["num_vertices"], **f)
    G = gt.Graph(); G.load_edges(c["edges"], c["num_vertices"], c["num_vertices"], True, True, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    x = rng.rand(og.nnzcols); y0 = rng.rand(og.nnzrows)
    want = og.spmv_plus_f64(x, y0.copy())
    dx, dy = dev(x), dev(y0)
    gt._lib.check(L.gt_spmv(G._h, gt._lib.GT_PLUS_F64, dx, dy, None))
    got = host(dy, y0)
    assert np.allclose(got, want, rtol=1e-12, atol=0)
    L.gt_free(dx); L.gt_free(dy); G.free()

    f = O.APP_FLAGS["sssp"]
    og = O.OracleGraph(c["wedges"], c["num_vertices"], weighted=True, **f)
    G = gt.Graph(weighted=True); G.load_edges(c["wedges"], c["num_vertices"], c["num_vertices"], True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    x = rng.randint(0, 1000, og.nnzcols).astype(np.uint32); x[rng.rand(og.nnzcols) < 0.5] = gt.INF
    y0 = rng.randint(0, 2000, og.nnzrows).astype(np.uint32); y0[rng.rand(og.nnzrows) < 0.5] = gt.INF
    for sr, ograph in ((gt._lib.GT_MINPLUS_U32, og),):
        want = ograph.spmv_min_u32(x, y0.copy())
        dx, dy = dev(x), dev(y0)
        gt._lib.check(L.gt_spmv(G._h, sr, dx, dy, None))
        assert (host(dy, y0) == want).all()
        L.gt_free(dx); L.gt_free(dy)
    G.free()


@pytest.mark.parametrize("name", CF_CASES + ["rmat16", "weighted"])
def test_tcsc_cf_format_matches_reference_and_oracle(gt, O, name):
    """gt_graph_tile_cf: IA with the source rows of every column in its tail IN THE REFERENCE'S SWAP ORDER and the four
    pair lists (ds/compressed_column.hpp:603-1120), bit for bit against the arrays dumped from the unmodified reference
    (tests/golden/tcsc_cf.npz) and against the oracle's restatement on a larger R-MAT; with weights, A must travel with IA."""
    weighted = name == "weighted"
    if name == "rmat16":
        from graphtap_amd.rmat import rmat_edges
        edges, nv, ref = rmat_edges(16, 16, 4), 1 << 16, None
    elif weighted:
        edges, nv, ref = load_case("rmat12")["wedges"], 1 << 12, None
    else:
        edges, nv, ref = load_cf_case(name)
    og = O.OracleGraph(edges, nv, weighted=weighted, **O.APP_FLAGS["pr"])
    want = og.tcsc_cf()
    G = gt.Graph(weighted=weighted); G.load_edges(edges, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
    got = G.tile_cf_to_host()
    keys = ["IA", "JA_REG_R_NNZ_C"] + [p + l for l in G.CF_LISTS for p in ("JA_", "JC_")]
    for k in keys:
        assert got[k].shape == want[k].shape and (got[k] == want[k]).all(), k
        if ref is not None: assert (got[k] == ref[k]).all(), k
    for l in G.CF_LISTS:
        assert got["NC_" + l] == want["NC_" + l]
    if weighted:
        assert (got["A"] == want["A"]).all()
        t = G.tile_to_host()
        pairs = lambda ia, a: np.sort(ia.astype(np.uint64) << 32 | a)
        for j in range(0, og.nnzcols, 7):
            a, b = t["JA"][j], t["JA"][j + 1]
            assert (pairs(got["IA"][a:b], got["A"][a:b]) == pairs(t["IA"][a:b], t["A"][a:b])).all()
    assert (G.tile_cf_to_host()["IA"] == got["IA"]).all()   # second call: the same arrays, not a rebuild of something else
    G.free()


@pytest.mark.parametrize("variant", ["edge", "pb"])
@pytest.mark.parametrize("name", CF_CASES)
def test_tcsc_cf_pair_list_spmv_matches_oracle(gt, O, name, variant, monkeypatch):
    """gt_spmv_cf = spmv_stationary's TCSC_CF branch (vp:1243-1317) for each of its three conditions; integer-valued
    messages make the f64 sums exact in any order. Under both layouts of x (the pb build's hubs first, the edge build's)."""
    monkeypatch.setenv("GRAPHTAP_SPMV", variant)
    edges, nv, _ = load_cf_case(name)
    L = gt._lib.lib()
    og = O.OracleGraph(edges, nv, **O.APP_FLAGS["pr"])
    G = gt.Graph(); G.load_edges(edges, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
    rng = np.random.RandomState(3)
    x = rng.randint(1, 1000, og.nnzcols).astype(np.float64)
    dx, dy = C.c_void_p(), C.c_void_p()
    gt._lib.check(L.gt_malloc(C.byref(dx), max(x.nbytes, 8))); gt._lib.check(L.gt_malloc(C.byref(dy), max(og.nnzrows * 8, 8)))
    gt._lib.check(L.gt_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes))
    for first, running, last in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 1, 1)):
        y0 = rng.randint(0, 50, og.nnzrows).astype(np.float64)
        want = og.spmv_cf_plus_f64(x, y0.copy(), first=first, running=running, last=last)
        gt._lib.check(L.gt_memcpy_h2d(dy, y0.ctypes.data_as(C.c_void_p), y0.nbytes))
        gt._lib.check(L.gt_spmv_cf(G._h, dx, dy, first, running, last, None))
        got = np.zeros_like(y0); gt._lib.check(L.gt_memcpy_d2h(got.ctypes.data_as(C.c_void_p), dy, got.nbytes))
        assert (got == want).all(), (first, running, last)
    L.gt_free(dx); L.gt_free(dy); G.free()


def test_tcsc_cf_pagerank_on_the_pair_lists(gt, known_answers, monkeypatch):
    """A GT_TCSC_CF PageRank on the edge-parallel variant runs over the pair lists (regular rows every iteration, source
    rows on the last one): the reference's ranks, fixed count and converge mode, and the statistics say so."""
    monkeypatch.setenv("GRAPHTAP_SPMV", "edge")
    for name in CASES:
        c = load_case(name); nv = c["num_vertices"]; n = nv + 1
        for iters, key in ((20, "np1_pr20"), (1, "np1_pr1"), (0, "np1_prconv_cf")):
            r = run_pr(gt, c["edges"], nv, iters, cf=True)
            ref = c[key + "_c"]
            assert r["iterations"] == known_answers[name][key]["iterations"]
            assert (np.abs(r["rank"][:n] - ref) / ref).max() < PR_RTOL and (r["degree"][:n] == c[key + "_a"]).all()
            assert r["stats"].cf_filtered_iterations == (r["iterations"] if iters == 0 else iters - 1)
        r = run_pr(gt, c["edges"], nv, 20, cf=False)
        assert r["stats"].cf_filtered_iterations == 0 and (np.abs(r["rank"][:n] - c["np1_pr1app20_c"]) / c["np1_pr1app20_c"]).max() < PR_RTOL


# ------------------------------------------------------------------------------- the five apps vs the reference
@pytest.mark.parametrize("name", CASES)
def test_deg_app_matches_reference(gt, name, known_answers):
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1
    G = gt.Graph(); G.load_edges(c["edges"], nv, nv, True, False, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)   # deg.cpp:27-35
    V = gt.Deg_Program(G, True, False, False, gt._ROW_); V.execute(1)
    assert (V.V["degree"][:n] == c["np1_deg_a"]).all()
    ka = known_answers[name]["np1_deg"]
    assert V.checksum(out=None) == (ka["checksum"], ka["reachable"]) and V.iteration == 1
    V.free(); G.free()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("iters", [1, 3, 20])
def test_pagerank_matches_reference(gt, name, iters, known_answers):
    c = load_case(name); n = c["num_vertices"] + 1
    key = "np1_pr%d" % iters
    for cf in (True, False):
        r = run_pr(gt, c["edges"], c["num_vertices"], iters, cf)
        assert r["iterations"] == iters
        assert (r["degree"][:n] == c[key + "_a"]).all()
        ref = c[key + "_c"]
        rel = np.abs(r["rank"][:n] - ref) / ref
        assert rel.max() < PR_RTOL, rel.max()
    ka = known_answers[name][key]
    assert r["checksum"][1] == ka["reachable"] and abs(r["checksum"][0] - ka["checksum"]) <= 1  # truncating sum: order-sensitive by 1 ulp cases


@pytest.mark.parametrize("variant", ["pb", "pb_f32msg", "edge"])
@pytest.mark.parametrize("name", ["rmat10", "rmat12"])
def test_pagerank_lean_iterations_leave_the_full_state(gt, name, variant, monkeypatch):
    """A fixed-count PageRank run owned by the library skips, in every iteration but the last two, the loads and stores of
    vertex state that nobody can observe (rank does not depend on the previous rank, pr.h:43-47; gt_internal.h pr_state).
    After execute(n) every field -- rank AND the changed flags of the last iteration (vp:1671-1691) -- equals what the full
    applicator (GRAPHTAP_PR_LEAN_STATE=0) leaves, also when a second execute continues the run."""
    c = load_case(name); nv = c["num_vertices"]
    monkeypatch.setenv("GRAPHTAP_SPMV", variant)

    def run(lean, plan):
        monkeypatch.setenv("GRAPHTAP_PR_LEAN_STATE", "1" if lean else "0")
        G = gt.Graph(); G.load_edges(c["edges"], nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
        V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
        P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V)
        out = []
        for iters in plan:
            P.execute(iters)
            H = G.info.tile_height
            act = np.zeros(H, np.uint8)
            gt._lib.check(gt._lib.lib().gt_program_copy_state(P._handle(), gt._lib.GT_F_ACTIVE, act.ctypes.data_as(C.c_void_p), H))
            out.append((P.iteration, P.V["rank"].copy(), act, P.checksum(out=None)))
        P.free(); V.free(); G.free()
        return out

    for plan in ([1], [2], [3], [7], [3, 8], [2, 3, 4]):
        full, lean = run(False, plan), run(True, plan)
        for (it0, r0, a0, k0), (it1, r1, a1, k1) in zip(full, lean):
            assert it0 == it1 and k0[1] == k1[1] and abs(k0[0] - k1[0]) <= 1
            assert (np.abs(r0 - r1) <= 1e-12 * np.abs(r0)).all()   # same arithmetic; the LDS atomics' order is free
            assert (a0 == a1).all()


@pytest.mark.parametrize("name", CASES)
def test_pagerank_converge_mode_matches_reference(gt, name, known_answers):
    c = load_case(name); n = c["num_vertices"] + 1
    for cf, key in ((False, "np1_prconv_tcsc"), (True, "np1_prconv_cf")):
        r = run_pr(gt, c["edges"], c["num_vertices"], 0, cf)
        assert r["iterations"] == known_answers[name][key]["iterations"]
        ref = c[key + "_c"]
        assert (np.abs(r["rank"][:n] - ref) / ref).max() < PR_RTOL


@pytest.mark.parametrize("name", CASES)
def test_pagerank_converge_mode_is_deterministic_with_f32_messages(gt, name, known_answers, monkeypatch):
    """The benched variant (pb_f32msg) in converge mode: the reference's iteration count and ranks on both layouts (hubs-first
    and -- GRAPHTAP_FORCE_EXCHANGE -- the exchange layout), because converge mode runs f64 messages (gt_program_prepare;
    apps/pr.h:43-47, vp:1885-1923); a fixed-count run after it on the same program is back to f32 messages."""
    c = load_case(name); n = c["num_vertices"] + 1
    monkeypatch.setenv("GRAPHTAP_SPMV", "pb_f32msg")
    for cf, key in ((False, "np1_prconv_tcsc"), (True, "np1_prconv_cf")):
        r = run_pr(gt, c["edges"], c["num_vertices"], 0, cf)
        assert r["iterations"] == known_answers[name][key]["iterations"]
        ref = c[key + "_c"]
        assert (np.abs(r["rank"][:n] - ref) / ref).max() < PR_RTOL
    # one program: execute(20) [f32 messages], initialize, execute() [f64 messages], initialize, execute(20) again
    G = gt.Graph(); G.load_edges(c["edges"], c["num_vertices"], c["num_vertices"], True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
    P = gt.PR_Program(G, True, False, False, gt._ROW_)
    P.initialize(V); P.execute(20); a = P.V["rank"].copy()
    P.initialize(V); P.execute(); assert P.iteration == known_answers[name]["np1_prconv_cf"]["iterations"]
    assert (np.abs(P.V["rank"][:n] - c["np1_prconv_cf_c"]) / c["np1_prconv_cf_c"]).max() < PR_RTOL
    P.initialize(V); P.execute(20); b = P.V["rank"]
    assert (np.abs(a[:n] - b[:n]) / b[:n]).max() < 1e-12   # same arithmetic (the LDS atomics' order is free)
    assert (np.abs(b[:n] - c["np1_pr20_c"]) / c["np1_pr20_c"]).max() < PR_RTOL
    P.free(); V.free(); G.free()


@pytest.mark.parametrize("name", CASES)
def test_bfs_sssp_cc_bit_exact(gt, name, known_answers):
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1
    roots = [(c["root"], "")] + ([(0, "0")] if c["root"] != 0 and "np1_bfs0_a" in c else [])
    for root, sfx in roots:
        r = run_min(gt, "bfs", c["edges"], nv, root)
        assert (r["parent"][:n] == c["np1_bfs%s_a" % sfx]).all() and (r["hops"][:n] == c["np1_bfs%s_b" % sfx]).all()
        ka = known_answers[name]["np1_bfs" + sfx]
        assert (r["iterations"], r["checksum"]) == (ka["iterations"], (ka["checksum"], ka["reachable"]))
        r = run_min(gt, "sssp", c["wedges"], nv, root)
        assert (r["distance"][:n] == c["np1_sssp%s_a" % sfx]).all()
        ka = known_answers[name]["np1_sssp" + sfx]
        assert (r["iterations"], r["checksum"]) == (ka["iterations"], (ka["checksum"], ka["reachable"]))
    r = run_min(gt, "cc", c["edges"], nv)
    assert (r["label"][:n] == c["np1_cc_a"]).all()
    ka = known_answers[name]["np1_cc"]
    assert (r["iterations"], r["checksum"]) == (ka["iterations"], (ka["checksum"], ka["reachable"]))


@pytest.mark.parametrize("name", CASES)
def test_tcsc_cf_computation_filtering(gt, name, known_answers, monkeypatch):
    """TCSC_CF's computation filtering (compressed_column.hpp:671-708, vp:1264-1317): the entries of SOURCE rows (vertices
    with in-edges but no out-edges) are left out of every SpMV but the last one -- never in converge mode, where the
    reference's source rows end at exactly alpha (SURVEY trap 5). Here they live in chunks of their own that PageRank under
    GT_TCSC_CF does not launch until the last iteration: the golden vectors must come out with the filtering on (default)
    and off, and the statistics must say that it happened."""
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1
    for env in (None, "1"):
        if env: monkeypatch.setenv("GRAPHTAP_NO_CF_FILTER", env)
        r = run_pr(gt, c["edges"], nv, 20, cf=True)
        ref = c["np1_pr20_c"]
        assert (np.abs(r["rank"][:n] - ref) / ref).max() < PR_RTOL and (r["degree"][:n] == c["np1_pr20_a"]).all()
        assert r["stats"].cf_filtered_iterations == (0 if env else 19)
        r = run_pr(gt, c["edges"], nv, 0, cf=True)
        ref = c["np1_prconv_cf_c"]
        assert r["iterations"] == known_answers[name]["np1_prconv_cf"]["iterations"]
        assert (np.abs(r["rank"][:n] - ref) / ref).max() < PR_RTOL
        assert r["stats"].cf_filtered_iterations == (0 if env else r["iterations"])
    monkeypatch.delenv("GRAPHTAP_NO_CF_FILTER")
    r = run_pr(gt, c["edges"], nv, 20, cf=False)      # plain TCSC (apps/pr1.cpp): nothing is filtered
    assert r["stats"].cf_filtered_iterations == 0


@pytest.mark.parametrize("mode", ["1", "0"])
@pytest.mark.parametrize("name", CASES)
def test_sparse_frontier_spmspv_is_bit_exact(gt, name, mode, known_answers, monkeypatch):
    """The reference switches a non-stationary iteration to its sparse SpMSpV when few columns are active (vp:754-784,
    1475-1489); here an iteration whose active columns hold <= nnz/64 entries runs the frontier-driven kernel
    (kernels.hip) instead of the streaming pass. GRAPHTAP_SPMSPV=1 forces it for every iteration, =0 forbids it: labels
    AND iteration counts must equal the reference's either way (y is a running minimum: the order of arrival is free)."""
    monkeypatch.setenv("GRAPHTAP_SPMSPV", mode)
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1; k = known_answers[name]
    r = run_min(gt, "bfs", c["edges"], nv, c["root"])
    assert (r["parent"][:n] == c["np1_bfs_a"]).all() and (r["hops"][:n] == c["np1_bfs_b"]).all() and r["iterations"] == k["np1_bfs"]["iterations"]
    assert (r["stats"].spmspv_iterations > 0) == (mode == "1")
    r = run_min(gt, "cc", c["edges"], nv)
    assert (r["label"][:n] == c["np1_cc_a"]).all() and r["iterations"] == k["np1_cc"]["iterations"]
    r = run_min(gt, "sssp", c["wedges"], nv, c["root"])
    assert (r["distance"][:n] == c["np1_sssp_a"]).all() and r["iterations"] == k["np1_sssp"]["iterations"]
    assert (r["stats"].spmspv_iterations > 0) == (mode == "1")


@pytest.mark.parametrize("lists", ["1", "0"])
@pytest.mark.parametrize("spmspv", [None, "1"])
def test_frontier_lists_are_bit_exact(gt, O, lists, spmspv, known_answers, monkeypatch):
    """Frontier lists (one rank, converge mode): while the last apply changed few vertices, the messenger touches only the
    slots of the previous and the new frontier, the SpMSpV takes its columns from the list and emits the rows it lowers, and
    apply visits those rows only. Labels, parents AND iteration counts must be the reference's with the lists on and off, with
    the sparse path by size and forced; mid-size R-MAT against the oracle, where big and small frontiers alternate."""
    from graphtap_amd.rmat import rmat_edges
    monkeypatch.setenv("GRAPHTAP_FRONTIER_LISTS", lists)
    if spmspv: monkeypatch.setenv("GRAPHTAP_SPMSPV", spmspv)

    def check_lists(r):   # by size, a graph this small never takes the sparse path (nnz / 1024 entries): only the forced mode must
        if lists == "0": assert r["stats"].list_iterations == 0
        elif spmspv == "1": assert r["stats"].list_iterations > 0
    for name in CASES:
        c = load_case(name); nv = c["num_vertices"]; n = nv + 1; k = known_answers[name]
        for root, tag in ((c["root"], ""), (0, "0")) if c["root"] != 0 else ((0, ""),):
            r = run_min(gt, "bfs", c["edges"], nv, root)
            assert (r["parent"][:n] == c["np1_bfs%s_a" % tag]).all() and (r["hops"][:n] == c["np1_bfs%s_b" % tag]).all()
            assert r["iterations"] == k["np1_bfs" + tag]["iterations"]
            check_lists(r)
            r = run_min(gt, "sssp", c["wedges"], nv, root)
            assert (r["distance"][:n] == c["np1_sssp%s_a" % tag]).all() and r["iterations"] == k["np1_sssp" + tag]["iterations"]
            check_lists(r)
        r = run_min(gt, "cc", c["edges"], nv)
        assert (r["label"][:n] == c["np1_cc_a"]).all() and r["iterations"] == k["np1_cc"]["iterations"]
        if lists == "0": assert r["stats"].list_iterations == 0
    w = rmat_edges(17, 16, 5, weighted=True); e = np.ascontiguousarray(w[:, :2]); nv = 1 << 17
    for root in (0, 77):
        want = O.run_app("bfs", e, nv, root=root); got = run_min(gt, "bfs", e, nv, root)
        assert (got["parent"] == want["parent"]).all() and (got["hops"] == want["hops"]).all() and got["iterations"] == want["iterations"]
        want = O.run_app("sssp", w, nv, root=root); got = run_min(gt, "sssp", w, nv, root)
        assert (got["distance"] == want["distance"]).all() and got["iterations"] == want["iterations"]
    want = O.run_app("cc", e, nv); got = run_min(gt, "cc", e, nv)
    assert (got["label"] == want["label"]).all() and got["iterations"] == want["iterations"]
    if lists == "1": assert got["stats"].list_iterations > 0   # the tail iterations of CC


@pytest.mark.parametrize("mode", ["all", "f2", "default", "off"])
def test_hybrid_streaming_pass_is_bit_exact(gt, O, mode, known_answers, monkeypatch):
    """The hybrid pass of the min programs (pb.hip, pb_run; round 4): the heavy windows of x (candidates: at least 1/256 of all
    entries each) whose active columns hold a small fraction of their entries leave the streaming pass -- those columns' entries go
    through column-driven kernels (atomicMin on y), the short columns eight lanes each, the long ones a workgroup per 8 192 entries
    -- decided per window on the device, no host round trip. Labels, distances, parents AND iteration counts must be the
    reference's / the oracle's with EVERY window with an active column sent that way ("all": every window a candidate, threshold
    1), with a low threshold ("f2": a mix of streamed and column-driven windows), by default and with the pass switched off; the sparse paths that would otherwise take the small frontiers are forbidden so that every iteration is a
    (hybrid) streaming pass, bottom-up BFS steps and CC's first-entry shortcut included in the forbidden ones."""
    from graphtap_amd.rmat import rmat_edges
    L = gt._lib.lib()
    if mode == "all": monkeypatch.setenv("GRAPHTAP_HYBRID_F", "1"); monkeypatch.setenv("GRAPHTAP_HYBRID_MIN_DIV", "4000000000")   # every window is a candidate
    if mode == "f2": monkeypatch.setenv("GRAPHTAP_HYBRID_F", "2"); monkeypatch.setenv("GRAPHTAP_HYBRID_MIN_DIV", "4000000000")
    monkeypatch.setenv("GRAPHTAP_HYBRID", "0" if mode == "off" else "1")   # (off by default: measured, +-1 %)
    for k in ("GRAPHTAP_SPMSPV", "GRAPHTAP_BFS_BOTTOM_UP", "GRAPHTAP_CC_FIRST", "GRAPHTAP_TAIL_KERNEL"): monkeypatch.setenv(k, "0")
    for name in CASES:
        c = load_case(name); nv = c["num_vertices"]; n = nv + 1; k = known_answers[name]
        r = run_min(gt, "bfs", c["edges"], nv, c["root"])
        assert (r["parent"][:n] == c["np1_bfs_a"]).all() and (r["hops"][:n] == c["np1_bfs_b"]).all() and r["iterations"] == k["np1_bfs"]["iterations"]
        r = run_min(gt, "sssp", c["wedges"], nv, c["root"])
        assert (r["distance"][:n] == c["np1_sssp_a"]).all() and r["iterations"] == k["np1_sssp"]["iterations"]
        r = run_min(gt, "cc", c["edges"], nv)
        assert (r["label"][:n] == c["np1_cc_a"]).all() and r["iterations"] == k["np1_cc"]["iterations"]
    raw = C.CDLL(gt._lib.LIB_PATH); raw.gt_graph_hybrid_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    w = rmat_edges(18, 16, 9, weighted=True); e = np.ascontiguousarray(w[:, :2]); nv = 1 << 18
    seen = {}
    for app, edges, flags, root in (("sssp", w, (True, True, False, False, False), 0), ("cc", e, (False, False, True, False, False), 0), ("bfs", e, (False, False, False, False, False), 5)):
        want = O.run_app(app, edges, nv, root=root) if app != "cc" else O.run_app(app, edges, nv)
        G = gt.Graph(weighted=(app == "sssp")); G.load_edges(edges, nv, nv, *flags, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = {"bfs": gt.BFS_Program, "cc": gt.CC_Program, "sssp": gt.SSSP_Program}[app](G, False, app != "bfs", app == "bfs", gt._ROW_)
        P.root = root; P.execute()
        key = {"bfs": "parent", "cc": "label", "sssp": "distance"}[app]
        assert (P.V[key] == want[key]).all() and P.iteration == want["iterations"], app
        st = (C.c_uint64 * 4)(); gt._lib.check(raw.gt_graph_hybrid_stats(G._h, st, 0)); seen[app] = tuple(int(v) for v in st)
        P.free(); G.free(); want["graph"].close()
    print("hybrid mode %s: (passes, entries to the column kernels, entries left out of the stream, windows left out) per app: %s" % (mode, seen))
    if mode == "off": assert all(v[0] == 0 for v in seen.values())
    if mode == "all": assert all(v[0] > 0 and v[1] > 0 and v[2] > 0 for v in seen.values()), seen


@pytest.mark.parametrize("early", [None, "0", "collect"])
@pytest.mark.parametrize("mode", ["1", "0", None])
def test_bfs_bottom_up_steps_are_bit_exact(gt, O, mode, early, known_answers, monkeypatch):
    """BFS on a symmetric graph (apps/bfs.cpp loads with directed = false) may run an iteration bottom-up: the unreached rows
    look at ALL their neighbours and take the minimum id among those on the current level -- what the push sweep's
    min-combiner (bfs.h:61-63) leaves in y. Parents, hops and iteration counts must be the reference's with the step forced
    on every iteration (=1), forbidden (=0) and chosen by size (unset); a directed graph never takes it. By default a row stops
    at its FIRST neighbour on the level (entries ascend by row, rows by vertex id: the first hit is the minimum), which also lets
    the step run against a large frontier; GRAPHTAP_BFS_BU_EARLY=0 is the form that looks at every neighbour."""
    from graphtap_amd.rmat import rmat_edges
    if mode is not None: monkeypatch.setenv("GRAPHTAP_BFS_BOTTOM_UP", mode)
    # default: the step works from the row bitmaps BFS's apply kernels keep (reached / current level); "collect": a pass over the vertex
    # states collects the unreached rows and the level for every step (GRAPHTAP_BFS_BU_MAPS=0); "0": that pass + every neighbour looked at
    if early == "0": monkeypatch.setenv("GRAPHTAP_BFS_BU_EARLY", "0")
    if early == "collect": monkeypatch.setenv("GRAPHTAP_BFS_BU_MAPS", "0")
    if early is not None and mode == "0": pytest.skip("no bottom-up step at all: covered by the default form")
    for name in CASES:
        c = load_case(name); nv = c["num_vertices"]; n = nv + 1; k = known_answers[name]
        for root, tag in ((c["root"], ""), (0, "0")) if c["root"] != 0 else ((0, ""),):
            r = run_min(gt, "bfs", c["edges"], nv, root)
            assert (r["parent"][:n] == c["np1_bfs%s_a" % tag]).all() and (r["hops"][:n] == c["np1_bfs%s_b" % tag]).all()
            assert r["iterations"] == k["np1_bfs" + tag]["iterations"]
            if mode == "1": assert r["stats"].spmspv_iterations == r["iterations"]   # every iteration took a sparse path
    e = rmat_edges(18, 16, 9); nv = 1 << 18
    for root in (0, 5, 131071):
        want = O.run_app("bfs", e, nv, root=root); got = run_min(gt, "bfs", e, nv, root)
        assert (got["parent"] == want["parent"]).all() and (got["hops"] == want["hops"]).all() and got["iterations"] == want["iterations"]
    # a DIRECTED graph under the BFS program: the column of a vertex does not list its in-neighbours -- push only
    G = gt.Graph(); G.load_edges(e, nv, nv, True, False, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    P = gt.BFS_Program(G, False, False, True, gt._ROW_); P.root = 0; P.execute()
    og = O.OracleGraph(e, nv, directed=True, transpose=False, self_loops=False, acyclic=False, parallel_edges=False)
    parent, hops, it = og.bfs(0)
    assert (P.V["parent"] == parent).all() and (P.V["hops"] == hops).all() and P.iteration == it
    P.free(); G.free()


@pytest.mark.parametrize("first", [None, "0"])
def test_cc_first_iteration_is_the_first_neighbour(gt, O, first, known_answers, monkeypatch):
    """CC on a symmetric graph held by one rank: in iteration 0 every vertex sends its own id (cc.h:33-40), so the min-combiner
    leaves the smallest neighbour id = the first entry of the vertex's column; the engine reads that instead of sweeping every
    entry (GRAPHTAP_CC_FIRST=0: the sweep). Labels and iteration counts are the reference's either way, with self-loops kept or
    dropped; a DIRECTED graph under the CC program always sweeps."""
    from graphtap_amd.rmat import rmat_edges
    if first is not None: monkeypatch.setenv("GRAPHTAP_CC_FIRST", first)
    for name in CASES:
        c = load_case(name); nv = c["num_vertices"]; n = nv + 1
        r = run_min(gt, "cc", c["edges"], nv)
        assert (r["label"][:n] == c["np1_cc_a"]).all() and r["iterations"] == known_answers[name]["np1_cc"]["iterations"]
        if first is None: assert r["stats"].spmspv_iterations >= 1
    e = rmat_edges(16, 8, 21); nv = 1 << 16
    for self_loops in (True, False):
        G = gt.Graph(); G.load_edges(e, nv, nv, False, False, self_loops, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = gt.CC_Program(G, False, True, False, gt._ROW_); P.execute()
        og = O.OracleGraph(e, nv, directed=False, transpose=False, self_loops=self_loops, acyclic=False, parallel_edges=False)
        label, it = og.cc()
        assert (P.V["label"] == label).all() and P.iteration == it
        P.free(); G.free()
    G = gt.Graph(); G.load_edges(e, nv, nv, True, False, True, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    P = gt.CC_Program(G, False, True, False, gt._ROW_); P.execute()
    og = O.OracleGraph(e, nv, directed=True, transpose=False, self_loops=True, acyclic=False, parallel_edges=False)
    label, it = og.cc()
    assert (P.V["label"] == label).all() and P.iteration == it
    P.free(); G.free()


def test_display_lines_match_survey_table(gt):
    """First states printed by the reference on its bundled sample."""
    c = load_case("rmat10")
    r = run_min(gt, "bfs", c["edges"], 1024, 0)
    assert r["display"][1:6] == ["vertex[1]:Parent=317,Hops=2", "vertex[2]:Parent=151,Hops=2", "vertex[3]:Parent=345,Hops=2",
                                 "vertex[4]:Parent=151,Hops=2", "vertex[5]:Parent=866,Hops=3"]
    r = run_min(gt, "sssp", c["wedges"], 1024, 0)
    assert r["display"][1:6] == ["vertex[1]:Distance=43", "vertex[2]:Distance=INF", "vertex[3]:Distance=INF",
                                 "vertex[4]:Distance=51", "vertex[5]:Distance=INF"]
    G = gt.Graph(); G.load_edges(c["edges"], 1024, 1024, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
    VR = gt.PR_Program(G, True, False, False, gt._ROW_); VR.initialize(V); VR.execute(20)
    d = VR.display(out=None)
    assert d[0] == "vertex[0]:Rank=0.165455,Degree=10" and d[3] == "vertex[3]:Rank=0.151325,Degree=0" and d[5] == "vertex[5]:Rank=0.150000,Degree=0"
    assert len(d) == 31
    VR.free(); V.free(); G.free()


# ------------------------------------------------------------------------------- mid-size vs the oracle
@pytest.mark.parametrize("scale,seed", [(16, 1), (18, 2)])
def test_midsize_rmat_against_oracle(gt, O, scale, seed):
    from graphtap_amd.rmat import rmat_edges
    nv = 1 << scale
    w = rmat_edges(scale, 16, seed, weighted=True); e = np.ascontiguousarray(w[:, :2])
    ref = O.run_app("pr", e, nv, iters=20)
    r = run_pr(gt, e, nv, 20)
    assert (r["degree"] == ref["degree"]).all()
    assert (np.abs(r["rank"] - ref["rank"]) / ref["rank"]).max() < PR_RTOL
    root = int(np.bincount(e[:, 0]).argmax())
    ref = O.run_app("bfs", e, nv, root=root); r = run_min(gt, "bfs", e, nv, root)
    assert (r["parent"] == ref["parent"]).all() and (r["hops"] == ref["hops"]).all() and r["iterations"] == ref["iterations"]
    ref = O.run_app("cc", e, nv); r = run_min(gt, "cc", e, nv)
    assert (r["label"] == ref["label"]).all() and r["iterations"] == ref["iterations"]
    ref = O.run_app("sssp", w, nv, root=root); r = run_min(gt, "sssp", w, nv, root)
    assert (r["distance"] == ref["distance"]).all() and r["iterations"] == ref["iterations"]


# ------------------------------------------------------------------------------- multi-rank layout on one GPU
def _loopback_engines(progs):
    from graphtap_amd.vertex_program import _HipEngine
    for p in progs:
        if not p._already_initialized: p.initialize()
    return [_HipEngine(p) for p in progs]


def _loopback_lockstep(progs, engs, iters):
    """dist.run for the p engines living in this process: its collectives replaced by device copies"""
    nranks = len(engs)
    check = iters == 0
    while True:
        for e_ in engs: e_.scatter_gather()
        if not engs[0].column_accumulators and engs[0].iteration % 3 != 1:   # like dist.run; skipped now and then: optional
            for e_ in engs: e_.arm_fused_apply(iters, check)
        if engs[0].needs_x_exchange:   # the K all-to-alls of gt_graph_exchange_plan, by device copies
            xs = [e_.x_tensor() for e_ in engs]; sends = [e_.send_tensor() for e_ in engs]
            plans = [e_.exchange_plan() for e_ in engs]
            for k in range(engs[0].x_slices):
                for r in range(nranks):          # source
                    so, _, sc, _ = plans[r]
                    for d in range(nranks):      # destination
                        _, ro, _, rc = plans[d]
                        assert sc[k][d] == rc[k][r] and sc[k][d] % 4 == 0
                        a = so[k] + sum(sc[k][:d]); b = ro[k] + sum(rc[k][:r])
                        assert b + rc[k][r] <= ro[k + 1]
                        xs[d][b:b + rc[k][r]].copy_(sends[r][a:a + sc[k][d]])
        if engs[0].x_slices > 1 and (len(progs) + engs[0].iteration) % 2 == 0:   # exercise the sliced entry point too
            for k in range(engs[0].x_slices):
                for e_ in engs: e_.combine_slice(k)
        else:
            for e_ in engs: e_.combine()
        if engs[0].column_accumulators:
            tot = sum(e_.y_tensor().clone() for e_ in engs)
            for e_ in engs: e_.y_tensor().copy_(tot)
        act = sum(e_.apply(iters, check) for e_ in engs)
        if check:
            if act == 0:
                for e_ in engs: e_.finish_converged()
                break
        elif engs[0].iteration >= iters:
            break


def _loopback_gather(progs, field, n):
    """global array over original vertex ids from the p ranks' state slots (gt_graph_vertex_ids)"""
    out = np.zeros(n, progs[0].V[field].dtype); seen = np.zeros(n, bool)
    for p in progs:
        vids = p.G.vertex_ids(); keep = vids != 0xFFFFFFFF
        out[vids[keep]] = p.V[field][keep]; assert not seen[vids[keep]].any(); seen[vids[keep]] = True
    assert seen.all()
    return out


@pytest.mark.parametrize("variant", ["pb", "pb_f32msg", "edge"])
@pytest.mark.parametrize("nranks", [2, 3, 8])
@pytest.mark.parametrize("name", ["tiny", "rmat10", "rmat12"])
def test_tile_rows_of_p_ranks_reproduce_the_single_rank_run(gt, name, nranks, variant, monkeypatch):
    """Every rank's tile-row on the same GPU, with the all-to-alls of x done by device copies: checks
    the p-rank data layout (H = nrows/p + 1, needed-columns exchange plan, owned-segment state)
    of the engine without RCCL. BFS labels must equal the reference's np=1 run bit for bit;
    PageRank within 1e-6 (the reference's own np>1 runs differ from np=1 by fp association too)."""
    import torch
    from graphtap_amd.vertex_program import _HipEngine
    monkeypatch.setenv("GRAPHTAP_SPMV", variant)   # read by gt_graph_build
    monkeypatch.setenv("GRAPHTAP_X_SLICES", str({"pb": 4, "pb_f32msg": 2, "edge": 1}[variant]))   # K slices of the exchange
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1

    def run_all(make):
        progs = [make(r) for r in range(nranks)]
        return progs, _loopback_engines(progs)
    lockstep = _loopback_lockstep
    gather = lambda progs, field: _loopback_gather(progs, field, n)

    # BFS
    graphs = []
    def mk_bfs(r):
        G = gt.Graph(); G.load_edges(c["edges"], nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks)
        graphs.append(G)
        P = gt.BFS_Program(G, False, False, True, gt._ROW_); P.root = c["root"]; P.initialize(); return P
    progs, engs = run_all(mk_bfs)
    assert sum(G.info.nnz_local for G in graphs) == load_nnz(gt, c, "bfs")
    lockstep(progs, engs, 0)
    assert (gather(progs, "parent") == c["np1_bfs_a"]).all() and (gather(progs, "hops") == c["np1_bfs_b"]).all()
    for p in progs: p.free()
    for G in graphs: G.free()
    # Deg(_COL_) + PageRank, apps/pr.cpp
    graphs = []
    def mk_deg(r):
        G = gt.Graph(); G.load_edges(c["edges"], nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=r, nranks=nranks)
        graphs.append(G)
        P = gt.Deg_Program(G, True, False, False, gt._COL_); P.initialize(); return P
    degs, dengs = run_all(mk_deg)
    lockstep(degs, dengs, 1)
    prs = []
    for r in range(nranks):
        P = gt.PR_Program(graphs[r], True, False, False, gt._ROW_); P.initialize(degs[r]); prs.append(P)
    pengs = [_HipEngine(p) for p in prs]
    lockstep(prs, pengs, 20)
    assert (gather(prs, "degree") == c["np1_pr20_a"]).all()
    ref = c["np1_pr20_c"]
    assert (np.abs(gather(prs, "rank") - ref) / ref).max() < PR_RTOL
    for p in prs + degs: p.free()
    for G in graphs: G.free()
    torch.cuda.synchronize()


@pytest.mark.parametrize("nranks,slices", [(4, 1), (8, 4), (3, 64)])
def test_multirank_degenerate_graphs(gt, O, nranks, slices, monkeypatch):
    """Tile-rows without entries, segments without columns, an empty edge list: the exchange plan must stay consistent
    (all-zero blocks included) and the programs must give the oracle's answers."""
    monkeypatch.setenv("GRAPHTAP_X_SLICES", str(slices))
    nv = 40
    cases = [np.zeros((0, 2), np.uint32),
             np.array([[1, 2], [2, 3], [3, 1], [7, 7], [39, 0]], np.uint32),
             np.array([[5, 6]], np.uint32)]
    for e in cases:
        graphs = []
        for r in range(nranks):
            G = gt.Graph(); G.load_edges(e, nv, nv, False, False, True, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks)
            graphs.append(G)
        if len(e):
            ref = O.run_app("cc", e, nv); want_nnz, want_label, want_it = ref["graph"].nnz, ref["label"][:nv + 1], ref["iterations"]
        else:
            want_nnz, want_label, want_it = 0, np.arange(nv + 1), 1
        assert sum(G.info.nnz_local for G in graphs) == want_nnz
        for G in graphs:   # every local column maps to a distinct global slot of a column that exists
            t = G.tile_to_host(); i = G.info
            l2g = t["L2G"]; used = l2g != 0xFFFFFFFF
            assert np.unique(l2g[used]).size == used.sum() and (l2g[used] % i.seg_stride < i.seg_stride).all()
            assert (np.diff(t["JA"].astype(np.int64))[~used] == 0).all() and t["JA"][-1] == i.nnz_local
        progs = [gt.CC_Program(G, False, True, False, gt._ROW_) for G in graphs]
        engs = _loopback_engines(progs)
        _loopback_lockstep(progs, engs, 0)
        assert (_loopback_gather(progs, "label", nv + 1) == want_label).all() and progs[0].iteration == want_it
        for p in progs: p.free()
        for G in graphs: G.free()


def test_multirank_midsize_pagerank_and_bfs_equal_single_rank(gt, monkeypatch):
    """R-MAT 18 over 8 tile-rows (several phase-1 windows per exchange slice, split row bins) against the same graph
    on one rank: PageRank (f32 messages) to 1e-6, BFS bit for bit."""
    from graphtap_amd.rmat import rmat_edges
    monkeypatch.setenv("GRAPHTAP_SPMV", "pb_f32msg")
    monkeypatch.setenv("GRAPHTAP_X_SLICES", "4")
    scale, p = 18, 8
    nv = 1 << scale; n = nv + 1
    e = rmat_edges(scale, 16, 5)

    def pagerank(nranks):
        graphs = [gt.Graph() for _ in range(nranks)]
        for r, G in enumerate(graphs):
            G.load_edges(e, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=r, nranks=nranks)
        degs = [gt.Deg_Program(G, True, False, False, gt._COL_) for G in graphs]
        _loopback_lockstep(degs, _loopback_engines(degs), 1)
        prs = []
        for G, D in zip(graphs, degs):
            P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(D); prs.append(P)
        L = gt._lib.lib()
        gt._lib.check(L.gt_program_enable_timing(prs[-1]._handle(), 1))
        _loopback_lockstep(prs, _loopback_engines(prs), 10)
        ms, launches = C.c_double(), C.c_uint32()
        gt._lib.check(L.gt_program_timing(prs[-1]._handle(), C.byref(ms), C.byref(launches), 1))
        assert launches.value == 10 and ms.value > 0      # one timed SpMV per iteration, sliced or not
        out = _loopback_gather(prs, "rank", n), _loopback_gather(prs, "degree", n)
        for P in prs + degs: P.free()
        for G in graphs: G.free()
        return out

    def bfs(nranks):
        graphs = [gt.Graph() for _ in range(nranks)]
        for r, G in enumerate(graphs):
            G.load_edges(e, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks)
        progs = []
        for G in graphs:
            P = gt.BFS_Program(G, False, False, True, gt._ROW_); P.root = 3; progs.append(P)
        _loopback_lockstep(progs, _loopback_engines(progs), 0)
        out = _loopback_gather(progs, "parent", n), _loopback_gather(progs, "hops", n), progs[0].iteration
        for P in progs: P.free()
        for G in graphs: G.free()
        return out

    r1, d1 = pagerank(1); r8, d8 = pagerank(p)
    assert (d1 == d8).all() and (np.abs(r8 - r1) / r1).max() < PR_RTOL
    a, b = bfs(1), bfs(p)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and a[2] == b[2]


def load_nnz(gt, c, app):
    from oracle import oracle as O
    return O.OracleGraph(c["edges"], c["num_vertices"], **O.APP_FLAGS[app]).nnz


# ------------------------------------------------------------------------------- full-size properties
def test_pagerank_rmat22_properties(gt):
    """BASELINE config 2 size (RMAT-22, 67 M edges, 20 iterations): too large for the oracle in a test,
    so check size-independent properties: (i) the SpMV conserves mass, sum(y) = sum_j x[j] * colcount[j];
    (ii) ranks of rows without in-edges stay alpha; (iii) determinism of the degree pass;
    (iv) all ranks >= alpha and finite."""
    import torch
    L = gt._lib.lib()
    scale, nv = 22, 1 << 22
    m = 16 << scale
    d = C.c_void_p(); gt._lib.check(L.gt_malloc(C.byref(d), m * 8))
    gt._lib.check(L.gt_rmat_generate(d, scale, 1, 0, 0, m, None))
    G = gt.Graph(); G.load_device(d.value, m, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
    gt._lib.check(L.gt_free(d))
    assert G.info.nnz_local == m
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
    VR = gt.PR_Program(G, True, False, False, gt._ROW_); VR.initialize(V)
    deg = V.V["degree"]
    assert int(deg.astype(np.int64).sum()) == m
    VR.execute(20)
    st = VR.V
    rank, degree = st["rank"], st["degree"]
    assert np.isfinite(rank).all() and rank.min() >= 0.15 - 1e-15
    t = G.tile_to_host()
    has_row = np.zeros(G.info.tile_height, bool); has_row[t["IR"]] = True
    assert (rank[~has_row] == 0.15).all() and (degree[~has_row] == 0).all()
    # mass conservation of one more SpMV, checked in fp64 on the host
    x = np.zeros(G.info.nnzcols); v = t["JC"]; nz = degree[v] > 0
    x[nz] = rank[v][nz] / degree[v][nz]
    colcount = np.diff(t["JA"][:G.info.nnzcols + 1].astype(np.int64))
    dx = C.c_void_p(); dy = C.c_void_p()
    gt._lib.check(L.gt_malloc(C.byref(dx), x.nbytes)); gt._lib.check(L.gt_malloc(C.byref(dy), G.info.nnzrows * 8))
    gt._lib.check(L.gt_memcpy_h2d(dx, x.ctypes.data_as(C.c_void_p), x.nbytes)); gt._lib.check(L.gt_memset(dy, 0, G.info.nnzrows * 8))
    gt._lib.check(L.gt_spmv(G._h, gt._lib.GT_PLUS_F64, dx, dy, None))
    y = np.zeros(G.info.nnzrows); gt._lib.check(L.gt_memcpy_d2h(y.ctypes.data_as(C.c_void_p), dy, y.nbytes))
    assert abs(y.sum() - (x * colcount).sum()) <= 1e-9 * y.sum()
    # the applied ranks are alpha + 0.85 * y of the previous iteration's x, so this y reproduces iteration 21
    L.gt_free(dx); L.gt_free(dy)
    VR.free(); V.free(); G.free()


# ------------------------------------------------------------------------------- SpMV variants
@pytest.mark.parametrize("scale,nranks,rank", [(12, 1, 0), (17, 1, 0), (20, 1, 0), (17, 3, 1), (20, 8, 5)])
def test_propagation_blocking_equals_edge_kernel(gt, scale, nranks, rank):
    """The production kernel pair (pb.hip) against the edge-parallel baseline on the same tile-row, for
    every semiring: integer semirings bit-exact, f64 sums to 1e-12 (both re-associate). Covers several
    row bins, split bins (atomic merge), partial windows and the local (needed-columns) column space
    of a multi-rank tile-row."""
    from graphtap_amd.rmat import rmat_edges
    L = gt._lib.lib()
    nv = 1 << scale
    w = rmat_edges(scale, 16, seed=4, weighted=True)
    rng = np.random.RandomState(scale)

    def dev(a):
        d = C.c_void_p(); gt._lib.check(L.gt_malloc(C.byref(d), max(a.nbytes, 1)))
        gt._lib.check(L.gt_memcpy_h2d(d, a.ctypes.data_as(C.c_void_p), a.nbytes)); return d

    def run(G, sr, x, y0):
        outs = []
        for variant in (gt._lib.GT_SPMV_EDGE, gt._lib.GT_SPMV_PB):
            gt._lib.check(L.gt_graph_select_spmv(G._h, variant))
            dx, dy = dev(x), dev(y0)
            gt._lib.check(L.gt_spmv(G._h, sr, dx, dy, None))
            o = np.zeros_like(y0); gt._lib.check(L.gt_memcpy_d2h(o.ctypes.data_as(C.c_void_p), dy, o.nbytes))
            L.gt_free(dx); L.gt_free(dy); outs.append(o)
        return outs

    G = gt.Graph(weighted=True)
    G.load_edges(w, nv, nv, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=rank, nranks=nranks)
    nx, ny = G.info.ncols_local, G.info.nnzrows
    x = rng.rand(nx); y0 = rng.rand(ny)
    a, b = run(G, gt._lib.GT_PLUS_F64, x, y0)
    assert np.allclose(a, b, rtol=1e-12, atol=0)
    xi = rng.randint(0, 1 << 20, nx).astype(np.uint32); xi[rng.rand(nx) < 0.3] = gt.INF
    yi = rng.randint(0, 1 << 21, ny).astype(np.uint32); yi[rng.rand(ny) < 0.5] = gt.INF
    for sr in (gt._lib.GT_MIN_U32, gt._lib.GT_MINPLUS_U32):
        a, b = run(G, sr, xi, yi)
        assert (a == b).all()
    a, b = run(G, gt._lib.GT_PLUS_U32, (xi & 0xFF).astype(np.uint32), (yi & 0xFFFF).astype(np.uint32))
    assert (a == b).all()
    G.free()


@pytest.mark.parametrize("scale,seed", [(12, 2), (16, 1), (18, 2), (20, 3)])
def test_pagerank_f32_message_variant_within_tolerance(gt, O, scale, seed, monkeypatch):
    """GT_SPMV_PB_F32MSG rounds the PageRank messages x = rank/degree to f32 in flight (sums, y and ranks
    stay f64). The north-star tolerance for PageRank is 1e-6 relative against the fp64 reference; this pins
    the variant to it on the same seeded inputs as the f64 path, 20 iterations."""
    from graphtap_amd.rmat import rmat_edges
    nv = 1 << scale
    e = rmat_edges(scale, 16, seed)
    ref = O.run_app("pr", e, nv, iters=20)
    monkeypatch.setenv("GRAPHTAP_SPMV", "pb_f32msg")
    r = run_pr(gt, e, nv, 20)
    monkeypatch.delenv("GRAPHTAP_SPMV")
    assert (r["degree"] == ref["degree"]).all()
    rel = np.abs(r["rank"] - ref["rank"]) / ref["rank"]
    print("f32-message PageRank scale %d: max rel err %.3g" % (scale, rel.max()))
    assert rel.max() < PR_RTOL


# ------------------------------------------------------------------------------- randomized graphs
@pytest.mark.parametrize("seed", range(12))
def test_random_graphs_all_programs_against_oracle(gt, O, seed):
    """Small random graphs of varied shape (isolated ids, dense hubs, duplicates, self loops, a path, ids up to
    N itself), every program end to end against the oracle; sizes cross the PB structure boundaries (more than
    one 8192-column window, more than one 16384-row bin on the larger draws)."""
    rng = np.random.RandomState(1000 + seed)
    nv = int(rng.choice([7, 64, 1000, 20000, 40000]))
    m = int(rng.choice([1, 5, 300, 5000, 200000]))
    kind = seed % 4
    if kind == 0:      # uniform
        e = rng.randint(0, nv + 1, size=(m, 2))
    elif kind == 1:    # hubs: a few vertices take most endpoints
        hubs = rng.randint(0, nv + 1, size=4)
        e = rng.randint(0, nv + 1, size=(m, 2))
        mask = rng.rand(m) < 0.6
        e[mask, rng.randint(0, 2)] = hubs[rng.randint(0, 4, size=mask.sum())]
    elif kind == 2:    # many duplicates and self loops
        e = rng.randint(0, min(nv, 30) + 1, size=(m, 2))
    else:              # a long path plus noise
        k = min(m, nv)
        path = np.stack([np.arange(k), np.arange(1, k + 1)], axis=1)
        e = np.concatenate([path, rng.randint(0, nv + 1, size=(max(m - k, 1), 2))])
    e = np.ascontiguousarray(e, dtype=np.uint32)
    w = np.concatenate([e, rng.randint(1, 129, size=(len(e), 1)).astype(np.uint32)], axis=1)
    root = int(e[0, 0])
    ref = O.run_app("pr", e, nv, iters=5); r = run_pr(gt, e, nv, 5)
    assert (r["degree"] == ref["degree"]).all() and np.allclose(r["rank"], ref["rank"], rtol=PR_RTOL, atol=0)
    ref = O.run_app("pr", e, nv, iters=0, cf=False); r = run_pr(gt, e, nv, 0, cf=False)
    assert r["iterations"] == ref["iterations"] and np.allclose(r["rank"], ref["rank"], rtol=PR_RTOL, atol=0)
    ref = O.run_app("bfs", e, nv, root=root); r = run_min(gt, "bfs", e, nv, root)
    assert (r["parent"] == ref["parent"]).all() and (r["hops"] == ref["hops"]).all() and r["iterations"] == ref["iterations"]
    ref = O.run_app("cc", e, nv); r = run_min(gt, "cc", e, nv)
    assert (r["label"] == ref["label"]).all() and r["iterations"] == ref["iterations"]
    ref = O.run_app("sssp", w, nv, root=root); r = run_min(gt, "sssp", w, nv, root)
    assert (r["distance"] == ref["distance"]).all() and r["iterations"] == ref["iterations"]
    G = gt.Graph(); G.load_edges(e, nv, nv, True, False, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    V = gt.Deg_Program(G, True, False, False, gt._ROW_); V.execute(1)
    assert (V.V["degree"] == O.run_app("deg", e, nv)["degree"]).all()
    V.free(); G.free()


@pytest.mark.parametrize("shape", ["star", "block", "hub_rows_hub_cols"])
def test_adversarial_shapes_against_oracle(gt, O, shape):
    """Shapes that stress the propagation-blocking build: one (window, row bin) cell far above the chunk size (cut by
    column residue), same-row stretches of hundreds of entries (lane-pair aggregation, long runs), hub windows cut by
    row bin. PageRank to 1e-6, BFS / CC / SSSP bit for bit."""
    rng = np.random.RandomState(7)
    if shape == "star":            # vertex 0 <-> everybody, twice the ids of one window
        nv = 40000
        k = np.arange(1, nv, dtype=np.uint32)
        e = np.concatenate([np.stack([np.zeros_like(k), k], 1), np.stack([k, np.zeros_like(k)], 1)])
    elif shape == "block":         # a dense 600 x 600 block: 360 000 entries in one (window, bin) cell
        nv = 30000
        a, b = np.meshgrid(np.arange(600, dtype=np.uint32), np.arange(600, dtype=np.uint32) + 100)
        e = np.stack([a.ravel(), b.ravel()], 1)
        e = np.concatenate([e, rng.randint(0, nv, size=(20000, 2)).astype(np.uint32)])
    else:                          # 40 hub rows x 40 hub columns carry most of 400 000 entries
        nv = 60000
        hubs = rng.randint(0, nv, size=40).astype(np.uint32)
        e = rng.randint(0, nv, size=(400000, 2)).astype(np.uint32)
        m1, m2 = rng.rand(len(e)) < 0.5, rng.rand(len(e)) < 0.5
        e[m1, 0] = hubs[rng.randint(0, 40, size=m1.sum())]; e[m2, 1] = hubs[rng.randint(0, 40, size=m2.sum())]
    e = np.ascontiguousarray(e, dtype=np.uint32)
    w = np.concatenate([e, rng.randint(1, 129, size=(len(e), 1)).astype(np.uint32)], axis=1)
    root = int(e[0, 0])
    ref = O.run_app("pr", e, nv, iters=6); r = run_pr(gt, e, nv, 6)
    assert (r["degree"] == ref["degree"]).all() and np.allclose(r["rank"], ref["rank"], rtol=PR_RTOL, atol=0)
    ref = O.run_app("bfs", e, nv, root=root); r = run_min(gt, "bfs", e, nv, root)
    assert (r["parent"] == ref["parent"]).all() and (r["hops"] == ref["hops"]).all() and r["iterations"] == ref["iterations"]
    ref = O.run_app("cc", e, nv); r = run_min(gt, "cc", e, nv)
    assert (r["label"] == ref["label"]).all() and r["iterations"] == ref["iterations"]
    ref = O.run_app("sssp", w, nv, root=root); r = run_min(gt, "sssp", w, nv, root)
    assert (r["distance"] == ref["distance"]).all() and r["iterations"] == ref["iterations"]


@pytest.mark.parametrize("wmax", [200, 300, 70000])
def test_sssp_weight_widths(gt, O, wmax):
    """The weight stream of min-plus travels as u8, u16 or u32 depending on the largest weight: all three against the oracle"""
    from graphtap_amd.rmat import rmat_edges
    rng = np.random.RandomState(wmax)
    e = rmat_edges(15, 8, 11)
    w = np.concatenate([e, rng.randint(1, wmax + 1, size=(len(e), 1)).astype(np.uint32)], axis=1)
    w[0, 2] = wmax                                    # the maximum is really there
    ref = O.run_app("sssp", w, 1 << 15, root=1); r = run_min(gt, "sssp", w, 1 << 15, 1)
    assert (r["distance"] == ref["distance"]).all() and r["iterations"] == ref["iterations"]


def test_bfs_cc_sssp_properties_at_rmat20(gt):
    """Size-independent properties (R-MAT 20 symmetrised: ~32 M stored entries), checked on
    the host against the edge list itself: BFS parent/hops consistency (parent is a neighbour one level up, the
    minimum such id), CC labels constant along every edge and equal to a member id, SSSP distances satisfy the
    triangle inequality on every edge with equality on at least one incoming edge of every reached vertex."""
    from graphtap_amd.rmat import rmat_edges
    scale, nv = 20, 1 << 20   # host-side checks stay in seconds at scale 20; the GPU path is the same code as at 22/26
    w = rmat_edges(scale, 16, 7, weighted=True); e = np.ascontiguousarray(w[:, :2])
    root = 0
    r = run_min(gt, "bfs", e, nv, root)
    hops, parent = r["hops"].astype(np.int64), r["parent"].astype(np.int64)
    INF = gt.INF
    a = np.concatenate([e[:, 0], e[:, 1]]).astype(np.int64); b = np.concatenate([e[:, 1], e[:, 0]]).astype(np.int64)
    keep = a != b; a, b = a[keep], b[keep]                      # symmetrised, no self loops (bfs.cpp:26-30)
    reached = hops != INF
    assert hops[root] == 0 and parent[root] == root
    assert (np.abs(hops[a] - hops[b])[reached[a] & reached[b]] <= 1).all() and (reached[a] == reached[b]).all()
    # parent[v] = min id among neighbours at level hops[v]-1
    cand = np.full(nv + 2, INF, np.int64)
    up = reached[b] & reached[a] & (hops[a] + 1 == hops[b])
    np.minimum.at(cand, b[up], a[up])
    chk = reached & (np.arange(nv + 2) != root)
    assert (parent[chk] == cand[chk]).all()
    r = run_min(gt, "cc", e, nv)
    lab = r["label"].astype(np.int64)
    assert (lab[a] == lab[b]).all() and (lab[lab[:nv + 1]] == lab[:nv + 1]).all() and (lab[:nv + 1] <= np.arange(nv + 1)).all()
    r = run_min(gt, "sssp", w, nv, root)
    d = r["distance"].astype(np.int64)
    src, dst, ww = w[:, 0].astype(np.int64), w[:, 1].astype(np.int64), w[:, 2].astype(np.int64)
    ok = (src != dst) & (d[src] != INF)
    assert (d[dst][ok] <= d[src][ok] + ww[ok]).all()
    best = np.full(nv + 2, INF, np.int64); np.minimum.at(best, dst[ok], d[src][ok] + ww[ok])
    reach = (d != INF) & (np.arange(nv + 2) != root)
    assert (d[reach] == best[reach]).all() and d[root] == 0


@pytest.mark.parametrize("name", CASES)
def test_pr1_flow_two_graphs(gt, name, known_answers):
    """apps/pr1.cpp: Degree in _ROW_ order on the UNtransposed graph, then PageRank on a second, transposed
    graph initialised from it (plain TCSC) -- initialize(other) across two graph objects."""
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1
    G = gt.Graph(); G.load_edges(c["edges"], nv, nv, True, False, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    V = gt.Deg_Program(G, True, False, False, gt._ROW_); V.execute(1)
    GR = gt.Graph(); GR.load_edges(c["edges"], nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    VR = gt.PR_Program(GR, True, False, False, gt._ROW_); VR.initialize(V)
    VR.execute(20)
    st = VR.V
    assert (st["degree"][:n] == c["np1_pr1app20_a"]).all()
    ref = c["np1_pr1app20_c"]
    assert (np.abs(st["rank"][:n] - ref) / ref).max() < PR_RTOL
    ka = known_answers[name]["np1_pr1app20"]
    assert VR.checksum(out=None)[1] == ka["reachable"]
    VR.free(); V.free(); GR.free(); G.free()


def test_api_misuse_is_reported_not_fatal(gt):
    L = gt._lib.lib()
    c = load_case("rmat8")
    # SSSP on an unweighted graph (the reference would need a different binary: -DHAS_WEIGHT)
    G = gt.Graph(); G.load_edges(c["edges"], 256, 256, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    P = gt.SSSP_Program(G, False, True, False, gt._ROW_)
    with pytest.raises(gt.GraphTapError, match="weighted"):
        P.execute()
    # _COL_ ordering exists for Degree only (the reference: "Not implemented", vp:1319-1322)
    P2 = gt.PR_Program(G, True, False, False, gt._COL_)
    with pytest.raises(gt.GraphTapError, match="_COL_"):
        P2.execute(1)
    # stationary flag must match the program (apps/*.cpp)
    with pytest.raises(gt.GraphTapError, match="stationary"):
        gt.BFS_Program(G, True, False, True, gt._ROW_)
    # switching a graph away from the f32-message variant under a live PageRank program is refused, not miscomputed
    gt._lib.check(L.gt_graph_select_spmv(G._h, gt._lib.GT_SPMV_PB_F32MSG))
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
    VR = gt.PR_Program(G, True, False, False, gt._ROW_); VR.initialize(V)
    gt._lib.check(L.gt_graph_select_spmv(G._h, gt._lib.GT_SPMV_EDGE))
    with pytest.raises(gt.GraphTapError, match="f32 message"):
        VR.execute(2)
    assert L.gt_graph_select_spmv(G._h, 7) != 0 and b"variant" in L.gt_last_error()
    VR.free(); V.free(); G.free()
    # execute() to convergence followed by execute(N) terminates: the reference's check_for_convergence flag is sticky
    # (vp:412-413) and every trip of its loop counts (vp:421), so the second call makes exactly one more (no-op) trip
    G = gt.Graph(); G.load_edges(c["edges"], 256, 256, False, False, True, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    P = gt.CC_Program(G, False, True, False, gt._ROW_); P.execute()
    it, lab = P.iteration, P.V["label"].copy()
    P.execute(it + 50)
    assert P.iteration == it + 1 and (P.V["label"] == lab).all()
    P.free()
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
    VR = gt.PR_Program(G, True, False, False, gt._ROW_); VR.initialize(V); VR.execute()
    it, rk = VR.iteration, VR.V["rank"].copy()
    VR.execute(it + 7)
    assert VR.iteration == it + 1 and (VR.V["rank"] == rk).all()
    VR.free(); V.free(); G.free()


def test_every_wait_of_execute_has_a_deadline(gt, monkeypatch):
    """gt_program_execute never spins for ever: with a deadline far shorter than the run (GRAPHTAP_TIMEOUT_S, read at every
    wait) the call returns GT_ERR_TIMEOUT with a message -- never a hang with a core at 100 %, never a re-exec -- for the
    fixed-count loop (the wait at the end of execute()) and for converge mode (the per-iteration read-back, gt_read_back).
    The device is drained afterwards and the same handles run again to the right answer with the default deadline."""
    from graphtap_amd.rmat import rmat_edges
    L = gt._lib.lib()
    scale, nv = 20, 1 << 20
    e = rmat_edges(scale, 16, 3)
    G = gt.Graph(); G.load_edges(e, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
    P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V)
    monkeypatch.setenv("GRAPHTAP_TIMEOUT_S", "1e-7")
    with pytest.raises(gt.GraphTapError, match="did not complete within"):
        P.execute(400)                       # ~20 ms of queued kernels against a 0.1-us deadline
    gt._lib.check(L.gt_device_synchronize())
    with pytest.raises(gt.GraphTapError, match="did not complete within"):
        P.initialize(V); P.execute()         # converge mode: the first read-back of the active count
    gt._lib.check(L.gt_device_synchronize())
    monkeypatch.delenv("GRAPHTAP_TIMEOUT_S")
    P.initialize(V); P.execute(20)
    got = P.V["rank"].copy()
    P2 = gt.PR_Program(G, True, False, False, gt._ROW_); P2.initialize(V); P2.execute(20)
    assert (np.abs(got - P2.V["rank"]) / P2.V["rank"]).max() < 1e-12
    P2.free(); P.free(); V.free(); G.free()


def test_handle_level_options_two_graphs_in_one_process_differ(gt, O, monkeypatch):
    """gt_graph_options / gt_program_options (ABI 3): what the GRAPHTAP_* environment variables choose, per handle. Two graphs of
    one process are built with different SpMV variants, hub thresholds and chunk sizes -- no environment variable is touched -- and
    both give the oracle's PageRank; a program-level option wins over the environment (SpMSpV forced / forbidden per program,
    seen in its statistics); a per-program deadline makes ONE program time out while another of the same graph runs on."""
    from graphtap_amd.rmat import rmat_edges
    for k in ("GRAPHTAP_SPMV", "GRAPHTAP_PB_HUB_DEG", "GRAPHTAP_PB_CH", "GRAPHTAP_SPMSPV", "GRAPHTAP_TIMEOUT_S"): monkeypatch.delenv(k, raising=False)
    scale, nv = 16, 1 << 16
    w = rmat_edges(scale, 16, 4, weighted=True); e = np.ascontiguousarray(w[:, :2])
    ref = O.run_app("pr", e, nv, iters=20)
    def width(P):   # bytes of a message of this program: 4 under GT_SPMV_PB_F32MSG (fixed-count runs), else 8
        wd = C.c_uint32(); gt._lib.check(gt._lib.lib().gt_program_x(P._handle(), None, None, C.byref(wd))); return wd.value
    widths, wide = [], []
    for opts in (gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_PB_F32MSG, hub_min_degree=64, chunk_log2=14),
                 gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_PB_F32MSG, wide_windows=1),
                 gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_EDGE), gt.GraphOptions(hubs_first=0), None):
        G = gt.Graph(options=opts); G.load_edges(e, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
        V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
        P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V); P.execute(20)
        assert (np.abs(P.V["rank"] - ref["rank"]) / ref["rank"]).max() < PR_RTOL
        widths.append(width(P)); wide.append(gt._lib.lib().gt_graph_has_wide_build(G._h))
        P.free(); V.free(); G.free()
    assert widths == [4, 4, 8, 8, 8], widths       # the variants really differed inside one process
    assert wide == [0, 1, 0, 0, 0], wide           # ... and so did the builds: the wide windows for the one graph that asked for them
    monkeypatch.setenv("GRAPHTAP_SPMV", "pb_f32msg")    # ... and an option wins over the environment
    G = gt.Graph(options=gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_PB)); G.load_edges(e, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
    P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V); P.execute(20)
    assert width(P) == 8
    # a per-program deadline: this program gives up, its sibling on the same graph does not
    P2 = gt.PR_Program(G, True, False, False, gt._ROW_); P2.initialize(V)
    P2.set_options(timeout_s=1e-7)
    with pytest.raises(gt.GraphTapError, match="did not complete within"):
        P2.execute(2000)
    gt._lib.check(gt._lib.lib().gt_device_synchronize())
    P.initialize(V); P.execute(20)
    assert (np.abs(P.V["rank"] - ref["rank"]) / ref["rank"]).max() < PR_RTOL
    P2.free(); P.free(); V.free(); G.free()
    monkeypatch.delenv("GRAPHTAP_SPMV")
    # program options of a min program: the sparse path forced / forbidden per program, same labels
    want = O.run_app("sssp", w, nv, root=3)
    G = gt.Graph(weighted=True); G.load_edges(w, nv, nv, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
    seen = {}
    for name, kw in (("never", dict(spmspv=0, tail_kernel=0)), ("always", dict(spmspv=1, tail_kernel=0)), ("hybrid", dict(hybrid=1, spmspv=0, tail_kernel=0))):
        S = gt.SSSP_Program(G, False, True, False, gt._ROW_); S.root = 3; S.set_options(**kw); S.execute()
        assert (S.V["distance"] == want["distance"]).all() and S.iteration == want["iterations"], name
        seen[name] = S.stats.spmspv_iterations
        S.free()
    assert seen["never"] == 0 and seen["always"] > 0, seen
    G.free(); ref["graph"].close(); want["graph"].close()


def test_groups_whose_256_entries_all_end_a_stretch(gt, O):
    """A permutation-like matrix with every column declared a hub (gt_graph_options.hub_min_degree = 1): every entry of a dense window
    is the only one of its row there, so every 256-entry group of phase 1 has 256 outputs -- the staging row of a wave full to its
    last entry, every lane storing four outputs, run constants changing inside groups. (Written for a trip counter that lived in that
    last entry, round 4 -- measured slower than the rotating priority and never merged: DESIGN.md 4.1 item 5; the case stays.) PageRank, f64 and f32 messages
    on the narrow and the wide build, against the oracle; plus a second graph with two entries per row and window, where stretches
    of two alternate with the full groups."""
    rng = np.random.default_rng(11)
    nv = 1 << 17
    for per_row in (1, 2):
        src = np.concatenate([rng.permutation(nv) for _ in range(per_row)]).astype(np.uint32)
        dst = np.tile(np.arange(nv, dtype=np.uint32), per_row)
        if per_row == 2: src[nv:] = (src[:nv] + 1) % nv          # the second entry of a row sits next to its first: stretches of two
        e = np.ascontiguousarray(np.stack([src, dst], 1))
        ref = O.run_app("pr", e, nv, iters=8)
        for opts in (gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_PB, hub_min_degree=1), gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_PB_F32MSG, hub_min_degree=1, wide_windows=0),
                     gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_PB_F32MSG, hub_min_degree=1, wide_windows=1)):
            G = gt.Graph(options=opts); G.load_edges(e, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
            V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
            P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V); P.execute(8)
            assert (np.abs(P.V["rank"] - ref["rank"]) / ref["rank"]).max() < PR_RTOL, (per_row, opts.spmv_variant, opts.wide_windows)
            assert (P.V["degree"] == ref["degree"]).all()
            P.free(); V.free(); G.free()
        ref["graph"].close()


def test_persistent_phase1_equals_the_dispatched_form(gt, O, monkeypatch):
    """Phase 1 as persistent workgroups drawing chunks from a counter (pb.hip, k_pb_scatter `queue`; round 4) against one workgroup per
    chunk: the same chunks, the same arithmetic inside each -- the value stream is the same bit for bit; the ranks agree to 1e-12
    relative (phase 2 adds the partial sums with LDS atomics in f64, whose order no two runs share: 1e-15 per sum), with f64
    messages (the narrow build's f64 kernel) and with f32 messages on the wide build; each against the oracle as well.
    GRAPHTAP_PB_PERSIST is read per launch, and the graph counts its persistent launches, so the test knows that both forms ran."""
    from graphtap_amd.rmat import rmat_edges
    L = gt._lib.lib(); L.gt_graph_persistent_launches.restype = C.c_uint64; L.gt_graph_persistent_launches.argtypes = [C.c_void_p]
    scale, nv = 20, 1 << 20
    e = rmat_edges(scale, 16, 9)
    ref = O.run_app("pr", e, nv, iters=10)
    for opts in (gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_PB), gt.GraphOptions(spmv_variant=gt._lib.GT_SPMV_PB_F32MSG, wide_windows=1)):
        ranks = {}
        for mode in ("0", None):
            if mode is None: monkeypatch.delenv("GRAPHTAP_PB_PERSIST", raising=False)
            else: monkeypatch.setenv("GRAPHTAP_PB_PERSIST", mode)
            G = gt.Graph(options=opts); G.load_edges(e, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
            V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
            P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V); P.execute(10)
            ranks[mode] = P.V["rank"].copy()
            n = L.gt_graph_persistent_launches(G._h)
            assert (n == 0) if mode == "0" else (n >= 10), (mode, n)
            assert (np.abs(ranks[mode] - ref["rank"]) / ref["rank"]).max() < PR_RTOL
            P.free(); V.free(); G.free()
        assert (np.abs(ranks["0"] - ranks[None]) / ranks[None]).max() < 1e-12
    ref["graph"].close()


@pytest.mark.parametrize("hub_deg", [None, "2", "1000000"])
def test_wide_build_gives_the_same_results(gt, O, hub_deg, known_answers, monkeypatch):
    """The WIDE propagation-blocking build (pb.hip, gt_pb::wide; round 4): windows of 32 766 / 32 768 slots -- pairs of the layout's
    windows -- 15 column bits, the run heads as a mask in the group record. By default only graphs of ~0.47 G entries and more get it
    (the full-size headline tests run on it); here GRAPHTAP_PB_WIDE=1 builds it for small and mid-size graphs: PageRank with f32
    messages against the reference's vectors and the oracle (fixed count, hand-stepped without the fused applicator, and converge
    mode -- which runs f64 messages on the NARROW build of the same graph), and -- the same switch sends the min programs
    through it too -- BFS / SSSP / CC bit for bit incl. iteration counts. Hub thresholds 2 / none: many / no dense windows, an odd
    number of them included; R-MAT-16 and 18 have separate chunks for source rows whose short runs put run heads in every lane
    (a first version sign-extended the low word of the head mask: a head in lane 31 became heads in lanes 32-63)."""
    from graphtap_amd.rmat import rmat_edges
    L = gt._lib.lib()
    monkeypatch.setenv("GRAPHTAP_PB_WIDE", "1"); monkeypatch.setenv("GRAPHTAP_SPMV", "pb_f32msg")
    if hub_deg: monkeypatch.setenv("GRAPHTAP_PB_HUB_DEG", hub_deg)
    for name in CASES:
        c = load_case(name); nv = c["num_vertices"]; n = nv + 1; k = known_answers[name]
        r = run_pr(gt, c["edges"], nv, 20)
        assert (np.abs(r["rank"][:n] - c["np1_pr20_c"]) / c["np1_pr20_c"]).max() < PR_RTOL and (r["degree"][:n] == c["np1_pr20_a"]).all()
        r = run_pr(gt, c["edges"], nv, 0)
        assert r["iterations"] == k["np1_prconv_cf"]["iterations"] and (np.abs(r["rank"][:n] - c["np1_prconv_cf_c"]) / c["np1_prconv_cf_c"]).max() < PR_RTOL
        r = run_min(gt, "bfs", c["edges"], nv, c["root"])
        assert (r["parent"][:n] == c["np1_bfs_a"]).all() and r["iterations"] == k["np1_bfs"]["iterations"]
        r = run_min(gt, "sssp", c["wedges"], nv, c["root"])
        assert (r["distance"][:n] == c["np1_sssp_a"]).all() and r["iterations"] == k["np1_sssp"]["iterations"]
        r = run_min(gt, "cc", c["edges"], nv)
        assert (r["label"][:n] == c["np1_cc_a"]).all() and r["iterations"] == k["np1_cc"]["iterations"]
    for scale, seed in ((16, 5), (18, 5)):
        nv = 1 << scale; w = rmat_edges(scale, 16, seed, weighted=True); e = np.ascontiguousarray(w[:, :2])
        ref = O.run_app("pr", e, nv, iters=10)
        G = gt.Graph(); G.load_edges(e, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
        V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
        P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V); P.execute(10)
        assert (np.abs(P.V["rank"] - ref["rank"]) / ref["rank"]).max() < PR_RTOL
        P.initialize(V); h = P._handle()
        for it in range(10):   # stepped by hand, the applicator fused in two iterations of three
            gt._lib.check(L.gt_program_scatter_gather(h))
            if it % 3 != 1: gt._lib.check(L.gt_program_fuse_apply(h, 10, 0))
            gt._lib.check(L.gt_program_combine(h)); gt._lib.check(L.gt_program_apply(h, 10, None))
        assert (np.abs(P.V["rank"] - ref["rank"]) / ref["rank"]).max() < PR_RTOL
        P.free(); V.free(); G.free(); ref["graph"].close()
        if scale == 16:
            want = O.run_app("sssp", w, nv, root=0); got = run_min(gt, "sssp", w, nv, 0)
            assert (got["distance"] == want["distance"]).all() and got["iterations"] == want["iterations"]
            want["graph"].close()
            want = O.run_app("cc", e, nv); got = run_min(gt, "cc", e, nv)
            assert (got["label"] == want["label"]).all() and got["iterations"] == want["iterations"]
            want["graph"].close()


def test_multirank_tile_rows_are_balanced(gt):
    """Contiguous id ranges of R-MAT are badly skewed (tile-row 0 of 8 would hold ~44 % of the entries); the hashed
    internal id space must give every rank a similar share of entries, rows and columns."""
    from graphtap_amd.rmat import rmat_edges
    scale, p = 18, 8
    e = rmat_edges(scale, 16, 1)
    nnz, rows, cols = [], [], []
    for r in range(p):
        G = gt.Graph(); G.load_edges(e, 1 << scale, 1 << scale, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=r, nranks=p)
        nnz.append(G.info.nnz_local); rows.append(G.info.nnzrows); cols.append(G.info.nnzcols)
        vids = G.vertex_ids()
        if r == 0:
            allv = []
        allv.append(vids[vids != 0xFFFFFFFF])
        G.free()
    assert sum(nnz) == len(e)
    assert max(nnz) < 1.25 * (sum(nnz) / p), nnz
    assert max(rows) < 1.1 * (sum(rows) / p) and max(cols) < 1.1 * (sum(cols) / p)
    allv = np.concatenate(allv)
    assert allv.size == (1 << scale) + 1 and np.unique(allv).size == allv.size   # every vertex owned exactly once


def test_no_device_allocation_inside_the_iteration_loop(gt):
    """initialize() reserves everything execute() needs -- the value stream of the SpMV (gt_pb_reserve_val; for PageRank at both
    message widths), the frontier buffers, the timing events: gt_exec_stats.allocs_in_execute stays 0 for all five programs,
    first execute included (round 2's first-process stall was the value stream allocated inside iteration 1's combine)."""
    c = load_case("rmat12"); nv = c["num_vertices"]
    G = gt.Graph(); G.load_edges(c["edges"], nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1); assert V.stats.allocs_in_execute == 0
    P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V); P.execute(20); assert P.stats.allocs_in_execute == 0
    P.initialize(V); P.execute(); assert P.stats.allocs_in_execute == 0
    P.free(); V.free(); G.free()
    for app in ("bfs", "cc", "sssp"):
        weighted = app == "sssp"
        G = gt.Graph(weighted=weighted)
        flags = {"bfs": (False, False, False, False, False), "cc": (False, False, True, False, False), "sssp": (True, True, False, False, False)}[app]
        G.load_edges(c["wedges"] if weighted else c["edges"], nv, nv, *flags, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = {"bfs": gt.BFS_Program, "cc": gt.CC_Program, "sssp": gt.SSSP_Program}[app](G, False, app != "bfs", app == "bfs", gt._ROW_)
        P.root = c["root"]; P.execute(); assert P.stats.allocs_in_execute == 0, app
        P.initialize(); P.execute(); assert P.stats.allocs_in_execute == 0, app
        P.free(); G.free()


@pytest.mark.parametrize("caps", [None, ("4", "1000000"), ("4096", "16"), ("64", "256")])
@pytest.mark.parametrize("scale,seed", [(12, 3), (16, 2)])
def test_persistent_tail_kernel_is_bit_exact(gt, O, scale, seed, caps, monkeypatch):
    """The tail of BFS / SSSP / CC in one launch (gt_tail_try: one workgroup loops over whole iterations on the device while the
    frontier list is short): states and iteration counts equal the oracle's and the run with the kernel switched off -- with
    the default limits and with tiny ones that make it give up on a long list / on too many entries and come back later."""
    from graphtap_amd.rmat import rmat_edges
    nv = 1 << scale
    w = rmat_edges(scale, 16, seed, weighted=True); e = np.ascontiguousarray(w[:, :2])
    if caps:
        monkeypatch.setenv("GRAPHTAP_TAIL_LIST", caps[0]); monkeypatch.setenv("GRAPHTAP_TAIL_ENTRIES", caps[1])
    for app in ("bfs", "sssp", "cc"):
        ref = O.run_app(app, w if app == "sssp" else e, nv, root=1)
        monkeypatch.setenv("GRAPHTAP_TAIL_KERNEL", "1")
        on = run_min(gt, app, w if app == "sssp" else e, nv, 1)
        monkeypatch.setenv("GRAPHTAP_TAIL_KERNEL", "0")
        off = run_min(gt, app, w if app == "sssp" else e, nv, 1)
        assert on["iterations"] == off["iterations"] == ref["iterations"], app
        for f in ("parent", "hops", "distance", "label"):
            if f in ref:
                assert (on[f] == ref[f]).all() and (off[f] == ref[f]).all(), (app, f)
        assert on["checksum"] == off["checksum"]
        assert on["stats"].list_iterations >= off["stats"].list_iterations
        if caps is None:
            assert on["stats"].list_iterations > 0
