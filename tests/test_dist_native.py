"""The C++ multi-rank driver (graphtap_amd/csrc/dist.hip: gt_dist_execute, host code in C++ calling RCCL directly).

* p ranks of one process on one GPU over the LOOPBACK transport (one host thread per rank, device copies instead of
  RCCL -- which refuses two ranks on one device): all five programs against the reference's golden vectors and
  against the Python driver (graphtap_amd.dist.run).
* the application mains under the built-in launcher: `GRAPHTAP_NGPUS=1 GRAPHTAP_FORCE_EXCHANGE=1 apps/bin/pr ...`
  forks one rank process, which runs every grouped ncclSend/ncclRecv round and ncclAllReduce of an N-GPU run with
  itself (world size 1 over RCCL); printed lines must equal the plain single-rank run's.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_case

pytestmark = pytest.mark.gpu

PR_RTOL = 1e-6


@pytest.fixture(scope="module")
def gt():
    import graphtap_amd as gt
    gt._lib.require_gpu()
    gt._lib.check(gt._lib.lib().gt_set_device(0))
    return gt


def _dist_execute_all(gt, dists, progs, iters):
    """gt_dist_execute of every rank on its own host thread (ctypes releases the GIL during the call)"""
    L = gt._lib.lib()
    out = [None] * len(progs)

    def work(r):
        st = gt._lib.ExecStats()
        L.gt_set_device(0)
        rc = L.gt_dist_execute(dists[r], progs[r]._handle(), iters, C.byref(st))
        out[r] = (rc, L.gt_last_error().decode() if rc else "", st.iterations, st.converged)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(len(progs))]
    for t in ts: t.start()
    for t in ts: t.join()
    for r, (rc, msg, _, _) in enumerate(out):
        assert rc == 0, "rank %d: %s" % (r, msg)
    assert len({o[2] for o in out}) == 1
    for p in progs:
        p._already_initialized = True
    return out[0][2], bool(out[0][3])


def _gather(progs, field, n):
    out = np.zeros(n, progs[0].V[field].dtype); seen = np.zeros(n, bool)
    for p in progs:
        vids = p.G.vertex_ids(); keep = vids != 0xFFFFFFFF
        out[vids[keep]] = p.V[field][keep]; assert not seen[vids[keep]].any(); seen[vids[keep]] = True
    assert seen.all()
    return out


@pytest.mark.parametrize("nranks,slices,variant", [(2, 4, "pb"), (3, 2, "pb_f32msg"), (8, 1, "pb"), (4, 2, "edge")])
@pytest.mark.parametrize("name", ["tiny", "rmat10", "rmat12"])
def test_cpp_driver_over_loopback_reproduces_the_reference(gt, name, nranks, slices, variant, monkeypatch, known_answers):
    L = gt._lib.lib()
    monkeypatch.setenv("GRAPHTAP_SPMV", variant)
    monkeypatch.setenv("GRAPHTAP_X_SLICES", str(slices))
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1
    hs = (C.c_void_p * nranks)()
    gt._lib.check(L.gt_dist_create_loopback(hs, nranks))
    dists = [C.c_void_p(hs[r]) for r in range(nranks)]
    try:
        # apps/pr.cpp: Deg in _COL_ order (all-reduce of the column counts), then PageRank, fixed count and converge mode
        Gs = []
        for r in range(nranks):
            G = gt.Graph(); G.load_edges(c["edges"], nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=r, nranks=nranks); Gs.append(G)
        degs = [gt.Deg_Program(G, True, False, False, gt._COL_) for G in Gs]
        for p in degs: p.initialize()
        it, _ = _dist_execute_all(gt, dists, degs, 1)
        assert it == 1
        prs = [gt.PR_Program(G, True, False, False, gt._ROW_) for G in Gs]
        for p, d in zip(prs, degs): p.initialize(d)
        it, conv = _dist_execute_all(gt, dists, prs, 20)
        assert it == 20 and not conv
        assert (_gather(prs, "degree", n) == c["np1_pr20_a"]).all()
        ref = c["np1_pr20_c"]
        assert (np.abs(_gather(prs, "rank", n) - ref) / ref).max() < PR_RTOL
        # converge mode (the f32-message variant runs f64 messages there, gt_program_prepare: the count is the reference's)
        for p, d in zip(prs, degs): p.initialize(d)
        it, conv = _dist_execute_all(gt, dists, prs, 0)
        assert conv and it == known_answers[name]["np1_prconv_cf"]["iterations"]
        ref = c["np1_prconv_cf_c"]
        assert (np.abs(_gather(prs, "rank", n) - ref) / ref).max() < PR_RTOL
        for p in prs + degs: p.free()
        for G in Gs: G.free()
        # BFS
        Gs = []
        for r in range(nranks):
            G = gt.Graph(); G.load_edges(c["edges"], nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks); Gs.append(G)
        ps = [gt.BFS_Program(G, False, False, True, gt._ROW_) for G in Gs]
        for p in ps: p.root = c["root"]; p.initialize()
        for d in dists: gt._lib.check(L.gt_dist_exchange_stats(d, None, None, None, 1))
        it, conv = _dist_execute_all(gt, dists, ps, 0)
        assert conv and it == known_answers[name]["np1_bfs"]["iterations"]
        assert (_gather(ps, "parent", n) == c["np1_bfs_a"]).all() and (_gather(ps, "hops", n) == c["np1_bfs_b"]).all()
        # sparse frontier exchange (vp:970-1013): a BFS frontier is a few vertices for most of its iterations, so the ranks
        # must have shipped fewer bytes than the dense blocks hold -- unless it is switched off
        sent, dense, nx = C.c_uint64(), C.c_uint64(), C.c_uint64()
        tot_sent = tot_dense = 0
        for d in dists:
            gt._lib.check(L.gt_dist_exchange_stats(d, C.byref(sent), C.byref(dense), C.byref(nx), 1))
            tot_sent += sent.value; tot_dense += dense.value
            assert nx.value == it
        assert tot_sent < tot_dense, (tot_sent, tot_dense)
        monkeypatch.setenv("GRAPHTAP_SPARSE_EXCHANGE", "0")
        for p in ps: p.initialize()
        it2, _ = _dist_execute_all(gt, dists, ps, 0)
        assert it2 == it and (_gather(ps, "parent", n) == c["np1_bfs_a"]).all()
        tot_sent = tot_dense = 0
        for d in dists:
            gt._lib.check(L.gt_dist_exchange_stats(d, C.byref(sent), C.byref(dense), C.byref(nx), 1))
            tot_sent += sent.value; tot_dense += dense.value
        assert tot_sent == tot_dense
        monkeypatch.delenv("GRAPHTAP_SPARSE_EXCHANGE")
        for p in ps: p.free()
        for G in Gs: G.free()
        # CC (self loops kept)
        Gs = []
        for r in range(nranks):
            G = gt.Graph(); G.load_edges(c["edges"], nv, nv, False, False, True, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks); Gs.append(G)
        ps = [gt.CC_Program(G, False, True, False, gt._ROW_) for G in Gs]
        for p in ps: p.initialize()
        it, conv = _dist_execute_all(gt, dists, ps, 0)
        assert conv and it == known_answers[name]["np1_cc"]["iterations"]
        assert (_gather(ps, "label", n) == c["np1_cc_a"]).all()
        for p in ps: p.free()
        for G in Gs: G.free()
        # SSSP
        Gs = []
        for r in range(nranks):
            G = gt.Graph(weighted=True); G.load_edges(c["wedges"], nv, nv, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks); Gs.append(G)
        ps = [gt.SSSP_Program(G, False, True, False, gt._ROW_) for G in Gs]
        for p in ps: p.root = c["root"]; p.initialize()
        it, conv = _dist_execute_all(gt, dists, ps, 0)
        assert conv and it == known_answers[name]["np1_sssp"]["iterations"]
        assert (_gather(ps, "distance", n) == c["np1_sssp_a"]).all()
        for p in ps: p.free()
        for G in Gs: G.free()
    finally:
        for d in dists: L.gt_dist_free(d)


@pytest.mark.parametrize("nranks,slices,variant", [(2, 4, "pb_f32msg"), (3, 2, "pb"), (8, 2, "pb_f32msg")])
@pytest.mark.parametrize("name", ["tiny", "rmat12"])
def test_phase2_part_by_part_loop_gives_the_same_ranks(gt, name, nranks, slices, variant, monkeypatch):
    """The pipelined loop of a fixed-count PageRank (GRAPHTAP_P2_PARTS=1; built and measured in round 4, off by default: it loses on
    the tile-rows it was meant for, engine.hip gt_program_parts_begin): phase 2 in K parts, slice k of the next iteration's
    exchange issued behind part k. Ranks against the reference's vectors and against the ordinary loop, runs continued by a
    second execute, tile-rows without entries (8 ranks on the tiny graph) included."""
    L = gt._lib.lib()
    monkeypatch.setenv("GRAPHTAP_SPMV", variant)
    monkeypatch.setenv("GRAPHTAP_X_SLICES", str(slices))
    monkeypatch.setenv("GRAPHTAP_TIMEOUT_S", "60")
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1
    hs = (C.c_void_p * nranks)()
    gt._lib.check(L.gt_dist_create_loopback(hs, nranks))
    dists = [C.c_void_p(hs[r]) for r in range(nranks)]
    try:
        Gs = []
        for r in range(nranks):
            G = gt.Graph(); G.load_edges(c["edges"], nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=r, nranks=nranks); Gs.append(G)
        degs = [gt.Deg_Program(G, True, False, False, gt._COL_) for G in Gs]
        for p in degs: p.initialize()
        _dist_execute_all(gt, dists, degs, 1)
        prs = [gt.PR_Program(G, True, False, False, gt._ROW_) for G in Gs]
        out = {}
        for parts in ("0", "1"):
            monkeypatch.setenv("GRAPHTAP_P2_PARTS", parts)
            for p, d in zip(prs, degs): p.initialize(d)
            it, _ = _dist_execute_all(gt, dists, prs, 3)
            assert it == 3
            r3 = _gather(prs, "rank", n)
            it, _ = _dist_execute_all(gt, dists, prs, 20)          # continued: 17 more
            assert it == 20
            out[parts] = (r3, _gather(prs, "rank", n))
        ref3, ref20 = c["np1_pr3_c"], c["np1_pr20_c"]
        for parts in out:
            assert (np.abs(out[parts][0] - ref3) / ref3).max() < PR_RTOL and (np.abs(out[parts][1] - ref20) / ref20).max() < PR_RTOL
        assert (np.abs(out["1"][1] - out["0"][1]) / out["0"][1]).max() < 1e-12
        for p in prs + degs: p.free()
        for G in Gs: G.free()
    finally:
        for d in dists: L.gt_dist_free(d)


def _min_apps_over_loopback(gt, c, nranks, known_answers, name):
    """BFS, CC, SSSP of one case on `nranks` loopback ranks: labels and iteration counts against the reference's goldens;
    returns the summed list_iterations / spmspv_iterations the ranks reported."""
    L = gt._lib.lib()
    nv = c["num_vertices"]; n = nv + 1
    hs = (C.c_void_p * nranks)()
    gt._lib.check(L.gt_dist_create_loopback(hs, nranks))
    dists = [C.c_void_p(hs[r]) for r in range(nranks)]
    tot = {"list": 0, "spmspv": 0}

    def run(progs):
        out = [None] * nranks

        def work(r):
            st = gt._lib.ExecStats()
            L.gt_set_device(0)
            rc = L.gt_dist_execute(dists[r], progs[r]._handle(), 0, C.byref(st))
            out[r] = (rc, L.gt_last_error().decode() if rc else "", st.iterations, st.converged, st.list_iterations, st.spmspv_iterations)
        ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
        for t in ts: t.start()
        for t in ts: t.join()
        for r, o in enumerate(out):
            assert o[0] == 0, "rank %d: %s" % (r, o[1])
        assert len({o[2] for o in out}) == 1 and all(o[3] for o in out)
        tot["list"] += sum(o[4] for o in out); tot["spmspv"] += sum(o[5] for o in out)
        for p in progs: p._already_initialized = True
        return out[0][2]
    try:
        for app, flags, weighted, fields, key in (
                ("bfs", (False, False, False, False, False), False, (("parent", "np1_bfs_a"), ("hops", "np1_bfs_b")), "np1_bfs"),
                ("cc", (False, False, True, False, False), False, (("label", "np1_cc_a"),), "np1_cc"),
                ("sssp", (True, True, False, False, False), True, (("distance", "np1_sssp_a"),), "np1_sssp")):
            Gs = []
            for r in range(nranks):
                G = gt.Graph(weighted=weighted)
                G.load_edges(c["wedges"] if weighted else c["edges"], nv, nv, *flags, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks); Gs.append(G)
            cls = {"bfs": gt.BFS_Program, "cc": gt.CC_Program, "sssp": gt.SSSP_Program}[app]
            args = {"bfs": (False, False, True), "cc": (False, True, False), "sssp": (False, True, False)}[app]
            ps = [cls(G, *args, gt._ROW_) for G in Gs]
            for p in ps:
                p.root = c["root"]; p.initialize()
            it = run(ps)
            assert it == known_answers[name][key]["iterations"], (app, it)
            for f, gold in fields:
                assert (_gather(ps, f, n) == c[gold]).all(), (app, f)
            for p in ps: p.initialize()      # a second run on the same programs and communicator
            assert run(ps) == it
            for f, gold in fields:
                assert (_gather(ps, f, n) == c[gold]).all(), (app, f, "second run")
            for p in ps: p.free()
            for G in Gs: G.free()
    finally:
        for d in dists: L.gt_dist_free(d)
    return tot


@pytest.mark.parametrize("mode", ["by_size", "lists_forced", "lists_forced_streaming", "lists_counted_then_streamed", "lists_off", "dense_protocol"])
@pytest.mark.parametrize("nranks,slices", [(2, 1), (3, 2), (4, 4), (8, 2)])
@pytest.mark.parametrize("name", ["tiny", "rmat10", "rmat12"])
def test_frontier_lists_on_several_ranks_are_bit_exact(gt, name, nranks, slices, mode, monkeypatch, known_answers):
    """The reference's sparse path at any np (vp:711-784, 970-1013, 1475-1489): per-rank frontier lists travel as (index, value)
    pairs, the receiving rank runs the SpMSpV straight from the pairs and applies the rows it lowered; counts ride the
    convergence all-reduce. Labels AND iteration counts are the reference's with the path forced, forbidden and chosen by size."""
    monkeypatch.setenv("GRAPHTAP_X_SLICES", str(slices))
    if mode == "lists_forced":
        monkeypatch.setenv("GRAPHTAP_DIST_LISTS", "1"); monkeypatch.setenv("GRAPHTAP_SPMSPV", "1")
    elif mode == "lists_forced_streaming":   # pairs travel, but they are scattered into x and the streaming pass runs
        monkeypatch.setenv("GRAPHTAP_DIST_LISTS", "1"); monkeypatch.setenv("GRAPHTAP_SPMSPV", "0")
    elif mode == "lists_counted_then_streamed":   # the pairs' entries are counted, found too many, and the pairs go through x after all
        monkeypatch.setenv("GRAPHTAP_DIST_LISTS", "1"); monkeypatch.setenv("GRAPHTAP_SPMSPV_FRACTION", "1")
        monkeypatch.setenv("GRAPHTAP_DIST_EXACT_FROM", "0"); monkeypatch.setenv("GRAPHTAP_DIST_MAX_ENTRIES", "3")
    elif mode == "lists_off":
        monkeypatch.setenv("GRAPHTAP_DIST_LISTS", "0")
    elif mode == "dense_protocol":
        monkeypatch.setenv("GRAPHTAP_DIST_PROTOCOL", "dense")
    tot = _min_apps_over_loopback(gt, load_case(name), nranks, known_answers, name)
    if mode == "lists_forced":
        assert tot["list"] > 0 and tot["spmspv"] > 0, tot
    if mode == "dense_protocol":
        assert tot["list"] == 0


@pytest.mark.parametrize("split", ["contiguous", "round_robin", "all_on_rank_0"])
@pytest.mark.parametrize("nranks,slices", [(2, 2), (3, 1), (8, 2)])
@pytest.mark.parametrize("name,app", [("tiny", "pr"), ("rmat10", "bfs"), ("rmat12", "sssp"), ("rmat12", "pr")])
def test_distributed_build_equals_the_replicated_build(gt, name, app, nranks, slices, split, monkeypatch, known_answers):
    """Matrix::distribute (mat/matrix.hpp:693-810): every rank brings a share of the records, the build shuffles them to the owners
    of their rows and takes the global pieces from collectives. Whatever the split, rank r's graph is the one gt_graph_build
    gives it from the full list -- info, TCSC arrays, exchange plan -- and the programs run on it to the reference's vectors."""
    L = gt._lib.lib()
    monkeypatch.setenv("GRAPHTAP_X_SLICES", str(slices))
    c = load_case(name); nv = c["num_vertices"]; n = nv + 1
    weighted = app == "sssp"
    e = c["wedges"] if weighted else c["edges"]
    flags = {"pr": (True, True, True, False, True), "bfs": (False, False, False, False, False), "sssp": (True, True, False, False, False)}[app]
    ct = gt._TCSC_CF_ if app == "pr" else gt._TCSC_
    shares = {"contiguous": [e[len(e) * r // nranks: len(e) * (r + 1) // nranks] for r in range(nranks)],
              "round_robin": [e[r::nranks] for r in range(nranks)],
              "all_on_rank_0": [e if r == 0 else e[:0] for r in range(nranks)]}[split]
    hs = (C.c_void_p * nranks)()
    gt._lib.check(L.gt_dist_create_loopback(hs, nranks))
    dists = [C.c_void_p(hs[r]) for r in range(nranks)]
    Gs, errs = [None] * nranks, [None] * nranks

    def build(r):
        try:
            L.gt_set_device(0)
            G = gt.Graph(weighted=weighted)
            G.load_share(dists[r], r, nranks, shares[r], nv, nv, *flags, gt._2DT_, ct)
            Gs[r] = G
        except Exception as ex:   # noqa: BLE001
            errs[r] = ex
    ts = [threading.Thread(target=build, args=(r,)) for r in range(nranks)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not any(errs), errs
    try:
        for r in range(nranks):
            R = gt.Graph(weighted=weighted); R.load_edges(e, nv, nv, *flags, gt._2DT_, ct, rank=r, nranks=nranks)
            a, b = Gs[r].info, R.info
            for f, _ in a._fields_:
                assert getattr(a, f) == getattr(b, f), (r, f, getattr(a, f), getattr(b, f))
            ta, tb = Gs[r].tile_to_host(), R.tile_to_host()
            for k in ta:
                if ta[k] is None: assert tb[k] is None
                elif k == "A":   # duplicates of a (row, col) pair keep the minimum weight either way; equal after sorting inside a column is implied by IA equality + dedupe
                    assert (ta[k] == tb[k]).all(), (r, k)
                else: assert (ta[k] == tb[k]).all(), (r, k)
            pa, pb = Gs[r].exchange_plan(), R.exchange_plan()
            assert pa == pb, (r, "exchange plan")
            R.free()
        if app == "pr":
            degs = [gt.Deg_Program(G, True, False, False, gt._COL_) for G in Gs]
            for p in degs: p.initialize()
            _dist_execute_all(gt, dists, degs, 1)
            ps = [gt.PR_Program(G, True, False, False, gt._ROW_) for G in Gs]
            for p, d in zip(ps, degs): p.initialize(d)
            _dist_execute_all(gt, dists, ps, 20)
            ref = c["np1_pr20_c"]
            assert (np.abs(_gather(ps, "rank", n) - ref) / ref).max() < PR_RTOL
            for p in ps + degs: p.free()
        else:
            ps = [(gt.BFS_Program(G, False, False, True, gt._ROW_) if app == "bfs" else gt.SSSP_Program(G, False, True, False, gt._ROW_)) for G in Gs]
            for p in ps: p.root = c["root"]; p.initialize()
            it, conv = _dist_execute_all(gt, dists, ps, 0)
            assert conv and it == known_answers[name]["np1_" + app]["iterations"]
            f, gold = ("parent", "np1_bfs_a") if app == "bfs" else ("distance", "np1_sssp_a")
            assert (_gather(ps, f, n) == c[gold]).all()
            for p in ps: p.free()
    finally:
        for G in Gs:
            if G is not None: G.free()
        for d in dists: L.gt_dist_free(d)


@pytest.mark.parametrize("nranks,bad_rank", [(2, 1), (3, 0), (8, 5)])
def test_distributed_build_with_a_bad_record_on_one_rank_fails_on_every_rank_together(gt, nranks, bad_rank, monkeypatch):
    """One out-of-range vertex id in ONE rank's share (the reference overflows its tile grid silently, matrix.hpp:218-220; this
    library rejects the input): after the shuffle only the record's new holder sees it. Every rank must return an error from
    gt_graph_build_distributed -- none may stay behind in the collectives that follow -- and quickly (the status word is summed
    before every collective stage, ingest.hip: ing_agree), not at the deadline. The ranks then build a good graph on the same
    communicator: it is still usable."""
    L = gt._lib.lib()
    monkeypatch.setenv("GRAPHTAP_X_SLICES", "2")
    monkeypatch.setenv("GRAPHTAP_TIMEOUT_S", "60")
    c = load_case("rmat12"); nv = c["num_vertices"]
    e = c["edges"]
    flags = (True, True, True, False, True)
    hs = (C.c_void_p * nranks)()
    gt._lib.check(L.gt_dist_create_loopback(hs, nranks))
    dists = [C.c_void_p(hs[r]) for r in range(nranks)]

    def run(shares):
        Gs, errs = [None] * nranks, [None] * nranks

        def build(r):
            try:
                L.gt_set_device(0)
                G = gt.Graph(); G.load_share(dists[r], r, nranks, shares[r], nv, nv, *flags, gt._2DT_, gt._TCSC_CF_)
                Gs[r] = G
            except Exception as ex:   # noqa: BLE001
                errs[r] = str(ex)
        ts = [threading.Thread(target=build, args=(r,)) for r in range(nranks)]
        for t in ts: t.start()
        for t in ts: t.join(timeout=45)
        assert not any(t.is_alive() for t in ts), "a rank is still inside the build: its peers left it in a collective"
        return Gs, errs
    good = [e[r::nranks] for r in range(nranks)]
    bad = [np.array(x, copy=True) for x in good]
    bad[bad_rank][len(bad[bad_rank]) // 2, 1] = nv + 12345          # one record names a vertex that does not exist
    Gs, errs = run(bad)
    assert all(G is None for G in Gs) and all(errs), errs
    assert sum("vertex id > num_vertices" in x for x in errs) >= 1, errs          # the holder's own message
    assert sum("failed while" in x for x in errs) == nranks - sum("vertex id > num_vertices" in x for x in errs), errs
    Gs, errs = run(good)
    assert not any(errs), errs
    for G in Gs: G.free()
    for d in dists: L.gt_dist_free(d)


def test_mismatched_exchange_plans_are_reported_not_hung(gt, monkeypatch):
    """Two ranks whose graphs were built from different edge lists: the first execute compares every rank's send counts with
    every receiver's recv counts (plan_verify) and every rank returns an error -- the grouped send/recv rounds would hang."""
    L = gt._lib.lib()
    monkeypatch.setenv("GRAPHTAP_X_SLICES", "2")
    c = load_case("rmat10"); nv = c["num_vertices"]
    hs = (C.c_void_p * 2)()
    gt._lib.check(L.gt_dist_create_loopback(hs, 2))
    dists = [C.c_void_p(hs[r]) for r in range(2)]
    Gs = []
    for r in range(2):
        e = c["edges"] if r == 0 else c["edges"][: len(c["edges"]) // 2]      # rank 1 saw half of the list
        G = gt.Graph(); G.load_edges(e, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=2); Gs.append(G)
    ps = [gt.BFS_Program(G, False, False, True, gt._ROW_) for G in Gs]
    for p in ps: p.root = c["root"]; p.initialize()
    out = [None, None]

    def work(r):
        st = gt._lib.ExecStats(); L.gt_set_device(0)
        rc = L.gt_dist_execute(dists[r], ps[r]._handle(), 0, C.byref(st))
        out[r] = (rc, L.gt_last_error().decode())
    ts = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a rank hangs"
    assert all(o is not None and o[0] != 0 for o in out), out
    assert any("exchange plan mismatch" in o[1] for o in out), out
    for p in ps: p.free()
    for G in Gs: G.free()
    for d in dists: L.gt_dist_free(d)


def _app(app, args, env_extra):
    exe = os.path.join(ROOT, "apps", "bin", app)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "apps"), "all"])
    env = dict(os.environ, **env_extra)
    for k in ("GRAPHTAP_SPMV",):
        env.pop(k, None)
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    keep = ("Iterations:", "Value checksum:", "Reachable vertices:", "vertex[")
    return [l for l in r.stdout.splitlines() if l.startswith(keep)]


@pytest.mark.parametrize("slices", [1, 2, 4])
def test_mains_under_the_rccl_launcher_print_the_single_rank_lines(slices):
    """World size 1 over RCCL through the C++ host path: Env::init() forks the rank process before any HIP call, the rank
    creates its communicator from the unique id, and gt_dist_execute runs the grouped send/recv rounds, the
    ncclAllReduce of Degree's column counts and of the convergence count with itself."""
    launch = {"GRAPHTAP_NGPUS": "1", "GRAPHTAP_FORCE_EXCHANGE": "1", "GRAPHTAP_X_SLICES": str(slices)}
    f, fw = os.path.join(GOLDEN, "rmat10_1024.bin"), os.path.join(GOLDEN, "rmat10_1024_w.bin")
    for app, args in (("pr", (f, 1024, 20)), ("pr", (f, 1024)), ("bfs", (f, 1024, 0)), ("cc", (f, 1024)), ("sssp", (fw, 1024, 0))):
        plain = _app(app, args, {})
        multi = _app(app, args, launch)
        assert multi == plain and len(plain) >= 34, (app, args, plain[:4], multi[:4])
    out = _app("pr", (f, 1024, 20), launch)
    assert "Value checksum: 70" in out and "vertex[4]:Rank=1.238176,Degree=2" in out   # SURVEY 8c known answers
