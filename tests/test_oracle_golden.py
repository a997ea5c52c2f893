"""Pins the CPU oracle (oracle/gt_oracle.c) to the UNMODIFIED reference.

Every expected value here was produced by running the reference itself
(tests/golden/make_golden.py -> oracle/_ref/dump_*): full vertex-state
vectors for Deg / PageRank / BFS / SSSP / CC on the reference's bundled
rmat10 samples and on seeded synthetic graphs, at np=1 and at np>1, plus the
lines its own checksum() prints. Integer programs must match bit for bit;
PageRank must match the np=1 run bit for bit in fp64 (same summation order)
and the np>1 runs to 1e-6 relative (the reference itself re-associates the
sums across tiles there).
"""
import os

import numpy as np
import pytest

from conftest import CASES, CF_CASES, OTHER_NP, load_case, load_cf_case
from oracle import oracle as O


@pytest.mark.parametrize("name", CASES)
def test_deg(name, known_answers):
    c = load_case(name)
    r = O.run_app("deg", c["edges"], c["num_vertices"])
    n = c["num_vertices"] + 1
    assert (r["degree"][:n] == c["np1_deg_a"]).all()
    ka = known_answers[name]["np1_deg"]
    assert O.checksum_u32(r["degree"], n, 0) == (ka["checksum"], ka["reachable"])


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("iters", [1, 3, 20])
def test_pagerank_fixed_iterations_bit_exact(name, iters, known_answers):
    c = load_case(name)
    n = c["num_vertices"] + 1
    key = "np1_pr%d" % iters
    for cf in (True, False):  # TCSC_CF (apps/pr.cpp) and TCSC (apps/pr1.cpp) agree for a fixed count
        r = O.run_app("pr", c["edges"], c["num_vertices"], iters=iters, cf=cf)
        assert r["iterations"] == iters
        assert (r["degree"][:n] == c[key + "_a"]).all()
        assert (r["rank"][:n] == c[key + "_c"]).all(), np.abs(r["rank"][:n] - c[key + "_c"]).max()
    ka = known_answers[name][key]
    assert O.checksum_f64(r["rank"], n) == (ka["checksum"], ka["reachable"])
    if iters == 20:
        assert (r["rank"][:n] == c["np1_pr1app20_c"]).all()
        other = c["np%d_pr20_c" % OTHER_NP[name]]
        assert np.allclose(r["rank"][:n], other, rtol=1e-6, atol=0)


@pytest.mark.parametrize("name", CASES)
def test_pagerank_converge_mode(name, known_answers):
    """iters=0: TCSC (pr1) and the TCSC_CF quirk of the shipped `pr` (SURVEY 8a trap 5)."""
    c = load_case(name)
    n = c["num_vertices"] + 1
    for cf, key in ((False, "np1_prconv_tcsc"), (True, "np1_prconv_cf")):
        r = O.run_app("pr", c["edges"], c["num_vertices"], iters=0, cf=cf)
        assert r["iterations"] == known_answers[name][key]["iterations"]
        assert (r["rank"][:n] == c[key + "_c"]).all()
        ka = known_answers[name][key]
        assert O.checksum_f64(r["rank"], n) == (ka["checksum"], ka["reachable"])


@pytest.mark.parametrize("name", CASES)
def test_bfs(name, known_answers):
    c = load_case(name)
    n = c["num_vertices"] + 1
    roots = [(c["root"], "np1_bfs")] + ([(0, "np1_bfs0")] if c["root"] != 0 and "np1_bfs0_a" in c else [])
    for root, key in roots:
        r = O.run_app("bfs", c["edges"], c["num_vertices"], root=root)
        assert (r["parent"][:n] == c[key + "_a"]).all()
        assert (r["hops"][:n] == c[key + "_b"]).all()
        ka = known_answers[name][key]
        assert r["iterations"] == ka["iterations"]
        assert O.checksum_u32(r["hops"], n, O.INF) == (ka["checksum"], ka["reachable"])


@pytest.mark.parametrize("name", CASES)
def test_sssp(name, known_answers):
    c = load_case(name)
    n = c["num_vertices"] + 1
    roots = [(c["root"], "np1_sssp")] + ([(0, "np1_sssp0")] if c["root"] != 0 and "np1_sssp0_a" in c else [])
    for root, key in roots:
        r = O.run_app("sssp", c["wedges"], c["num_vertices"], root=root)
        assert (r["distance"][:n] == c[key + "_a"]).all()
        ka = known_answers[name][key]
        assert r["iterations"] == ka["iterations"]
        assert O.checksum_u32(r["distance"], n, O.INF) == (ka["checksum"], ka["reachable"])


@pytest.mark.parametrize("name", CASES)
def test_cc(name, known_answers):
    c = load_case(name)
    n = c["num_vertices"] + 1
    r = O.run_app("cc", c["edges"], c["num_vertices"])
    assert (r["label"][:n] == c["np1_cc_a"]).all()
    ka = known_answers[name]["np1_cc"]
    assert r["iterations"] == ka["iterations"]
    assert O.checksum_u32(r["label"], n, O.INF) == (ka["checksum"], ka["reachable"])


def test_survey_known_answers_on_bundled_sample(known_answers):
    """The table in SURVEY.md section 8c, captured again from the running reference."""
    k = known_answers["rmat10"]
    assert k["np1_pr20"] == {"iterations": 20, "checksum": 70, "reachable": 1025}
    assert k["np1_bfs"] == {"iterations": 4, "checksum": 1912, "reachable": 887}
    assert k["np1_cc"] == {"iterations": 4, "checksum": 69590, "reachable": 1025}
    assert k["np1_sssp"] == {"iterations": 7, "checksum": 53366, "reachable": 471}
    assert k["np1_deg"] == {"iterations": 1, "checksum": 16384, "reachable": 571}
    assert k["np4_pr20"] == k["np1_pr20"] and k["np4_bfs"] == k["np1_bfs"]
    # singlenode harness: TCSC kernel (5) class counts and the common final value
    sn = known_answers["singlenode_rmat10"]
    assert any("317.018" in l for l in sn["0"]) and any("317.018" in l for l in sn["5"])


def test_tcsc_structure_matches_singlenode_harness():
    """nnzrows=866 regulars=550 sources=316 / nnzcols=571 regulars=550 sinks=21
    (printed by /root/reference/src/singlenode/tcsc_spmspv2.hpp for rmat10)."""
    c = load_case("rmat10")
    g = O.OracleGraph(c["edges"], 1024, **O.APP_FLAGS["pr"])
    assert (g.nnzrows, g.nnzcols, g.nnz) == (866, 571, 16384)
    assert g.class_counts() == {"regular": 550, "source_rows": 316, "sink_cols": 21}
    JA, IA = g.JA, g.IA
    assert JA[0] == 0 and JA[-1] == g.nnz and (np.diff(JA.astype(np.int64)) > 0).all()
    assert IA.max() == g.nnzrows - 1
    # BFS/CC build: symmetrised, self loops dropped (bfs) / kept (cc), deduped
    gb = O.OracleGraph(c["edges"], 1024, **O.APP_FLAGS["bfs"])
    gc = O.OracleGraph(c["edges"], 1024, **O.APP_FLAGS["cc"])
    e = c["edges"].astype(np.int64)
    pairs = set(map(tuple, e[e[:, 0] != e[:, 1]])) | set((b, a) for a, b in map(tuple, e[e[:, 0] != e[:, 1]]))
    assert gb.nnz == len(pairs)
    assert gc.nnz == len(pairs) + len(set(map(tuple, e[e[:, 0] == e[:, 1]])))


@pytest.mark.parametrize("name", CF_CASES)
def test_tcsc_cf_format_is_the_references(name):
    """The oracle's TCSC_CF restatement (source rows swapped to the tail of every column in the reference's swap order, the
    four pair lists with their quirks) against the arrays dumped from the unmodified reference's compressor
    (ds/compressed_column.hpp:603-1120; oracle/ref/dump_main.cpp -DAPP_TCSC_CF)."""
    edges, nv, ref = load_cf_case(name)
    g = O.OracleGraph(edges, nv, **O.APP_FLAGS["pr"])
    assert (g.JA == ref["JA"]).all() and (g.JC == ref["JC"]).all() and (g.IR == ref["IR"]).all()
    cf = g.tcsc_cf()
    for k in ("IA", "JA_REG_R_NNZ_C"):
        assert cf[k].shape == ref[k].shape and (cf[k] == ref[k]).all(), k
    for l in g.CF_LISTS:
        assert cf["NC_" + l] == int(ref["NC_" + l]), l
        for k in ("JA_" + l, "JC_" + l):
            assert cf[k].shape == ref[k].shape and (cf[k] == ref[k]).all(), k
    # what the format promises: per column the same rows as TCSC, source rows last
    src = ~np.isin(g.IR, g.JC)                       # compressed row -> its vertex has no column
    for j in range(g.nnzcols):
        a, b = g.JA[j], g.JA[j + 1]
        assert sorted(cf["IA"][a:b]) == sorted(g.IA[a:b])
        s = src[cf["IA"][a:b]]
        assert not (s[:-1] & ~s[1:]).any()          # never a regular row after a source row
    if name in ("rmat8", "rmat10", "rmat12", "mixed"):
        assert (cf["IA"] != g.IA).any()              # the swap had work to do


@pytest.mark.parametrize("name", CF_CASES)
def test_tcsc_cf_spmv_adds_up_to_the_plain_spmv(name):
    """vp:1243-1317 over the pair lists, all three conditions on, equals the TCSC SpMV whenever sink columns message 0
    (they always do in PageRank: such a vertex has no row, hence degree 0 after initialize(other), vp:476-483)."""
    edges, nv, _ = load_cf_case(name)
    g = O.OracleGraph(edges, nv, **O.APP_FLAGS["pr"])
    rng = np.random.RandomState(5)
    x = rng.randint(1, 1000, g.nnzcols).astype(np.float64)
    x[~np.isin(g.JC, g.IR)] = 0                      # sink columns
    want = g.spmv_plus_f64(x, np.zeros(g.nnzrows))
    got = g.spmv_cf_plus_f64(x, np.zeros(g.nnzrows), first=True, running=True, last=True)
    assert (got == want).all()
    src = ~np.isin(g.IR, g.JC)
    part = g.spmv_cf_plus_f64(x, np.zeros(g.nnzrows), first=True, running=True, last=False)
    assert (part[~src] == want[~src]).all() and (part[src] == 0).all()


def test_host_pagerank_step_model_is_pinned_to_the_oracle():
    """tests/host_models.pagerank_step_from_records -- what the full-size GPU test recomputes iteration 20 of the headline
    configuration with, from the raw record stream -- equals the oracle's iteration (itself bit-identical to the reference's
    vectors above): rank_20 from rank_19 and the degrees, on the bundled sample and on a seeded R-MAT, records in ragged slices."""
    from host_models import pagerank_step_from_records
    from graphtap_amd.rmat import rmat_edges
    cases = [(np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rmat10_1024.bin"), dtype="<u4").reshape(-1, 2), 1024), (rmat_edges(12, 16, 5), 1 << 12)]
    for e, nv in cases:
        r19 = O.run_app("pr", e, nv, iters=19); r20 = O.run_app("pr", e, nv, iters=20)
        cut = len(e) // 3 + 1
        got = pagerank_step_from_records([e[:cut], e[cut:2 * cut + 5], e[2 * cut + 5:]], r19["rank"], r19["degree"])
        rel = np.abs(got - r20["rank"]) / r20["rank"]
        assert rel.max() < 1e-13, rel.max()
        r19["graph"].close(); r20["graph"].close()
