#!/usr/bin/env python3
"""Times the four non-headline programs (BFS, CC, SSSP, Degree) on synthetic R-MAT on one GPU.
GTEPS = stored entries x iterations / Execute time -- the reference's convention, which counts entries the
activity filter never touches. So for the min programs the line also carries a roofline figure on the bytes an iteration
REQUIRES: with F_t the frontier (vertices whose state changed in the previous apply; iteration 0: the root, or every
vertex for CC), E_t the stored entries of F_t's columns and D_t the vertices iteration t changes,
    required_t = 4 E_t (row ids) + 8 |F_t| (column pointer + message) + 8 |D_t| (accumulator read + state write)
(a lower bound: rows that are reached but do not change are not counted). `required_frac` = sum_t required_t / Execute time
/ 8 TB/s; the frontier sizes come from an untimed second run stepped phase by phase. Not a bench.py line: the BASELINE
configs name these as parity cases; this script gives the numbers quoted in DESIGN.md.
  python tools/bench_apps.py --scale 24 [--apps bfs,cc,sssp]"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import graphtap_amd as gt
from graphtap_amd import _lib


def required_bytes(L, G, P, app):
    """Steps the program again (scatter_gather / combine / apply) and returns (sum of required_t, per-iteration records)."""
    t, i = G.tile(), G.info
    JA = np.zeros(i.ncols_local + 1, np.uint32); JC = np.zeros(i.nnzcols, np.uint32)
    _lib.check(L.gt_memcpy_d2h(JA.ctypes.data_as(C.c_void_p), t.JA, JA.nbytes)); _lib.check(L.gt_memcpy_d2h(JC.ctypes.data_as(C.c_void_p), t.JC, JC.nbytes))
    outdeg = np.zeros(i.tile_height, np.int64); outdeg[JC] = np.diff(JA.astype(np.int64))[:i.nnzcols]
    has_col = np.zeros(i.tile_height, bool); has_col[JC] = True
    P.initialize()
    h = P._handle()
    key = {"bfs": "hops", "cc": "label", "sssp": "distance"}[app]
    prev = P.V[key].copy()
    frontier = has_col.copy() if app == "cc" else np.zeros(i.tile_height, bool)
    if app != "cc": frontier[P.root] = True
    total, per_it = 0, []
    for _ in range(1000):
        active = C.c_uint64()
        ms = []
        for call in (lambda: L.gt_program_scatter_gather(h), lambda: L.gt_program_combine(h), lambda: L.gt_program_apply(h, 0, C.byref(active))):
            _lib.check(L.gt_device_synchronize()); t0 = time.perf_counter()
            _lib.check(call()); _lib.check(L.gt_device_synchronize()); ms.append(round((time.perf_counter() - t0) * 1e3, 3))
        cur = P.V[key]
        changed = cur != prev
        f = frontier & has_col
        E, F, D = int(outdeg[f].sum()), int(f.sum()), int(changed.sum())
        total += 4 * E + 8 * F + 8 * D
        per_it.append({"frontier": F, "entries": E, "changed": D, "stepped_ms": ms})   # scatter_gather, combine, apply with a sync after each
        prev, frontier = cur.copy(), changed
        if active.value == 0: break
    return total, per_it

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=24)
ap.add_argument("--apps", default="bfs,cc,sssp")
ap.add_argument("--root", type=int, default=None)
ap.add_argument("--edge-factor", type=int, default=16)
args = ap.parse_args()
L = _lib.lib(); _lib.require_gpu(); _lib.check(L.gt_set_device(0))
scale, nv = args.scale, 1 << args.scale
m = args.edge_factor << scale
for app in args.apps.split(","):
    weighted = app == "sssp"
    d = C.c_void_p(); _lib.check(L.gt_malloc(C.byref(d), m * (12 if weighted else 8)))
    _lib.check(L.gt_rmat_generate(d, scale, 1, int(weighted), 0, m, None))
    t0 = time.perf_counter()
    G = gt.Graph(weighted=weighted)
    if app == "bfs":
        G.load_device(d.value, m, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = gt.BFS_Program(G, False, False, True, gt._ROW_)
    elif app == "cc":
        G.load_device(d.value, m, nv, nv, False, False, True, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = gt.CC_Program(G, False, True, False, gt._ROW_)
    elif app == "sssp":
        G.load_device(d.value, m, nv, nv, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = gt.SSSP_Program(G, False, True, False, gt._ROW_)
    else:
        G.load_device(d.value, m, nv, nv, True, False, True, False, True, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
        P = gt.Deg_Program(G, True, False, False, gt._ROW_)
    _lib.check(L.gt_free(d)); _lib.check(L.gt_device_synchronize())
    ingress = time.perf_counter() - t0
    P.root = args.root if args.root is not None else 0   # vertex 0 is the largest R-MAT hub
    P.execute(1 if app == "deg" else 0)
    st = P.stats
    cs = P.checksum(out=None)
    P.initialize(); P.execute(1 if app == "deg" else 0)   # the same run again: the first one also pays for cold caches and page tables
    warm = P.stats.seconds
    assert P.checksum(out=None) == cs and P.stats.iterations == st.iterations
    extra = {}
    try:   # the hybrid passes of the two timed runs (engine-internal diagnostic, csrc/pb.hip)
        raw = C.CDLL(_lib.LIB_PATH); raw.gt_graph_hybrid_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        hs = (C.c_uint64 * 4)(); _lib.check(raw.gt_graph_hybrid_stats(G._h, hs, 1))
        extra_h = {"hybrid": {"passes": int(hs[0]) // 2, "entries_to_column_kernels": int(hs[1]) // 2, "entries_left_out_of_the_stream": int(hs[2]) // 2, "windows_left_out": int(hs[3]) // 2}}
    except Exception:
        extra_h = {}
    if app != "deg":
        req, per_it = required_bytes(L, G, P, app)
        assert len(per_it) == st.iterations, (len(per_it), st.iterations)
        extra = {"required_bytes": req, "required_GBps": req / st.seconds / 1e9, "required_frac": req / st.seconds / 8e12, "required_frac_warm": req / warm / 8e12, "per_iteration": per_it}
    extra.update(extra_h)
    print(json.dumps({"app": app, "scale": scale, "edge_factor": args.edge_factor, "root": int(P.root), "spmv": os.environ.get("GRAPHTAP_SPMV", "pb"), "stored_entries": int(G.info.nnz_local),
                      "iterations": st.iterations, "sparse_iterations": int(st.spmspv_iterations), "execute_s": st.seconds, "GTEPS": G.info.nnz_local * st.iterations / st.seconds / 1e9,
                      "execute_warm_s": warm, "GTEPS_warm": G.info.nnz_local * st.iterations / warm / 1e9, "list_iterations": int(st.list_iterations),
                      "spmv_ms_mean": st.spmv_ms / max(st.spmv_launches, 1), "ingress_s": round(ingress, 3),
                      "value_checksum": cs[0], "reachable": cs[1], **extra}), flush=True)
    P.free(); G.free()
