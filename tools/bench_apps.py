#!/usr/bin/env python3
"""Times the four non-headline programs (BFS, CC, SSSP, Degree) on synthetic R-MAT on one GPU.
GTEPS = stored entries x iterations / Execute time. Not a bench.py line: the BASELINE configs
name these as parity cases; this script gives the numbers quoted in DESIGN.md.
  python tools/bench_apps.py --scale 24 [--apps bfs,cc,sssp]"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphtap_amd as gt
from graphtap_amd import _lib

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
    print(json.dumps({"app": app, "scale": scale, "edge_factor": args.edge_factor, "root": int(P.root), "spmv": os.environ.get("GRAPHTAP_SPMV", "pb"), "stored_entries": int(G.info.nnz_local),
                      "iterations": st.iterations, "sparse_iterations": int(st.spmspv_iterations), "execute_s": st.seconds, "GTEPS": G.info.nnz_local * st.iterations / st.seconds / 1e9,
                      "spmv_ms_mean": st.spmv_ms / max(st.spmv_launches, 1), "ingress_s": round(ingress, 3),
                      "value_checksum": cs[0], "reachable": cs[1]}), flush=True)
    P.free(); G.free()
