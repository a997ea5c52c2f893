#!/usr/bin/env python3
"""Bytes the exchange of x moves per iteration with and without the sparse (index, value) form, for BFS / SSSP / CC over p
tile-rows: all p ranks of the C++ driver (gt_dist_execute) in THIS process over the loopback transport (one host thread per
rank, one GPU) -- the volumes are exactly what p GPUs would send over RCCL.
  python tools/bench_exchange.py --scale 22 --nranks 8 [--apps bfs,sssp,cc]"""
import argparse, ctypes as C, json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (before the library: see tests/conftest.py)
import numpy as np
import graphtap_amd as gt
from graphtap_amd import _lib
from graphtap_amd.rmat import rmat_edges

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22); ap.add_argument("--nranks", type=int, default=8)
ap.add_argument("--apps", default="bfs,sssp,cc"); ap.add_argument("--root", type=int, default=0)
a = ap.parse_args()
L = _lib.lib(); _lib.require_gpu(); _lib.check(L.gt_set_device(0))
nv, p = 1 << a.scale, a.nranks
w = rmat_edges(a.scale, 16, 1, weighted=True); e = np.ascontiguousarray(w[:, :2])
hs = (C.c_void_p * p)(); _lib.check(L.gt_dist_create_loopback(hs, p)); dists = [C.c_void_p(hs[r]) for r in range(p)]


def execute_all(progs):
    out = [None] * p
    def work(r):
        st = _lib.ExecStats(); L.gt_set_device(0)
        out[r] = (L.gt_dist_execute(dists[r], progs[r]._handle(), 0, C.byref(st)), st.iterations, st.seconds)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(p)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert all(o[0] == 0 for o in out), L.gt_last_error()
    return out[0][1]


for app in a.apps.split(","):
    for sparse in ("1", "0"):
        os.environ["GRAPHTAP_SPARSE_EXCHANGE"] = sparse
        Gs, ps = [], []
        for r in range(p):
            if app == "bfs":
                G = gt.Graph(); G.load_edges(e, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=p); P = gt.BFS_Program(G, False, False, True, gt._ROW_)
            elif app == "cc":
                G = gt.Graph(); G.load_edges(e, nv, nv, False, False, True, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=p); P = gt.CC_Program(G, False, True, False, gt._ROW_)
            else:
                G = gt.Graph(weighted=True); G.load_edges(w, nv, nv, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=p); P = gt.SSSP_Program(G, False, True, False, gt._ROW_)
            P.root = a.root; P.initialize(); Gs.append(G); ps.append(P)
        for d in dists: _lib.check(L.gt_dist_exchange_stats(d, None, None, None, 1))
        it = execute_all(ps)
        sent = dense = 0
        s_, d_, n_ = C.c_uint64(), C.c_uint64(), C.c_uint64()
        for d in dists:
            _lib.check(L.gt_dist_exchange_stats(d, C.byref(s_), C.byref(d_), C.byref(n_), 1)); sent += s_.value; dense += d_.value
        print(json.dumps({"app": app, "scale": a.scale, "nranks": p, "sparse_exchange": sparse == "1", "iterations": it,
                          "bytes_sent_per_iteration_all_ranks": sent // max(it, 1), "dense_bytes_per_iteration_all_ranks": dense // max(it, 1),
                          "ratio": round(sent / max(dense, 1), 4)}), flush=True)
        for P in ps: P.free()
        for G in Gs: G.free()
for d in dists: L.gt_dist_free(d)
