#!/usr/bin/env python3
"""Per-iteration times of the C++ multi-rank driver for the min programs (BFS / SSSP / CC, converge mode).

  --transport rccl      world size 1 over RCCL with the exchange layout forced on (GRAPHTAP_FORCE_EXCHANGE): every grouped
                        send/recv round, the all-reduce of the convergence word + pair counts and the host round trip are the
                        real ones of an N-GPU run, with itself as the only peer (what a 1-GPU box can time)
  --transport loopback  p ranks of one process (one host thread each, device copies): the per-rank kernels at the sizes of a
                        p-way split; its host barriers are NOT representative of RCCL latencies

  python tools/bench_dist_iters.py --scale 22 --apps bfs --transport rccl
  python tools/bench_dist_iters.py --scale 22 --nranks 8 --apps bfs --transport loopback
Prints one JSON line per app: iterations, wall ms, and per iteration [pack, first slice, last slice, SpMV span, apply, rest, mode]."""
import argparse, ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22); ap.add_argument("--nranks", type=int, default=8)
ap.add_argument("--apps", default="bfs,sssp,cc"); ap.add_argument("--root", type=int, default=0)
ap.add_argument("--transport", choices=["rccl", "loopback"], default="rccl")
a = ap.parse_args()
if a.transport == "rccl":
    os.environ["GRAPHTAP_FORCE_EXCHANGE"] = "1"
import torch  # noqa: F401
import numpy as np
import graphtap_amd as gt
from graphtap_amd import _lib, dist_native
from graphtap_amd.rmat import rmat_edges

L = _lib.lib(); _lib.require_gpu(); _lib.check(L.gt_set_device(0))
nv = 1 << a.scale
w = rmat_edges(a.scale, 16, 1, weighted=True); e = np.ascontiguousarray(w[:, :2])
if a.transport == "rccl":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dist_native.init(0, 1)
    p = 1; dists = [dist_native.handle()]
else:
    p = a.nranks
    hs = (C.c_void_p * p)(); _lib.check(L.gt_dist_create_loopback(hs, p)); dists = [C.c_void_p(hs[r]) for r in range(p)]


def build(app, r):
    if app == "bfs":
        G = gt.Graph(); G.load_edges(e, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=p); P = gt.BFS_Program(G, False, False, True, gt._ROW_)
    elif app == "cc":
        G = gt.Graph(); G.load_edges(e, nv, nv, False, False, True, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=p); P = gt.CC_Program(G, False, True, False, gt._ROW_)
    else:
        G = gt.Graph(weighted=True); G.load_edges(w, nv, nv, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=p); P = gt.SSSP_Program(G, False, True, False, gt._ROW_)
    P.root = a.root; P.initialize()
    return G, P


def execute_all(progs):
    out = [None] * p
    def work(r):
        st = _lib.ExecStats(); L.gt_set_device(0)
        out[r] = (L.gt_dist_execute(dists[r], progs[r]._handle(), 0, C.byref(st)), st.iterations, st.seconds, st.list_iterations, st.spmspv_iterations)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(p)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert all(o[0] == 0 for o in out), L.gt_last_error()
    return out


for app in a.apps.split(","):
    for proto in ("lists", "dense"):
        if proto == "dense": os.environ["GRAPHTAP_DIST_PROTOCOL"] = "dense"
        else: os.environ.pop("GRAPHTAP_DIST_PROTOCOL", None)
        built = [build(app, r) for r in range(p)]
        ps = [b[1] for b in built]
        execute_all(ps)                      # warm (code objects, allocations)
        for P in ps: P.initialize()
        out = execute_all(ps)
        d0 = dist_native.diagnostics(dists[0])
        print(json.dumps({"app": app, "scale": a.scale, "transport": a.transport, "ranks": p, "protocol": proto, "iterations": out[0][1],
                          "execute_ms": round(max(o[2] for o in out) * 1e3, 3), "list_iterations": out[0][3], "spmspv_iterations_rank0": out[0][4],
                          "host_round_trips_rank0": d0["host_round_trips"], "bytes_sent_rank0": d0["bytes_sent"],
                          "fields": list(dist_native.TIME_FIELDS), "per_iteration_rank0": d0["per_iteration"]}), flush=True)
        for G, P in built: P.free(); G.free()
if a.transport == "rccl":
    dist_native.free(); dist.destroy_process_group()
else:
    for d in dists: L.gt_dist_free(d)
