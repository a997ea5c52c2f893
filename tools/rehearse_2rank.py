#!/usr/bin/env python3
"""2-process rehearsal of the multi-rank apps on ONE GPU (backend gloo, host-staged collectives): PageRank, BFS, CC and
SSSP through graphtap_amd.dist, results assembled with gather_global() and compared with a 1-rank run.
  GRAPHTAP_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/rehearse_2rank.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import graphtap_amd as gt
from graphtap_amd import _lib
from graphtap_amd.rmat import rmat_edges

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); _lib.check(_lib.lib().gt_set_device(0))
dist.init_process_group("gloo", rank=rank, world_size=world)
scale, nv = 16, 1 << 16
w = rmat_edges(scale, 16, 9, weighted=True); e = np.ascontiguousarray(w[:, :2])

def run(nranks, r):
    out = {}
    G = gt.Graph(); G.load_edges(e, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=r, nranks=nranks)
    V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
    P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V); P.execute(20)
    out["pr"] = P.gather_global(); out["pr_disp"] = P.display(out=None)[:3]; P.free(); V.free(); G.free()
    G = gt.Graph(); G.load_edges(e, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks)
    P = gt.BFS_Program(G, False, False, True, gt._ROW_); P.root = 5; P.execute(); out["bfs"] = P.gather_global(); out["bfs_it"] = P.iteration
    out["bfs_cs"] = P.checksum(out=None); P.free(); G.free()
    G = gt.Graph(); G.load_edges(e, nv, nv, False, False, True, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks)
    P = gt.CC_Program(G, False, True, False, gt._ROW_); P.execute(); out["cc"] = P.gather_global(); out["cc_it"] = P.iteration; P.free(); G.free()
    G = gt.Graph(weighted=True); G.load_edges(w, nv, nv, True, True, False, False, False, gt._2DT_, gt._TCSC_, rank=r, nranks=nranks)
    P = gt.SSSP_Program(G, False, True, False, gt._ROW_); P.root = 5; P.execute(); out["sssp"] = P.gather_global(); out["sssp_it"] = P.iteration; P.free(); G.free()
    return out

multi = run(world, rank)
if rank == 0:
    single = run(1, 0)
    assert (multi["bfs"]["parent"] == single["bfs"]["parent"]).all() and (multi["bfs"]["hops"] == single["bfs"]["hops"]).all()
    assert multi["bfs_it"] == single["bfs_it"] and multi["bfs_cs"] == single["bfs_cs"]
    assert (multi["cc"]["label"] == single["cc"]["label"]).all() and multi["cc_it"] == single["cc_it"]
    assert (multi["sssp"]["distance"] == single["sssp"]["distance"]).all() and multi["sssp_it"] == single["sssp_it"]
    assert (multi["pr"]["degree"] == single["pr"]["degree"]).all()
    rel = np.abs(multi["pr"]["rank"] - single["pr"]["rank"]) / single["pr"]["rank"]
    assert rel.max() < 1e-6 and multi["pr_disp"] == single["pr_disp"], (rel.max(), multi["pr_disp"], single["pr_disp"])
    print("%d-rank rehearsal ok: BFS/CC/SSSP bit-exact, PageRank max rel err %.2e, display %s" % (world, rel.max(), multi["pr_disp"][0]))
dist.barrier(); dist.destroy_process_group()
