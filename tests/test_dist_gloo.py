"""world_size-2 (and 3) `gloo` runs of the multi-GPU iteration driver graphtap_amd/dist.py on CPU.

The driver is the code that runs on the GPU box with backend "nccl" (= RCCL); here the tile engine it
drives is a small numpy model of one rank's tile-row built from the CPU oracle's TCSC arrays (test
infrastructure), so the partition arithmetic (H = nrows/p + 1, the needed-columns exchange plan with its K
slices and padded blocks), the all-to-all pattern and the convergence all-reduce are exercised across real processes.
Results must equal the reference's golden vectors: integer programs bit for bit, PageRank to 1e-6."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_case

INF = 2147483647


class NumpyTileEngine:
    """One rank's tile-row, same five phase methods as vertex_program._HipEngine."""

    def __init__(self, kind, rank, nranks, edges, nv, root=0, order_col=False, x_slices=2):
        from oracle import oracle as O
        self.kind, self.rank, self.nranks, self.root = kind, rank, nranks, root
        app = {"deg": "pr", "pr": "pr", "bfs": "bfs", "sssp": "sssp", "cc": "cc"}[kind]
        g = O.OracleGraph(edges, nv, weighted=(kind == "sssp"), **O.APP_FLAGS[app])
        JA, IA, JC, IR = g.JA.astype(np.int64), g.IA.astype(np.int64), g.JC.astype(np.int64), g.IR.astype(np.int64)
        col_of_entry = np.repeat(np.arange(g.nnzcols), np.diff(JA))
        rows_v, cols_v = IR[IA], JC[col_of_entry]
        w = g.A.astype(np.int64) if kind == "sssp" else None
        n = nv + 1
        p = nranks
        self.H = H = n // p + 1                                  # mat/matrix.hpp:193
        span = p * H
        rowflag = np.zeros(span, bool); colflag = np.zeros(span, bool)
        rowflag[rows_v] = True; colflag[cols_v] = True
        Srow = np.concatenate([[0], np.cumsum(rowflag)]); Scol = np.concatenate([[0], np.cumsum(colflag)])
        nnzcols_seg = [Scol[(s + 1) * H] - Scol[s * H] for s in range(p)]
        # exchange plan (include/graphtap_amd.h, gt_exchange_plan): slice of compressed column j = j // T; rank r's x
        # holds, slice after slice and source segment after source segment, the columns tile-row r has an entry in
        self.x_slices = K = x_slices
        self.S = S = max(max(nnzcols_seg), 1)
        self.T = T = -(-S // K)
        PAD, ALIGN = 4, 8          # block sizes round up to 4 elements, slice starts to a "window" (8 here)
        owner = rows_v // H
        cseg_all = cols_v // H
        j_all = Scol[cols_v] - Scol[cseg_all * H]

        def need(r, s):            # sorted compressed columns of segment s that tile-row r reads
            return np.unique(j_all[(owner == r) & (cseg_all == s)])
        up = lambda n, a: -(-n // a) * a
        lo = rank * H
        mine = owner == rank
        self.r = (Srow[rows_v[mine]] - Srow[lo]).astype(np.int64)            # local compressed row
        cseg, j = cseg_all[mine], j_all[mine]
        self.c = np.zeros(j.size, np.int64)                                   # local column of every entry
        self.recv_off, self.send_off, self.recv_counts, self.send_counts = [], [], [], []
        xo = so = 0
        send_idx = []
        for k in range(K):
            self.recv_off.append(xo); self.send_off.append(so)
            rc, sc = [], []
            for s_ in range(p):
                nd = need(rank, s_); nd = nd[nd // T == k]
                sel = (cseg == s_) & (j // T == k)
                self.c[sel] = xo + np.searchsorted(nd, j[sel])
                rc.append(up(nd.size, PAD)); xo += rc[-1]
                out = need(s_, rank); out = out[out // T == k]               # what destination s_ reads of MY columns
                sc.append(up(out.size, PAD)); so += sc[-1]
                send_idx += list(out) + [0] * (sc[-1] - out.size)
            self.recv_counts.append(rc); self.send_counts.append(sc)
            xo = up(xo, ALIGN)
        self.recv_off.append(max(xo, ALIGN)); self.send_off.append(so)
        self.send_idx = np.array(send_idx, np.int64)
        self.cg = (cseg * S + j).astype(np.int64)                             # global [segment][S] column (Degree, _COL_)
        self.w = w[mine] if w is not None else None
        self.nnz_local = int(mine.sum())
        self.I = rowflag[lo:lo + H]; self.J = colflag[lo:lo + H]
        self.IR = np.nonzero(self.I)[0]; self.JC = np.nonzero(self.J)[0]
        self.nr, self.nc = self.IR.size, self.JC.size
        self.column_accumulators = order_col
        self.needs_x_exchange = not order_col
        self.iteration = 0
        self.converged = False
        fp = np.float64 if kind == "pr" else np.int32
        self.x = np.zeros(self.recv_off[-1], fp)
        self.send = np.zeros(max(so, 1), fp)[:so]
        self.xseg = np.zeros(max(self.nc, 1), fp)
        self.y = np.zeros(p * S if order_col else self.nr, fp)
        vid = lo + np.arange(H)
        if kind == "deg":
            self.degree = np.zeros(H, np.int64); self.C = np.ones(H, bool)
        elif kind == "pr":
            self.degree = np.zeros(H, np.int64); self.rank_ = np.full(H, 0.15); self.C = np.ones(H, bool)
        elif kind == "bfs":
            self.parent = np.where(vid == root, vid, 0); self.hops = np.where(vid == root, 0, INF); self.C = vid == root
            self.y[:] = INF
        elif kind == "sssp":
            self.s0 = np.where(vid == root, 0, INF); self.C = vid == root; self.y[:] = INF
        else:
            self.s0 = vid.copy(); self.C = np.ones(H, bool); self.y[:] = INF
        self.vid = vid

    def x_tensor(self): return torch.from_numpy(self.x)
    def y_tensor(self): return torch.from_numpy(self.y)
    def send_tensor(self): return torch.from_numpy(self.send)
    def exchange_plan(self): return self.send_off, self.recv_off, self.send_counts, self.recv_counts

    def scatter_gather(self):
        if self.column_accumulators or self.nc == 0: return
        v = self.JC
        if self.kind == "deg": self.xseg[:] = 1
        elif self.kind == "pr": self.xseg[:] = np.where(self.degree[v] > 0, self.rank_[v] / np.maximum(self.degree[v], 1), 0.0)
        elif self.kind == "bfs": self.xseg[:] = np.where(self.C[v], self.vid[v], INF)
        else: self.xseg[:] = np.where(self.C[v], self.s0[v], INF)
        self.send[:] = self.xseg[self.send_idx]

    def combine(self):
        for k in range(self.x_slices): self.combine_slice(k)

    def combine_slice(self, k):
        """entries whose column lies in slice k of the message vector (only that slice of x is guaranteed current)"""
        if self.converged: return
        if self.column_accumulators:
            if k == self.x_slices - 1: self.y[:] = np.bincount(self.cg, minlength=self.y.size)
            return
        sel = (self.c >= self.recv_off[k]) & (self.c < self.recv_off[k + 1])
        if self.kind in ("deg", "pr"):
            if k == 0: self.y[:] = 0
            np.add.at(self.y, self.r[sel], self.x[self.c[sel]])
        else:
            xv = self.x[self.c[sel]].astype(np.int64)
            ok = xv != INF
            cand = xv + (self.w[sel] if self.w is not None else 0)
            y = self.y.astype(np.int64)
            np.minimum.at(y, self.r[sel][ok], cand[ok])
            self.y[:] = y

    def apply(self, iters, want_active):
        if self.converged: return 0
        v = self.IR
        if self.iteration == 0: self.C[~self.I] = False
        if self.kind == "deg":
            if self.column_accumulators:
                self.degree[self.JC] = self.y[self.rank * self.S + np.arange(self.nc)]; self.C[self.JC] = False
            else:
                self.degree[v] = self.y; self.C[v] = False
        elif self.kind == "pr":
            new = 0.15 + 0.85 * self.y
            self.C[v] = np.abs(new - self.rank_[v]) > 1e-5
            self.rank_[v] = new
        elif self.kind == "bfs":
            hit = (self.hops[v] == INF) & (self.y != INF)
            self.hops[v[hit]] = self.iteration + 1; self.parent[v[hit]] = self.y[hit]; self.C[v] = hit
        else:
            new = np.minimum(self.y, self.s0[v]); self.C[v] = new != self.s0[v]; self.s0[v] = new
        self.iteration += 1
        return int(self.C[v].sum())

    def finish_converged(self): self.converged = True


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphtap_amd import dist as gdist
    c = load_case(case); nv = c["num_vertices"]
    res = {}
    # apps/pr.cpp: Deg in _COL_ order (column-space accumulators are all-reduced), then PageRank
    d = NumpyTileEngine("deg", rank, world, c["edges"], nv, order_col=True); gdist.run(d, 1)
    pr = NumpyTileEngine("pr", rank, world, c["edges"], nv); pr.degree = np.where(pr.I, d.degree, 0)
    res["pr_iters"], _ = gdist.run(pr, 20)
    res["pr_rank"], res["pr_degree"] = pr.rank_, pr.degree
    # apps/deg.cpp: _ROW_ order on the untransposed graph is covered on the GPU; here BFS / SSSP / CC to convergence
    b = NumpyTileEngine("bfs", rank, world, c["edges"], nv, root=c["root"]); res["bfs_iters"], conv = gdist.run(b, 0); assert conv
    res["parent"], res["hops"] = b.parent, b.hops
    s = NumpyTileEngine("sssp", rank, world, c["wedges"], nv, root=c["root"]); res["sssp_iters"], _ = gdist.run(s, 0)
    res["distance"] = s.s0
    cc = NumpyTileEngine("cc", rank, world, c["edges"], nv); res["cc_iters"], _ = gdist.run(cc, 0)
    res["label"] = cc.s0
    res["nnz_local"] = pr.nnz_local
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        torch.save(gathered, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("rmat10", 2), ("tiny", 2), ("rmat12", 3)])
def test_driver_over_gloo_matches_reference(tmp_path, case, world, known_answers):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
    parts = torch.load(out, weights_only=False)
    c = load_case(case); n = c["num_vertices"] + 1
    cat = lambda k: np.concatenate([p[k] for p in parts])[:n]
    assert (cat("pr_degree") == c["np1_pr20_a"]).all()
    ref = c["np1_pr20_c"]
    assert (np.abs(cat("pr_rank") - ref) / ref).max() < 1e-6
    assert (cat("parent") == c["np1_bfs_a"]).all() and (cat("hops") == c["np1_bfs_b"]).all()
    assert (cat("distance") == c["np1_sssp_a"]).all()
    assert (cat("label") == c["np1_cc_a"]).all()
    ka = known_answers[case]
    assert parts[0]["bfs_iters"] == ka["np1_bfs"]["iterations"]
    assert parts[0]["sssp_iters"] == ka["np1_sssp"]["iterations"]
    assert parts[0]["cc_iters"] == ka["np1_cc"]["iterations"]
    assert sum(p["nnz_local"] for p in parts) == len(c["edges"])   # PR keeps every record (pr.cpp:28-30)


def _plan_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from graphtap_amd import dist as gdist
    # K = 2 slices, p = 2: rank 0 sends 4/8 (slice 0/1) to rank 1, rank 1 sends 12/0 to rank 0; own blocks 4/4
    send = {0: [[4, 4], [4, 8]], 1: [[12, 4], [0, 4]]}[rank]
    recv = {0: [[4, 12], [4, 0]], 1: [[4, 4], [8, 4]]}[rank]
    gdist.verify_plan(([0, 8, 20], [0, 16, 24], send, recv), "cpu")            # consistent: passes
    bad = [row[:] for row in recv]
    if rank == 1: bad[1][0] = 4                                                  # rank 1 expects 4 where rank 0 sends 8
    try:
        gdist.verify_plan(([0, 8, 20], [0, 16, 24], send, bad), "cpu"); raised = False
    except RuntimeError as e:
        raised = "exchange plan mismatch" in str(e)
    flags = [None] * world
    dist.all_gather_object(flags, raised)
    if rank == 0: torch.save(flags, out)
    dist.barrier(); dist.destroy_process_group()


def test_exchange_plan_mismatch_is_reported(tmp_path):
    out = str(tmp_path / "flags.pt")
    mp.spawn(_plan_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out, weights_only=False) == [False, True]
