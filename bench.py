#!/usr/bin/env python3
"""bench.py -- PageRank GTEPS on synthetic R-MAT (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scale S] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A STEP is one PageRank iteration of the hot path (scatter_gather -> combine/SpMV -> apply,
/root/reference/src/vp/vertex_program.hpp:417-421) over the whole graph. The workload is the one
the metric is quoted on: PageRank, R-MAT scale 26, edge factor 16, (a,b,c,d)=(.57,.19,.19,.05),
seed 1, flags of apps/pr.cpp (directed, transposed, self loops and parallel edges
kept, TCSC_CF); the reference's `fp` is double (apps/deg.h:19) and so is ours ("dtype": "f64").
For N > 1 the SAME graph is split by tile-rows over the N GPUs (strong scaling); the only
collective on the data path is the all-gather of the message vector x (graphtap_amd/dist.py).

Timed region: exactly K steps, bracketed by barrier + device synchronize on both sides, MAX over
ranks. The edge list is generated in HBM and the TCSC build runs on the device before the timed
region (inputs resident, like the reference's "Execute time" which excludes ingress).

GTEPS = stored entries x K / t / 1e9. roofline.achieved = algorithmic bytes of one
SpMV launch / mean SpMV kernel duration measured with HIP events on the launch stream;
B_alg = 4 nnz + 4 (nnzcols+1) + F nnzcols + F nnzrows  (BASELINE.md section 3), F = 8.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(scale, seed, budget_iters=3):
    """Oracle ("port") PageRank on a bounded sample of the same workload, timed like the reference's
    Execute time (iteration loop only), single thread."""
    import numpy as np
    from graphtap_amd.rmat import rmat_edges
    from oracle import oracle as O
    sample_edges = min(16 << scale, 1 << 26)   # first 2^26 records of the same R-MAT stream
    nv = 1 << scale
    chunks = []
    step = 1 << 22
    for first in range(0, sample_edges, step):
        chunks.append(rmat_edges(scale, 16, seed, first=first, count=min(step, sample_edges - first)))
    e = np.concatenate(chunks)
    g = O.OracleGraph(e, nv, **O.APP_FLAGS["pr"])
    d = g.degree(1)
    t0 = time.perf_counter()
    _, _, it = g.pagerank(d, budget_iters, cf=True)
    dt = time.perf_counter() - t0
    nnz = g.nnz
    g.close()
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": nnz * it / dt / 1e9, "unit": "GTEPS", "cores": 1, "kind": "port",
            "sample": "oracle/gt_oracle.c PageRank, first %d records of the same R-MAT-%d stream (seed %d), %d iterations, %.1f s, %s"
                      % (sample_edges, scale, seed, it, dt, model)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # the reference's PageRank runs are 20 iterations (graphtap.slurm:72)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on a 1-GPU box)")
    ap.add_argument("--spmv", choices=["pb", "pb_f32msg", "edge"], default=os.environ.get("GRAPHTAP_SPMV", "pb"),
                    help="SpMV implementation (gt_spmv_variant): propagation blocking (default), the same with f32 messages, or the edge-atomic baseline")
    args = ap.parse_args()

    os.environ["GRAPHTAP_SPMV"] = args.spmv
    import torch
    import torch.distributed as dist
    import graphtap_amd as gt
    from graphtap_amd import _lib
    L = _lib.lib()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    _lib.require_gpu()
    if os.environ.get("GRAPHTAP_SHARE_GPU"):   # rehearsal of the N-rank path on a 1-GPU box: all ranks on device 0
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    _lib.check(L.gt_set_device(local))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic input, generated in HBM; device-side TCSC build (untimed, like the reference's ingress)
    scale, nv = args.scale, 1 << args.scale
    m = 16 << scale
    t_in0 = time.perf_counter()
    d = C.c_void_p()
    _lib.check(L.gt_malloc(C.byref(d), m * 8))
    _lib.check(L.gt_rmat_generate(d, scale, args.seed, 0, 0, m, None))
    G = gt.Graph()
    G.load_device(d.value, m, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=rank, nranks=world)  # apps/pr.cpp:26-36
    _lib.check(L.gt_free(d))
    V = gt.Deg_Program(G, True, False, False, gt._COL_)   # apps/pr.cpp:37-42
    V.execute(1)
    VR = gt.PR_Program(G, True, False, False, gt._ROW_)   # apps/pr.cpp:46-50
    VR.initialize(V)
    V.free()
    barrier()
    t_ingress = time.perf_counter() - t_in0

    h = VR._handle()
    # ---- warmup, then exactly K timed steps
    done = 0
    if args.warmup:
        VR.execute(done + args.warmup); done += args.warmup
    _lib.check(L.gt_program_enable_timing(h, 1))
    ms0, n0 = C.c_double(), C.c_uint32()
    _lib.check(L.gt_program_timing(h, C.byref(ms0), C.byref(n0), 1))
    barrier()
    t0 = time.perf_counter()
    VR.execute(done + args.steps); done += args.steps
    barrier()
    dt = time.perf_counter() - t0
    if world == 1:
        spmv_ms, launches = VR.stats.spmv_ms, VR.stats.spmv_launches
    else:
        a, b = C.c_double(), C.c_uint32()
        _lib.check(L.gt_program_timing(h, C.byref(a), C.byref(b), 1))
        spmv_ms, launches = a.value, b.value
    assert launches == args.steps, (launches, args.steps)

    i = G.info
    F = 8
    b_alg = 4 * i.nnz_local + 4 * (i.nranks * i.seg_stride + 1) + F * i.nranks * i.seg_stride + F * i.nnzrows
    if world == 1:
        b_alg = 4 * i.nnz_local + 4 * (i.nnzcols + 1) + F * i.nnzcols + F * i.nnzrows
    kernel_ms = spmv_ms / launches
    t = torch.tensor([dt, kernel_ms, float(b_alg)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, kernel_ms, b_alg = float(t[0]), float(t[1]), float(t[2])
    nnz = G.nnz_global
    value = nnz * args.steps / dt / 1e9
    achieved = b_alg / (kernel_ms * 1e-3) / 1e9     # GB/s of the slowest rank's SpMV launch

    checksum = VR.checksum(out=None)
    traffic = None
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written by profiles/collect_pmc.py from rocprofv3 --pmc passes
    if os.path.exists(tp):
        try:
            rec = json.load(open(tp))
            key = "scale%d_gpus%d" % (scale, world)
            traffic = rec.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "PageRank GTEPS on RMAT-%d" % scale, "value": value, "unit": "GTEPS", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64" if args.spmv != "pb_f32msg" else "f64 accumulate / f32 messages", "data": "synthetic",
        "config": {"workload": "PageRank R-MAT scale %d edge-factor 16 seed %d, flags of apps/pr.cpp (TCSC_CF), 1 step = 1 iteration" % (scale, args.seed),
                   "num_vertices": nv, "edge_records": m, "stored_entries": nnz, "nnzrows": int(i.nnzrows_global), "nnzcols": int(i.nnzcols_global),
                   "spmv": args.spmv, "partition": "tile-rows x%d (1-D), all-gather of x per step" % world if world > 1 else "single tile",
                   "ingress_s": round(t_ingress, 3), "iterations_total": VR.iteration, "value_checksum": checksum[0], "reachable": checksum[1]},
        "roofline": {"bound": "hbm", "kernel": {"pb": "k_pb_scatter<double,double> + k_pb_gather<double,double> (one SpMV = this launch pair)",
                                                  "pb_f32msg": "k_pb_scatter<double,float> + k_pb_gather<double,float> (one SpMV = this launch pair)",
                                                  "edge": "k_spmv_edge<GT_PLUS_F64>"}[args.spmv], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes_per_launch": b_alg,
                     "kernel_ms": kernel_ms, "launches": launches},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scale, args.seed)
    VR.free(); G.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
