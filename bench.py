#!/usr/bin/env python3
"""bench.py -- PageRank GTEPS on synthetic R-MAT (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scale S] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A STEP is one PageRank iteration of the hot path (scatter_gather -> combine/SpMV -> apply,
/root/reference/src/vp/vertex_program.hpp:417-421) over the whole graph. The workload is the one
the metric is quoted on: PageRank, R-MAT scale 26, edge factor 16, (a,b,c,d)=(.57,.19,.19,.05),
seed 1, flags of apps/pr.cpp (directed, transposed, self loops and parallel edges
kept, TCSC_CF). The reference's `fp` is double (apps/deg.h:19): ranks, accumulators and y are f64
here too; by default the per-edge messages are rounded to f32 in flight (--spmv pb_f32msg, max
relative rank error 3.7e-8 against the fp64 oracle, tests/test_gpu_parity.py; BASELINE.json quotes
the GPU configuration as fp32 with a 1e-6 tolerance). --spmv pb keeps the messages in f64.
For N > 1 the SAME graph is split by tile-rows over the N GPUs (strong scaling); the only
collective on the data path is the all-to-all of the needed columns of the message vector x (graphtap_amd/dist.py).

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


def _cpu_model():
    try:
        return [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        return "unknown"


def _sample_edges(scale, seed, count):
    """The first `count` records of the workload's R-MAT stream, generated on the GPU (bit-identical to
    graphtap_amd/rmat.py) and copied to the host."""
    import numpy as np
    from graphtap_amd import _lib
    L = _lib.lib()
    d = C.c_void_p()
    _lib.check(L.gt_malloc(C.byref(d), count * 8))
    _lib.check(L.gt_rmat_generate(d, scale, seed, 0, 0, count, None))
    e = np.empty((count, 2), np.uint32)
    _lib.check(L.gt_memcpy_d2h(e.ctypes.data_as(C.c_void_p), d, count * 8))
    _lib.check(L.gt_free(d))
    return e


def cpu_baseline_port(e, scale, seed, iters=3):
    """Oracle ("port") PageRank, timed like the reference's Execute time (iteration loop only), one thread."""
    from oracle import oracle as O
    g = O.OracleGraph(e, 1 << scale, **O.APP_FLAGS["pr"])
    d = g.degree(1)
    t0 = time.perf_counter()
    _, _, it = g.pagerank(d, iters, cf=True)
    dt = time.perf_counter() - t0
    nnz = g.nnz
    g.close()
    return {"value": nnz * it / dt / 1e9, "unit": "GTEPS", "cores": 1, "kind": "port",
            "sample": "oracle/gt_oracle.c PageRank, first %d records of the same R-MAT-%d stream (seed %d), %d iterations, %.1f s, %s"
                      % (len(e), scale, seed, it, dt, _cpu_model())}


def cpu_baseline_reference(e, scale, seed, iters=20, nproc=8):
    """The UNMODIFIED reference (oracle/_ref/pr, built by oracle/ref/Makefile in the build container) run
    under the image's MPICH on this box's host cores; its own "Execute time" line is the timer."""
    import re
    import subprocess
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref")
    exe, mpirun = os.path.join(ref, "pr"), "/opt/conda/bin/mpirun"
    if not (os.path.exists(exe) and os.path.exists(mpirun)):
        return None
    with tempfile.TemporaryDirectory(prefix="gtref") as tmp:
        path = os.path.join(tmp, "sample.bin")
        e.tofile(path)
        env = dict(os.environ, PATH=os.path.join(ref, "fileshim") + ":" + os.environ.get("PATH", ""))
        try:
            t0 = time.perf_counter()
            r = subprocess.run([mpirun, "-np", str(nproc), exe, path, str(1 << scale), str(iters)], env=env,
                               capture_output=True, text=True, timeout=150)
            wall = time.perf_counter() - t0
        except Exception:
            return None
    ex = re.findall(r"^Execute time: ([0-9.eE+-]+) seconds", r.stdout, re.M)
    its = re.findall(r"^Iterations: (\d+)", r.stdout, re.M)
    ing = re.findall(r"^Ingress time: ([0-9.eE+-]+) seconds", r.stdout, re.M)
    if r.returncode != 0 or len(ex) < 2 or not its or int(its[-1]) != iters:
        return None
    t = float(ex[-1])   # the PageRank pass (the first Execute line is the Degree pass)
    return {"value": len(e) * iters / t / 1e9, "unit": "GTEPS", "cores": nproc, "kind": "reference",
            "sample": "unmodified GraphTap `pr` (mpirun -np %d), first %d records of the same R-MAT-%d stream (seed %d), %d iterations, "
                      "Execute %.2f s, Ingress %s s, wall %.0f s, %s" % (nproc, len(e), scale, seed, iters, t, ing[-1] if ing else "?", wall, _cpu_model())}


def cpu_baseline(scale, seed):
    sample = min(16 << scale, 1 << 27)   # bounded sample: the first 2^27 records (1 GiB) of the same stream
    e = _sample_edges(scale, seed, sample)
    port = cpu_baseline_port(e[:min(sample, 1 << 26)], scale, seed)
    ref = cpu_baseline_reference(e, scale, seed)
    if ref is None:
        return port, None
    return ref, port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # the reference's PageRank runs are 20 iterations (graphtap.slurm:72)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse on a 1-GPU box)")
    ap.add_argument("--spmv", choices=["pb", "pb_f32msg", "edge"], default=os.environ.get("GRAPHTAP_SPMV", "pb_f32msg"),
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
    # IA + JA + x + y of the tile-row (on several ranks x / JA span the ncols_local columns the tile-row reads)
    ncols = i.nnzcols if world == 1 else i.ncols_local
    b_alg = 4 * i.nnz_local + 4 * (ncols + 1) + F * ncols + F * i.nnzrows
    # one rank: the SpMV launch pair also runs PageRank's applicator + next messenger for the rows of the bins one
    # phase-2 workgroup owns; its compulsory bytes per such row: rank read + write, changed flag, degree,
    # row->column map, message = 8 + 8 + 1 + 4 + 4 + (4 | 8)
    fused_rows = int(VR.stats.fused_apply_rows) if world == 1 else 0
    b_fused = fused_rows * (25 + (4 if args.spmv == "pb_f32msg" else 8))
    b_spmv = b_alg
    b_alg += b_fused
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
            key = "scale%d_gpus%d" % (scale, world)   # collected for the default variant (pb_f32msg) only
            traffic = rec.get(key, {}).get("hbm_bytes_per_launch") if args.spmv == "pb_f32msg" else None
        except Exception:
            traffic = None
    out = {
        "metric": "PageRank GTEPS on RMAT-%d" % scale, "value": value, "unit": "GTEPS", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64" if args.spmv != "pb_f32msg" else "f64 accumulate / f32 messages", "data": "synthetic",
        "config": {"workload": "PageRank R-MAT scale %d edge-factor 16 seed %d, flags of apps/pr.cpp (TCSC_CF), 1 step = 1 iteration" % (scale, args.seed),
                   "num_vertices": nv, "edge_records": m, "stored_entries": nnz, "nnzrows": int(i.nnzrows_global), "nnzcols": int(i.nnzcols_global),
                   "spmv": args.spmv, "partition": "tile-rows x%d (1-D), needed-columns all-to-all of x per step" % world if world > 1 else "single tile",
                   "ingress_s": round(t_ingress, 3), "iterations_total": VR.iteration, "value_checksum": checksum[0], "reachable": checksum[1]},
        "roofline": {"bound": "hbm", "kernel": {"pb": "k_pb_scatter<double,double> + k_pb_gather<double,double> (one SpMV = this launch pair)",
                                                  "pb_f32msg": "k_pb_scatter<double,float> + k_pb_gather<double,float> (one SpMV = this launch pair)",
                                                  "edge": "k_spmv_edge<GT_PLUS_F64>"}[args.spmv], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes_per_launch": b_alg,
                     "algorithmic_bytes_spmv": float(b_spmv), "algorithmic_bytes_fused_apply": float(b_fused), "fused_apply_rows": fused_rows,
                     "kernel_ms": kernel_ms, "launches": launches},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        VR.free(); G.free()   # give the HBM back before the host-side runs
        main_b, port_b = cpu_baseline(scale, args.seed)
        out["cpu_baseline"] = main_b
        if port_b is not None:
            out["cpu_baseline_port"] = port_b
    VR.free(); G.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
