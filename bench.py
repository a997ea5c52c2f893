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
B_alg = 4 nnz + 4 (nnzcols+1) + Fx nnzcols + 8 nnzrows (Fx = 4 with f32 messages, else 8): SpMV bytes only.
The fused applicator's bytes are reported separately (`frac_with_apply`); `f64_messages` is the all-f64 run of the
same graph, `full_applicator` the run whose applicator stores rank / changed flags in every iteration like the reference's;
`measured_ceiling_GBps` (1:1 copy), `read_ceiling_GBps` and `mixed_ceiling_GBps` are this box's streaming rates, measured in this run.
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
                               capture_output=True, text=True, timeout=300)
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


def _host_cores():
    """Cores this process may use: the scheduler affinity, cut by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(scale, seed):
    """The reference on the box's host cores: as many MPI ranks as it has cores (the largest count <= 64 that the reference's
    integer_factorize, src/mat/tiling.hpp:65-73, splits into a near-square rank grid), over a bounded sample of the same stream:
    the first 2^28 records with 16+ cores, 2^27 below."""
    cores = _host_cores()
    nproc = max(n for n in (1, 2, 4, 6, 8, 9, 12, 16, 20, 24, 25, 30, 32, 36, 42, 48, 49, 56, 64) if n <= max(cores, 1))
    sample = min(16 << scale, 1 << (28 if nproc >= 16 else 27))
    e = _sample_edges(scale, seed, sample)
    port = cpu_baseline_port(e[:min(sample, 1 << 26)], scale, seed)
    ref = cpu_baseline_reference(e, scale, seed, nproc=nproc)
    if ref is None and nproc > 8:        # e.g. not enough memory for that many ranks: the 8-rank run of the earlier rounds
        ref = cpu_baseline_reference(e[:1 << 27], scale, seed, nproc=8)
    if ref is None:
        return port, None
    ref["host_cores"] = cores
    return ref, port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # the reference's PageRank runs are 20 iterations (graphtap.slurm:72)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f64", action="store_true", help="skip the second, all-f64 run whose numbers go into `f64_messages`")
    ap.add_argument("--driver", choices=["native", "python"], default="native",
                    help="N > 1: the iteration loop in C++ calling RCCL directly (csrc/dist.hip, default) or in Python over torch.distributed (graphtap_amd/dist.py)")
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
    # GRAPHTAP_FORCE_EXCHANGE=1 on one GPU: the N-rank code path (exchange layout, gt_dist_execute over RCCL at world size 1, the
    # per-rank diagnostics below) rehearsed on a 1-GPU box
    dist_on = world > 1 or bool(os.environ.get("GRAPHTAP_FORCE_EXCHANGE"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        if args.driver == "native" and args.backend == "nccl":
            from graphtap_amd import dist_native
            dist_native.init(rank, world)

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
    G = gt.Graph()
    native = dist_on and args.driver == "native" and args.backend == "nccl"
    if native and os.environ.get("GRAPHTAP_BENCH_REPLICATED_BUILD", "0") in ("", "0"):
        # Matrix::distribute (mat/matrix.hpp:693-810), the path the application mains take: every rank generates ITS 1/N of the
        # record stream (the generator is counter based: records [first, first + count) of the same stream) and
        # gt_graph_build_distributed shuffles the records to the owners of their rows over RCCL. `ingress_s` is then the time of
        # the distributed build (per rank in `per_rank`), not of N replicated ones.
        first = m * rank // world
        count = m * (rank + 1) // world - first
        _lib.check(L.gt_malloc(C.byref(d), max(count, 1) * 8))
        _lib.check(L.gt_rmat_generate(d, scale, args.seed, 0, first, count, None))
        G.load_share(dist_native.handle(), rank, world, d.value, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, m_share=count)
        build_kind = "distributed: 1/%d of the records per rank, shuffled by gt_graph_build_distributed" % world
    else:
        _lib.check(L.gt_malloc(C.byref(d), m * 8))
        _lib.check(L.gt_rmat_generate(d, scale, args.seed, 0, 0, m, None))
        G.load_device(d.value, m, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=rank, nranks=world)  # apps/pr.cpp:26-36
        build_kind = "replicated: every rank sees all records and keeps its tile-row" if dist_on else "single rank"
    _lib.check(L.gt_free(d))
    wide_build = bool(L.gt_graph_has_wide_build(G._h))   # windows of 32 766 slots for the 4-byte-message SpMVs
    V = gt.Deg_Program(G, True, False, False, gt._COL_)   # apps/pr.cpp:37-42
    V.execute(1)
    VR = gt.PR_Program(G, True, False, False, gt._ROW_)   # apps/pr.cpp:46-50
    VR.initialize(V)
    torch.cuda.synchronize()
    t_ingress_mine = time.perf_counter() - t_in0   # this rank's own ingress (per_rank); the line's `ingress_s` is up to the barrier
    barrier()
    t_ingress = time.perf_counter() - t_in0

    phase_ms = []   # per timed run: mean [phase 1, phase 2] ms of the SpMV, when GRAPHTAP_PB_PHASE_TIMING=1

    def timed_run(VR):
        """W untimed warm-up steps, a fresh initialize(V) (untimed), then ONE execute(K) -- so that the timed steps are
        exactly the reference's `pr <file> <n> K` (its TCSC_CF `last iteration` rule fires once, at step K) and the
        printed checksum is the one the reference prints for K iterations."""
        h = VR._handle()
        if args.warmup:
            VR.execute(args.warmup)
            VR.initialize(V)
        _lib.check(L.gt_program_enable_timing(h, 1))
        ms0, n0 = C.c_double(), C.c_uint32()
        _lib.check(L.gt_program_timing(h, C.byref(ms0), C.byref(n0), 1))
        p1, p2, pn = C.c_double(), C.c_double(), C.c_uint32()
        _lib.check(L.gt_graph_phase_times(G._h, C.byref(p1), C.byref(p2), C.byref(pn), 1))   # diagnostics (GRAPHTAP_PB_PHASE_TIMING=1): drop the warm-up's samples
        barrier()
        t0 = time.perf_counter()
        VR.execute(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        _lib.check(L.gt_graph_phase_times(G._h, C.byref(p1), C.byref(p2), C.byref(pn), 1))
        phase_ms.append([p1.value, p2.value] if pn.value else None)
        if not G.exchange or VR.stats is not None and dist_on and args.driver == "native" and args.backend == "nccl":
            spmv_ms, launches = VR.stats.spmv_ms, VR.stats.spmv_launches
        else:
            a, b = C.c_double(), C.c_uint32()
            _lib.check(L.gt_program_timing(h, C.byref(a), C.byref(b), 1))
            spmv_ms, launches = a.value, b.value
        assert launches == args.steps, (launches, args.steps)
        return dt, spmv_ms / launches, launches

    dt, kernel_ms, launches = timed_run(VR)

    i = G.info
    # B_alg = 4 nnz [IA] + 4 (nnzcols+1) [JA] + Fx nnzcols [x] + 8 nnzrows [y] of the tile-row (on several ranks x / JA
    # span the ncols_local columns the tile-row reads); Fx = bytes of a message as this variant keeps it in HBM
    ncols = i.ncols_local if G.exchange else i.nnzcols

    def b_spmv_of(variant):
        fx = 4 if variant == "pb_f32msg" else 8
        return 4 * i.nnz_local + 4 * (ncols + 1) + fx * ncols + 8 * i.nnzrows
    b_spmv = b_spmv_of(args.spmv)
    # one rank: the SpMV launch pair also runs PageRank's applicator + next messenger for the rows of the bins one
    # phase-2 workgroup owns; its compulsory bytes per such row with the full applicator: rank read + write, changed
    # flag, degree, row->column map, message = 8 + 8 + 1 + 4 + 4 + (4 | 8). Reported beside the headline fraction, never inside it.
    # Since round 3 a fixed-count run writes rank / changed flags only where somebody can see them (gt_internal.h, pr_state): the last
    # iteration is the full applicator, the one before it writes rank (8 + 4 + 4 + Fx), the others only the next messages
    # (degree, row->slot map, message = 4 + 4 + Fx); the figure below is the mean over the K timed steps.
    fused_rows = int(VR.stats.fused_apply_rows) if not G.exchange else 0
    fx_b = 4 if args.spmv == "pb_f32msg" else 8
    lean = os.environ.get("GRAPHTAP_PR_LEAN_STATE", "1") != "0"
    K = args.steps
    per_row = (25 + fx_b) if not lean else ((25 + fx_b) + (16 + fx_b) * min(1, K - 1) + (8 + fx_b) * max(0, K - 2)) / K
    b_fused = fused_rows * per_row
    t = torch.tensor([dt, kernel_ms, float(b_spmv), float(b_fused)], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, kernel_ms, b_spmv, b_fused = float(t[0]), float(t[1]), float(t[2]), float(t[3])
    nnz = G.nnz_global
    value = nnz * args.steps / dt / 1e9
    achieved = b_spmv / (kernel_ms * 1e-3) / 1e9     # GB/s of the slowest rank's SpMV launch, SpMV bytes only

    checksum = VR.checksum(out=None)
    iterations_total = VR.iteration
    # roofline.traffic: HBM bytes per SpMV from the rocprofv3 --pmc passes (profiles/collect_pmc.py). It is a number of the
    # builder's collection, not of this run -- so it carries where it came from, and it is null as soon as the kernels' source
    # (graphtap_amd/csrc/pb.hip) is no longer the one the counters were collected from.
    traffic, traffic_source = None, None
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written by profiles/collect_pmc.py from rocprofv3 --pmc passes
    if os.path.exists(tp):
        try:
            import hashlib
            rec = json.load(open(tp)).get("scale%d_gpus%d" % (scale, world), {})   # collected for the default variant (pb_f32msg) only
            sha = hashlib.sha256(open(os.path.join(ROOT, "graphtap_amd", "csrc", "pb.hip"), "rb").read()).hexdigest()[:16]
            col = rec.get("collected", {})
            current = col.get("pb_hip_sha16") == sha
            traffic_source = {"file": "profiles/pmc_traffic.json", "collected": col.get("date"), "commit": col.get("commit"),
                              "kernel_source_unchanged_since": bool(current)}
            if args.spmv == "pb_f32msg" and current:
                traffic = rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic, traffic_source = None, None

    # the like-for-like applicator beside it: the reference's apply_stationary (vp:1641-1693) writes rank and changed flags in
    # EVERY iteration; the headline run elides the stores nobody can see. Same graph, same variant, full applicator.
    full_rec = None
    if world == 1 and not dist_on and lean and not args.no_f64:
        os.environ["GRAPHTAP_PR_LEAN_STATE"] = "0"
        try:
            dt3, kms3, _ = timed_run(VR)
        finally:
            del os.environ["GRAPHTAP_PR_LEAN_STATE"]
        full_rec = {"applicator": "full state every iteration (GRAPHTAP_PR_LEAN_STATE=0)", "spmv": args.spmv, "value": nnz * args.steps / dt3 / 1e9, "unit": "GTEPS",
                    "ms_per_step": dt3 * 1e3 / args.steps, "kernel_ms": kms3, "frac": b_spmv / (kms3 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "value_checksum": VR.checksum(out=None)[0], **({"phase_ms": phase_ms[-1]} if phase_ms[-1] else {})}

    # the all-f64 path beside it (the reference's fp is double, apps/deg.h:19): same graph, messages kept in f64
    f64_rec = None
    if world == 1 and not dist_on and args.spmv == "pb_f32msg" and not args.no_f64:
        VR.free()
        _lib.check(L.gt_graph_select_spmv(G._h, _lib.GT_SPMV_PB))
        VR = gt.PR_Program(G, True, False, False, gt._ROW_)
        VR.initialize(V)
        dt2, kms2, _ = timed_run(VR)
        f64_rec = {"spmv": "pb", "value": nnz * args.steps / dt2 / 1e9, "unit": "GTEPS", "ms_per_step": dt2 * 1e3 / args.steps, "kernel_ms": kms2,
                   "algorithmic_bytes_spmv": float(b_spmv_of("pb")), "frac": b_spmv_of("pb") / (kms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "value_checksum": VR.checksum(out=None)[0], **({"phase_ms": phase_ms[-1]} if len(phase_ms) > 1 and phase_ms[-1] else {})}

    # the streaming ceilings of THIS box, measured live with the library's diagnostic kernels (csrc/diag.hip: the shapes of
    # tools/hbm_ceiling2.hip): DESIGN 4.1's argument -- pure reads run at ~7 TB/s here, ANY read/write mix at ~5.2 -- in the line
    ceiling, ceilings = None, None
    if world == 1 and not dist_on:
        VR.free(); V.free(); G.free()
        torch.cuda.empty_cache()
        ceilings = {}
        for name, mode in (("read_nt", 0), ("write", 1), ("copy_1to1", 2), ("mix_3r_2w", 3), ("mix_16r_1w_nt", 4), ("mix_7r_1w_nt", 5)):
            v = C.c_double()
            if L.gt_diag_hbm_ceiling(mode, 4 << 30, C.byref(v)) == 0:
                ceilings[name] = round(v.value, 1)
        ceiling = ceilings.get("copy_1to1")
    out = {
        "metric": "PageRank GTEPS on RMAT-%d" % scale, "value": value, "unit": "GTEPS", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64" if args.spmv != "pb_f32msg" else "f64 accumulate / f32 messages", "data": "synthetic",
        "config": {"workload": "PageRank R-MAT scale %d edge-factor 16 seed %d, flags of apps/pr.cpp (TCSC_CF), 1 step = 1 iteration" % (scale, args.seed),
                   "num_vertices": nv, "edge_records": m, "stored_entries": nnz, "nnzrows": int(G.nnzrows_global), "nnzcols": int(i.nnzcols_global),
                   "spmv": args.spmv, "pb_build": ("wide: 32 766-slot windows for the f32-message SpMVs, the narrow build beside it for f64 messages" if wide_build and args.spmv == "pb_f32msg" else "narrow: 16 383-slot windows"), "applicator": "rank / changed flags stored by the last two iterations only (dead stores elided; GRAPHTAP_PR_LEAN_STATE=0: every iteration)" if lean else "full state every iteration",
                   "partition": "tile-rows x%d (1-D), needed-columns all-to-all of x per step" % world if dist_on else "single tile",
                   "driver": ("C++ gt_dist_execute over RCCL" if args.driver == "native" and args.backend == "nccl" else "python dist.run over torch.distributed/" + args.backend) if dist_on else "C++ gt_program_execute",
                   "ingress_s": round(t_ingress, 3), "build": build_kind, "iterations_total": iterations_total, "value_checksum": checksum[0], "reachable": checksum[1]},
        "roofline": {"bound": "hbm", "kernel": {"pb": "k_pb_scatter* + k_pb_gather<double,double> (one SpMV = this launch group)",
                                                  "pb_f32msg": "k_pb_scatter* + k_pb_gather<double,float> (one SpMV = this launch group)",
                                                  "edge": "k_spmv_edge<GT_PLUS_F64>"}[args.spmv], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": b_spmv,
                     "frac_with_apply": (b_spmv + b_fused) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "algorithmic_bytes_fused_apply": float(b_fused), "fused_apply_rows": fused_rows,
                     "measured_ceiling_GBps": ceiling, "read_ceiling_GBps": (ceilings or {}).get("read_nt"),
                     "mixed_ceiling_GBps": ({"3:2 (phase 1's mix)": ceilings.get("mix_3r_2w"), "16:1 nt (phase 2's mix, lean applicator)": ceilings.get("mix_16r_1w_nt"),
                                             "7:1 nt (phase 2's mix, full applicator)": ceilings.get("mix_7r_1w_nt"), "write only": ceilings.get("write")} if ceilings else None),
                     "kernel_ms": kernel_ms, "launches": launches,
                     **({"phase_ms": phase_ms[0]} if phase_ms and phase_ms[0] else {})},
    }
    if full_rec is not None:
        out["full_applicator"] = full_rec
    if f64_rec is not None:
        out["f64_messages"] = f64_rec
    if rank == 0 and world == 1 and not dist_on and not args.no_cpu_baseline:
        main_b, port_b = cpu_baseline(scale, args.seed)
        out["cpu_baseline"] = main_b
        if port_b is not None:
            out["cpu_baseline_port"] = port_b
    if dist_on:
        # every rank's own account of the run (C++ driver): mean per-iteration pack / exchange landing / SpMV span / apply times by
        # HIP events, bytes it sent, how many ranks RCCL counts -- so that a scaling run explains itself
        if args.driver == "native" and args.backend == "nccl":
            mine = dist_native.diagnostics()
            mine.pop("per_iteration", None)
            mine.update(rank=rank, ingress_s=round(t_ingress_mine, 3), spmv_kernel_ms=round(float(VR.stats.spmv_ms) / max(int(VR.stats.spmv_launches), 1), 4), nnz_local=int(i.nnz_local),
                        ncols_local=int(i.ncols_local))
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            out["per_rank"] = allr
            out["rccl_ranks"] = mine["transport_ranks"]
        VR.free(); V.free(); G.free()
        if args.driver == "native" and args.backend == "nccl":
            dist_native.free()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
