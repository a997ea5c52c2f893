#!/usr/bin/env python3
"""Per-rank compute time of the N-GPU PageRank step, measured on ONE GPU: builds tile-row `rank` of `nranks`
of the R-MAT graph and runs the three local phases (scatter_gather, combine, apply) without the exchange.
The missing term of a real N-GPU step is the exchange of x (K all-to-alls over RCCL), which a 1-GPU box cannot run;
the bytes this rank would receive and send per step are reported.
  python tools/bench_tilerow.py --scale 26 --nranks 8 --rank 0"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GRAPHTAP_SPMV", "pb_f32msg")
import graphtap_amd as gt
from graphtap_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=26); ap.add_argument("--nranks", type=int, default=8)
ap.add_argument("--rank", type=int, default=0); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--sliced", action="store_true", help="drive the SpMV slice by slice (gt_program_combine_slice), as the pipelined multi-GPU loop does")
ap.add_argument("--parts", type=int, default=0, help="1: phase 2 part by part on the compute stream, each part followed by the packing of its slice (the pipelined loop of round 4, "
                "gt_program_phase2_part); 2: the parts side by side on streams of their own. Implies --sliced")
a = ap.parse_args()
L = _lib.lib(); _lib.require_gpu(); _lib.check(L.gt_set_device(0))
nv, m = 1 << a.scale, 16 << a.scale
d = C.c_void_p(); _lib.check(L.gt_malloc(C.byref(d), m * 8)); _lib.check(L.gt_rmat_generate(d, a.scale, 1, 0, 0, m, None))
def ingest():
    _lib.check(L.gt_device_synchronize()); t = time.perf_counter()
    G = gt.Graph(); G.load_device(d.value, m, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=a.rank, nranks=a.nranks)
    _lib.check(L.gt_device_synchronize()); return G, time.perf_counter() - t   # ingest of this tile-row: TCSC build + layout + PB build
G, t_first = ingest(); G.free()   # the first build of a process also pays for the first hipMallocs of gigabytes and the code objects
G, t_in = ingest()
_lib.check(L.gt_free(d))
V = gt.Deg_Program(G, True, False, False, gt._COL_); V.initialize()
h = V._handle(); _lib.check(L.gt_program_scatter_gather(h)); _lib.check(L.gt_program_combine(h)); _lib.check(L.gt_program_apply(h, 1, None))
VR = gt.PR_Program(G, True, False, False, gt._ROW_); VR.initialize(V)   # partial degrees: timing only
h = VR._handle()
_lib.check(L.gt_program_enable_timing(h, 1))
K = G.info.x_slices
RAW = C.CDLL(_lib.LIB_PATH)   # engine-internal entry points of the pipelined loop (csrc/gt_internal.h; not part of the ABI header)
RAW.gt_program_parts_begin.restype = C.c_bool; RAW.gt_program_parts_begin.argtypes = [C.c_void_p]
RAW.gt_program_phase2_part.restype = C.c_int; RAW.gt_program_phase2_part.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
RAW.gt_program_pack_slice.restype = C.c_int; RAW.gt_program_pack_slice.argtypes = [C.c_void_p, C.c_uint32]
first = [True]
def step():
    if a.parts:
        if first[0]: _lib.check(L.gt_program_scatter_gather(h)); first[0] = False   # afterwards the parts' applicator writes the messages
        _lib.check(L.gt_program_fuse_apply(h, 0x7fffffff, 0))
        assert RAW.gt_program_parts_begin(h)
        for k in range(K): _lib.check(L.gt_program_combine_slice(h, k))
        for k in range(K):
            _lib.check(RAW.gt_program_phase2_part(h, k, 0x7fffffff, 1 if a.parts == 2 else 0))
            _lib.check(RAW.gt_program_pack_slice(h, k))   # (on the compute stream here; the driver packs on the communication stream)
        return
    _lib.check(L.gt_program_scatter_gather(h))
    _lib.check(L.gt_program_fuse_apply(h, 0x7fffffff, 0))   # as graphtap_amd/dist.py does (GRAPHTAP_FUSE_APPLY=0: no-op)
    if a.sliced:
        for k in range(K): _lib.check(L.gt_program_combine_slice(h, k))
    else:
        _lib.check(L.gt_program_combine(h))
    _lib.check(L.gt_program_apply(h, 0x7fffffff, None))
for _ in range(3): step()
ms, n = C.c_double(), C.c_uint32(); _lib.check(L.gt_program_timing(h, C.byref(ms), C.byref(n), 1))
_lib.check(L.gt_device_synchronize()); t0 = time.perf_counter()
for _ in range(a.steps): step()
_lib.check(L.gt_device_synchronize()); dt = time.perf_counter() - t0
_lib.check(L.gt_program_timing(h, C.byref(ms), C.byref(n), 1))
i = G.info
print(json.dumps({"ingest_s": round(t_in, 4), "ingest_first_s": round(t_first, 4), "scale": a.scale, "rank": a.rank, "nranks": a.nranks, "x_slices": int(K), "sliced": bool(a.sliced), "parts": a.parts, "nnz_local": int(i.nnz_local), "nnzrows": int(i.nnzrows),
                  "seg_stride": int(i.seg_stride), "ms_per_step_compute_only": dt * 1e3 / a.steps, "spmv_ms": ms.value / max(n.value, 1),
                  "ncols_local": int(i.ncols_local), "recv_bytes_f32": int(sum(map(sum, G.exchange_plan()[3])) * 4),
                  "send_bytes_f32": int(i.send_elems * 4), "allgather_bytes_f32": int(i.nnzcols_global * 4)}))
