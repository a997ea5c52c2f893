#!/usr/bin/env python3
"""The first BFS of a process on R-MAT-26, phase by phase (scatter_gather / combine / apply with a device sync after each), three
times, then one whole execute() whose gt_exec_stats.allocs_in_execute is printed. Run as the FIRST process of a gpurun lease it
shows what a cold process pays where: round 2 saw 2.7-3.9 s in iteration 1's combine of the first process of a fresh box -- the
value stream was allocated (a multi-GB hipMalloc) and first touched exactly there; initialize() reserves it now
(gt_pb_reserve_val, the [build] line of GRAPHTAP_PB_STATS carries its time).
  GRAPHTAP_PB_STATS=1 python tools/cold_steps.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import graphtap_amd as gt
from graphtap_amd import _lib
L = _lib.lib(); _lib.require_gpu(); _lib.check(L.gt_set_device(0))
scale = 26; nv = 1 << scale; m = 16 << scale
t0 = time.perf_counter()
d = C.c_void_p(); _lib.check(L.gt_malloc(C.byref(d), m * 8)); _lib.check(L.gt_rmat_generate(d, scale, 1, 0, 0, m, None))
G = gt.Graph(); G.load_device(d.value, m, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
_lib.check(L.gt_free(d)); _lib.check(L.gt_device_synchronize())
print("generate + build: %.3f s" % (time.perf_counter() - t0), flush=True)
P = gt.BFS_Program(G, False, False, True, gt._ROW_); P.root = 0
for run in range(3):
    t0 = time.perf_counter(); P.initialize(); _lib.check(L.gt_device_synchronize()); t_init = time.perf_counter() - t0
    h = P._handle(); out = []
    for it in range(20):
        active = C.c_uint64(); ms = []
        for call in (lambda: L.gt_program_scatter_gather(h), lambda: L.gt_program_combine(h), lambda: L.gt_program_apply(h, 0, C.byref(active))):
            _lib.check(L.gt_device_synchronize()); t0 = time.perf_counter(); _lib.check(call()); _lib.check(L.gt_device_synchronize()); ms.append(round((time.perf_counter() - t0) * 1e3, 3))
        out.append(ms)
        if active.value == 0: break
    print("run", run, "initialize %.1f ms," % (t_init * 1e3), "iterations total", round(sum(map(sum, out)), 2), "ms", out, flush=True)
P.initialize(); P.execute()
print("execute(): %d iterations, %.3f ms, allocations inside the iteration loop: %d" % (P.stats.iterations, P.stats.seconds * 1e3, P.stats.allocs_in_execute), flush=True)
