import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import graphtap_amd as gt
from graphtap_amd import _lib
L = _lib.lib(); _lib.require_gpu(); _lib.check(L.gt_set_device(0))
scale=26; nv=1<<scale; m=16<<scale
d = C.c_void_p(); _lib.check(L.gt_malloc(C.byref(d), m*8)); _lib.check(L.gt_rmat_generate(d, scale, 1, 0, 0, m, None))
G = gt.Graph(); G.load_device(d.value, m, nv, nv, False, False, False, False, False, gt._2DT_, gt._TCSC_, rank=0, nranks=1)
_lib.check(L.gt_free(d))
P = gt.BFS_Program(G, False, False, True, gt._ROW_); P.root = 0
for run in range(3):
    P.initialize(); h = P._handle(); out=[]
    for it in range(20):
        active = C.c_uint64(); ms=[]
        for call in (lambda: L.gt_program_scatter_gather(h), lambda: L.gt_program_combine(h), lambda: L.gt_program_apply(h, 0, C.byref(active))):
            _lib.check(L.gt_device_synchronize()); t0=time.perf_counter(); _lib.check(call()); _lib.check(L.gt_device_synchronize()); ms.append(round((time.perf_counter()-t0)*1e3,3))
        out.append(ms)
        if active.value == 0: break
    print("run", run, "total", round(sum(map(sum,out)),2), out, flush=True)
