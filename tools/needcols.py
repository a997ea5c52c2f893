import ctypes as C, os, sys, numpy as np
sys.path.insert(0, '.')
os.environ["GRAPHTAP_SPMV"] = "edge"
import graphtap_amd as gt
from graphtap_amd import _lib
L = _lib.lib(); _lib.check(L.gt_set_device(0))
for scale, p in ((26, 8), (26, 4), (26, 2)):
    nv, m = 1 << scale, 16 << scale
    d = C.c_void_p(); _lib.check(L.gt_malloc(C.byref(d), m * 8)); _lib.check(L.gt_rmat_generate(d, scale, 1, 0, 0, m, None))
    G = gt.Graph(); G.load_device(d.value, m, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=1, nranks=p)
    _lib.check(L.gt_free(d))
    i = G.info
    n = i.nranks * i.seg_stride + 1
    JA = np.zeros(n, np.uint32); _lib.check(L.gt_memcpy_d2h(JA.ctypes.data_as(C.c_void_p), G.tile().JA, n * 4))
    present = int((np.diff(JA.astype(np.int64)) > 0).sum())
    print("scale %d p=%d: columns with local entries %d of %d non-empty columns = %.3f; nnz_local %d" % (scale, p, present, i.nnzcols_global, present / i.nnzcols_global, i.nnz_local))
    G.free()
