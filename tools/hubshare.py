import ctypes as C, sys, torch
sys.path.insert(0, '.')
from graphtap_amd import _lib
L = _lib.lib(); _lib.check(L.gt_set_device(0))
for scale in (22, 26):
    m = 16 << scale
    t = torch.empty((m, 2), dtype=torch.int32, device='cuda')
    _lib.check(L.gt_rmat_generate(C.c_void_p(t.data_ptr()), scale, 1, 0, 0, m, None))
    torch.cuda.synchronize()
    dst = t[:, 1].long()
    deg = torch.bincount(dst, minlength=1 << scale)
    d, _ = torch.sort(deg, descending=True)
    cs = torch.cumsum(d, 0).double() / m
    nz = int((deg > 0).sum())
    print("scale", scale, "rows with in-edges", nz)
    for k in (1024, 4096, 8192, 12288, 16384, 32768, 65536, 262144, 1 << 20):
        print("  top %8d rows hold %5.1f%% of entries (degree >= %d)" % (k, 100 * cs[k - 1].item(), int(d[k - 1])))
    del t, dst, deg, d, cs
