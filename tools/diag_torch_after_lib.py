import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import graphtap_amd as gt
L = gt._lib.lib()
print("count", L.gt_device_count())
print("set", L.gt_set_device(0))
import ctypes as C
d = C.c_void_p(); print("malloc", L.gt_malloc(C.byref(d), 1024))
import torch
print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
try:
    print(torch.zeros(1, device="cuda"))
except Exception as e:
    print("ERR", e)
print({k: v for k, v in os.environ.items() if "VISIBLE" in k or "HSA" in k or "HIP" in k})
