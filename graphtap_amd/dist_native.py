"""The C++ multi-rank driver (csrc/dist.hip, gt_dist_*: RCCL called directly from the library) behind the Python mirror.

    from graphtap_amd import dist_native
    dist_native.init()            # once per process, after torch.distributed.init_process_group: the RCCL unique id
                                  # travels over the existing process group, every rank builds its communicator
    ... Vertex_Program.execute() then runs gt_dist_execute instead of graphtap_amd.dist.run

Results are the same as with the Python driver (same phase calls, same exchange plan); the loop itself, the grouped
ncclSend/ncclRecv rounds and the small all-reduces are host C++."""
import ctypes as C

from . import _lib

_handle = None


def handle():
    return _handle


def init(rank=None, world=None):
    """Collective over the torch.distributed default group (any backend): creates this rank's gt_dist."""
    global _handle
    import torch
    import torch.distributed as dist
    if _handle is not None:
        return _handle
    L = _lib.lib()
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        _lib.check(L.gt_dist_unique_id(buf))
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(list(buf), dtype=torch.uint8, device=dev)
    if world > 1:
        dist.broadcast(t, 0)
    raw = bytes(t.cpu().tolist())
    h = C.c_void_p()
    _lib.check(L.gt_dist_create(C.byref(h), raw, rank, world))
    _handle = h
    return h


def free():
    global _handle
    if _handle is not None:
        _lib.check(_lib.lib().gt_dist_free(_handle))
        _handle = None


def execute(prog, iters):
    """gt_dist_execute for a host-side Vertex_Program; returns (iterations, converged, ExecStats)."""
    st = _lib.ExecStats()
    _lib.check(_lib.lib().gt_dist_execute(_handle, prog._handle(), int(iters), C.byref(st)))
    return st.iterations, bool(st.converged), st
