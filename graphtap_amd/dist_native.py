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


TIME_FIELDS = ("pack_ms", "first_slice_ms", "last_slice_ms", "spmv_span_ms", "apply_ms", "rest_ms", "mode")


def diagnostics(h=None):
    """Per-rank record of the last gt_dist_execute (gt_dist_iteration_times, gt_dist_info, gt_dist_exchange_stats): mean
    per-iteration times, bytes sent, ranks the transport counts. bench.py gathers one per rank."""
    L = _lib.lib()
    h = _handle if h is None else h
    n = C.c_uint32()
    _lib.check(L.gt_dist_iteration_times(h, None, 0, C.byref(n)))
    buf = (C.c_double * (7 * max(n.value, 1)))()
    _lib.check(L.gt_dist_iteration_times(h, buf, n.value, C.byref(n)))
    rows = [[buf[i * 7 + j] for j in range(7)] for i in range(n.value)]
    ranks, li, ps, rt = C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    _lib.check(L.gt_dist_info(h, C.byref(ranks), C.byref(li), C.byref(ps), C.byref(rt)))
    sent, dense, nx = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _lib.check(L.gt_dist_exchange_stats(h, C.byref(sent), C.byref(dense), C.byref(nx), 0))

    def mean(j):
        v = [r[j] for r in rows if r[j] >= 0]
        return round(sum(v) / len(v), 4) if v else None
    out = {TIME_FIELDS[j]: mean(j) for j in range(6)}
    out.update(timed_iterations=n.value, transport_ranks=ranks.value, list_iterations=li.value, pair_spmspv_iterations=ps.value,
               host_round_trips=rt.value, bytes_sent=sent.value, bytes_if_dense=dense.value, exchanges=nx.value)
    # exchange_wait_ms: how long the SpMV span exceeds... is not separable from one run; the span and the landing times are both given
    out["per_iteration"] = [[round(x, 4) for x in r] for r in rows[:32]]
    return out
