"""Multi-GPU iteration driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests) for the two exchange steps of an iteration.

Partitioning (the a=1 b=p alternative, which is what the reference itself does at
p=2): rank k owns vertex segment k (height H = nrows/p + 1, src/mat/matrix.hpp:193) and tile-row k
of the p x p grid. Every iteration

  scatter_gather  rank k computes the messages x_k of ITS non-empty columns   (vp:688-758)
  exchange        all-gather of the p segments of x, in K slices                (replaces the
                  MPI_Ibcast down column groups, vp:843-862 / 970-1013)
  combine         local SpMV over tile-row k; y_k is complete, no row-group reduce is needed
                  (the reference's M6/M7 Isend/Irecv, vp:1083-1111, disappears with a = 1)
  apply           local                                                       (vp:1610-1802)
  converged?      all-reduce of one integer                                   (vp:1918)

Degree in _COL_ order (apps/pr.cpp:40-42) accumulates in COLUMN space, where every rank holds a
partial count: there the accumulators are summed with an all-reduce before apply.

The driver only needs the five methods of a "tile engine"; the product engine is
vertex_program._HipEngine (C ABI, device buffers); tests drive the same code over gloo."""
import torch
import torch.distributed as dist


def _staged(t):
    """gloo cannot move device tensors for every collective: rehearsal runs of the N-rank path on a box
    without RCCL peers stage through the host. The product path (backend nccl = RCCL) never does."""
    return dist.get_backend() == "gloo" and t.is_cuda


def all_gather_segments(x, mine, group=None):
    if _staged(x):
        h = x.cpu(); m = mine.cpu().clone()
        dist.all_gather_into_tensor(h, m, group=group)
        x.copy_(h)
    else:
        dist.all_gather_into_tensor(x, mine, group=group)


def all_reduce_sum(t, group=None):
    if _staged(t):
        h = t.cpu(); dist.all_reduce(h, group=group); t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


def run(engine, iters, group=None):
    """Vertex_Program::execute (vp:408-441) across ranks. Returns (iterations, converged).

    The message vector is [K][p][T] (K = engine.x_slices slices of width T of every rank's segment). With K > 1
    the K all-gathers of an iteration are issued back to back (async_op) on RCCL's stream and the local SpMV is
    driven slice by slice: phase 1 of slice k runs while slices k+1.. are still in flight."""
    check = (iters == 0)
    p, k = engine.nranks, engine.rank
    K = getattr(engine, "x_slices", 1)
    x = engine.x_tensor()               # [K * p * T] messages
    T = x.numel() // (K * p)
    slices = [x[s * p * T:(s + 1) * p * T] for s in range(K)]
    mine = [sl[k * T:(k + 1) * T] for sl in slices]
    pipelined = p > 1 and K > 1 and engine.needs_x_exchange and not _staged(x) and hasattr(engine, "combine_slice")
    converged = False
    while True:
        engine.scatter_gather()
        if pipelined:
            works = [dist.all_gather_into_tensor(slices[s], mine[s], group=group, async_op=True) for s in range(K)]
            for s in range(K):
                works[s].wait()          # the compute stream waits for slice s only
                engine.combine_slice(s)
        else:
            if p > 1 and engine.needs_x_exchange:
                for s in range(K):
                    all_gather_segments(slices[s], mine[s], group=group)
            engine.combine()
        if p > 1 and engine.column_accumulators:
            all_reduce_sum(engine.y_tensor(), group=group)
        active = engine.apply(iters, check)
        if check:
            if p > 1:
                t = torch.tensor([active], dtype=torch.int64, device=x.device)
                all_reduce_sum(t, group=group)
                active = int(t.item())
            if active == 0:
                engine.finish_converged()
                converged = True
                break
        elif engine.iteration >= iters:
            break
    return engine.iteration, converged
