"""Multi-GPU iteration driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests) for the two exchange steps of an iteration.

Partitioning (the a=1 b=p alternative, which is what the reference itself does at
p=2): rank k owns vertex segment k (height H = nrows/p + 1, src/mat/matrix.hpp:193) and tile-row k
of the p x p grid. Every iteration

  scatter_gather  rank k computes the messages of ITS non-empty columns and packs, per
                  destination d, the ones tile-row d has an entry for         (vp:688-758)
  exchange        K all-to-alls (slice by slice) deliver to every rank exactly the
                  columns its tile-row reads -- 47 % of an all-gather's volume at 8 ranks on
                  R-MAT-26, all 7 xGMI links of a GPU busy at once (replaces the
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


def exchange_slice(x, send, plan, k, group=None, async_op=False):
    """Slice k of the message exchange (gt_graph_exchange_plan): block d of the send buffer's slice k goes to rank d,
    whose x receives it as block `me` of ITS slice k. Returns the work handle when async_op."""
    send_off, recv_off, send_counts, recv_counts = plan
    out = x[recv_off[k]:recv_off[k] + sum(recv_counts[k])]
    inp = send[send_off[k]:send_off[k] + sum(send_counts[k])]
    if _staged(x):
        ho, hi = out.cpu(), inp.cpu()
        dist.all_to_all_single(ho, hi, recv_counts[k], send_counts[k], group=group)
        out.copy_(ho)
        return None
    return dist.all_to_all_single(out, inp, recv_counts[k], send_counts[k], group=group, async_op=async_op)


def verify_plan(plan, device, group=None):
    """Once per engine: every rank tells every peer how much it is going to send per slice; a disagreement with the
    receiver's own plan (which would make the all-to-alls hang or scribble) is reported instead."""
    _, _, send_counts, recv_counts = plan
    K, p = len(send_counts), len(send_counts[0])
    dev = "cpu" if dist.get_backend() == "gloo" else device
    mine = torch.tensor([[send_counts[k][d] for k in range(K)] for d in range(p)], dtype=torch.int64, device=dev)   # [dest][slice]
    theirs = torch.empty_like(mine)
    dist.all_to_all_single(theirs, mine, group=group)                                                               # [source][slice]
    want = torch.tensor([[recv_counts[k][s] for k in range(K)] for s in range(p)], dtype=torch.int64, device=dev)
    if not torch.equal(theirs, want):
        raise RuntimeError("exchange plan mismatch between ranks (were the graphs built from the same edge list and flags?)")


def all_reduce_sum(t, group=None):
    if _staged(t):
        h = t.cpu(); dist.all_reduce(h, group=group); t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


def run(engine, iters, group=None):
    """Vertex_Program::execute (vp:408-441) across ranks. Returns (iterations, converged).

    With K = engine.x_slices > 1 the K all-to-alls of an iteration are issued back to back (async_op) on RCCL's
    stream and the local SpMV is driven slice by slice: phase 1 of slice k runs while slices k+1.. are in flight."""
    if iters == 0:
        engine.check_sticky = True      # vp:412-413: check_for_convergence is never cleared once execute(0) ran
    check = getattr(engine, "check_sticky", False)
    if hasattr(engine, "prepare"):
        engine.prepare(iters)           # gt_program_prepare: before the buffers are looked at (it fixes their element width)
    p = engine.nranks
    K = getattr(engine, "x_slices", 1)
    x = engine.x_tensor()               # the message vector the local SpMV reads
    if x.is_cuda and getattr(engine, "stream", None) is not None and torch.cuda.current_stream().cuda_stream != engine.stream:
        # work.wait() orders the CURRENT torch stream behind a collective; the engine enqueues on the stream it was given
        raise RuntimeError("dist.run: torch's current stream is not the stream the engine launches on")
    forced = getattr(engine, "force_exchange", False)   # one rank with the exchange layout: every collective runs, with itself
    multi = p > 1 or forced
    exchange = multi and engine.needs_x_exchange
    if exchange:
        send, plan = engine.send_tensor(), engine.exchange_plan()
        if not getattr(engine, "_plan_verified", False):
            verify_plan(plan, x.device, group)
            engine._plan_verified = True
    pipelined = exchange and K > 1 and not _staged(x) and hasattr(engine, "combine_slice")
    converged = False
    arm = getattr(engine, "arm_fused_apply", None) if not engine.column_accumulators else None
    while True:
        engine.scatter_gather()
        if arm:
            arm(iters, check)        # apply follows combine at once: the engine may fuse the two (PageRank)
        if pipelined:
            works = [exchange_slice(x, send, plan, s, group, async_op=True) for s in range(K)]
            for s in range(K):
                works[s].wait()          # the compute stream waits for slice s only
                engine.combine_slice(s)
        else:
            if exchange:
                for s in range(K):
                    exchange_slice(x, send, plan, s, group)
            engine.combine()
        if multi and engine.column_accumulators:
            all_reduce_sum(engine.y_tensor(), group=group)
        active = engine.apply(iters, check)
        if check:
            if multi:
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
