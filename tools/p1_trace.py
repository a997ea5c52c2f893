#!/usr/bin/env python3
"""Per-workgroup timeline of the two SpMV phases (experiment build: tools/build_variant.sh trace pb.hip -DGT_EXP_TRACE).
  GRAPHTAP_LIB=$PWD/graphtap_amd/lib/variants/trace.so python tools/p1_trace.py [--scale 26] [--spmv pb_f32msg]
Every workgroup of the LAST phase-1 / phase-2 launch stamps the 100-MHz wall clock at its start and end (phase 2: also when its
stream has been combined), what it worked on and the CU it ran on. Printed: how long the chip was full, the tail during which
CUs idle, per-CU busy time, the gaps between a CU's consecutive workgroups, and the rate by chunk size."""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=26); ap.add_argument("--spmv", default="pb_f32msg"); ap.add_argument("--iters", type=int, default=6)
a = ap.parse_args()
os.environ["GRAPHTAP_SPMV"] = a.spmv
import graphtap_amd as gt
from graphtap_amd import _lib
L = _lib.lib()
_lib.require_gpu()
scale, nv = a.scale, 1 << a.scale
m = 16 << scale
d = C.c_void_p()
_lib.check(L.gt_malloc(C.byref(d), m * 8)); _lib.check(L.gt_rmat_generate(d, scale, 1, 0, 0, m, None))
G = gt.Graph(); G.load_device(d.value, m, nv, nv, True, True, True, False, True, gt._2DT_, gt._TCSC_CF_, rank=0, nranks=1)
_lib.check(L.gt_free(d))
V = gt.Deg_Program(G, True, False, False, gt._COL_); V.execute(1)
P = gt.PR_Program(G, True, False, False, gt._ROW_); P.initialize(V)
h = P._handle()
for it in range(a.iters):   # stepped by hand as iterations of a 20-iteration run: none is the last one (whose launch adds the source rows' chunks)
    _lib.check(L.gt_program_scatter_gather(h)); _lib.check(L.gt_program_fuse_apply(h, 20, 0)); _lib.check(L.gt_program_combine(h)); _lib.check(L.gt_program_apply(h, 20, None))
raw = C.CDLL(os.environ["GRAPHTAP_LIB"])
p1 = np.zeros(4 * 16384, np.uint64); p2 = np.zeros(4 * 16384, np.uint64)
assert raw.gt_exp_trace_dump(p1.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p)) == 0
TICK = 0.01   # us per tick (100 MHz)


def cu_of(where):
    where = where & np.uint64((1 << 36) - 1)   # (above: the chunk's outputs, phase 1)
    hw = where & 0xFFFFFFFF; xcc = (where >> 32) & 0xF
    return (xcc.astype(np.int64) << 16) | ((hw >> 8) & 0xF).astype(np.int64) | (((hw >> 12) & 0xF).astype(np.int64) << 4)   # XCC, cu_id, sh/se


def report(name, start, end, where, size, extra=None):
    ok = start > 0
    start, end, where, size = start[ok].astype(np.int64), end[ok].astype(np.int64), where[ok], size[ok].astype(np.int64)
    t0 = start.min(); s = (start - t0) * TICK; e = (end - t0) * TICK
    total = e.max()
    cu = cu_of(where); cus = np.unique(cu)
    print("== %s: %d workgroups on %d CUs, %.1f us from the first start to the last end" % (name, len(s), len(cus), total))
    # occupancy over time: how many workgroups run at t
    ts = np.linspace(0, total, 201)[:-1]
    occ = np.array([((s <= t) & (e > t)).sum() for t in ts])
    full = occ.max()
    print("   workgroups in flight: max %d; time at >= 95 %% of it: %.1f %%; at < 50 %%: %.1f %%" % (full, 100 * (occ >= 0.95 * full).mean(), 100 * (occ < 0.5 * full).mean()))
    last_full = ts[occ >= 0.95 * full].max() if (occ >= 0.95 * full).any() else 0
    print("   the chip stays full until %.1f us (%.1f %% of the launch); busy integral %.1f %% of max x duration" % (last_full, 100 * last_full / total, 100 * occ.sum() / (full * len(ts))))
    dur = e - s
    print("   workgroup duration: median %.1f us, p10 %.1f, p90 %.1f, max %.1f; first start spread %.1f us" % (np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.max(), np.sort(s)[min(len(s) - 1, full - 1)]))
    # per CU: busy time and gaps between consecutive workgroups
    busy, gaps = [], []
    for c in cus:
        i = np.where(cu == c)[0]; o = i[np.argsort(s[i])]
        busy.append(dur[o].sum())
        # (several workgroups may share a CU: merge overlapping intervals for the gaps)
        cur_e = e[o[0]]
        for j in o[1:]:
            if s[j] > cur_e: gaps.append(s[j] - cur_e)
            cur_e = max(cur_e, e[j])
    busy = np.array(busy); gaps = np.array(gaps) if gaps else np.zeros(1)
    print("   per-CU busy (sum of its workgroups' durations): min %.1f median %.1f max %.1f us; idle gaps between a CU's workgroups: n %d median %.2f p90 %.2f sum/CU %.1f us" % (busy.min(), np.median(busy), busy.max(), len(gaps), np.median(gaps), np.percentile(gaps, 90), gaps.sum() / len(cus)))
    # rate by size
    if size.max() > 0:
        rate = size / np.maximum(dur, 1e-3)
        for lo, hi in ((0, 25), (25, 50), (50, 75), (75, 100)):
            q0, q1 = np.percentile(size, lo), np.percentile(size, hi)
            sel = (size >= q0) & (size <= q1)
            print("   size quartile %3d-%3d %% (%8d .. %8d units): median duration %7.1f us, median rate %7.1f units/us" % (lo, hi, q0, q1, np.median(dur[sel]), np.median(rate[sel])))
    if extra is not None: extra(ok, s, e, t0)


w1 = p1.reshape(-1, 8)
# a duration model of a phase-1 chunk: a * entries + b * outputs + c (least squares over the launch), and what list scheduling
# in the order of THAT estimate would give against the order by entries the library uses
ok1 = w1[:, 0] > 0
E1 = ((w1[ok1, 2] & np.uint64(0xFFFFFFFF)) * np.uint64(4)).astype(np.float64); D1 = (w1[ok1, 1].astype(np.int64) - w1[ok1, 0].astype(np.int64)) * TICK
S1 = (w1[ok1, 3] >> np.uint64(36)).astype(np.float64); S1 = np.where(S1 == 0, E1, S1)
A = np.stack([E1, S1, np.ones_like(E1)], 1); coef, *_ = np.linalg.lstsq(A, D1, rcond=None)
pred = A @ coef
print("== phase-1 chunk duration ~ %.4f us per 1000 entries + %.4f us per 1000 outputs + %.2f us; residual rms %.1f us (median duration %.1f)" % (coef[0] * 1e3, coef[1] * 1e3, coef[2], np.sqrt(((pred - D1) ** 2).mean()), np.median(D1)))


def makespan(order, dur, m=256):
    import heapq
    h = [0.0] * m
    for i in order:
        t = heapq.heappop(h); heapq.heappush(h, t + dur[i])
    return max(h)


print("   list scheduling of the measured durations on 256 slots: in the order of the entries %.1f us, of the estimate %.1f us, of the durations themselves %.1f us, sum/256 = %.1f us" % (
    makespan(np.argsort(-E1), D1), makespan(np.argsort(-pred), D1), makespan(np.argsort(-D1), D1), D1.sum() / 256))
# where the time between two chunks of a workgroup goes (persistent form): barrier after the last wave -> draw back -> chunk entry -> window staged
if (w1[ok1, 6] > 0).any():
    t = w1[ok1].astype(np.int64); sel = t[:, 6] > 0
    t = t[sel]
    print("== between two chunks of a persistent workgroup (medians, us): draw (atomic + description) %.2f; to the chunk's entry %.2f; staging the window (incl. the first trip's loads) %.2f; wave 0 done -> last wave done %.2f" % (
        np.median(t[:, 7] - t[:, 6]) * TICK, np.median(t[:, 0] - t[:, 7]) * TICK, np.median(t[:, 4] - t[:, 0]) * TICK, np.median(t[:, 1] - t[:, 5]) * TICK))
    # the same per CU: previous chunk's last wave done -> this chunk's barrier passed needs the CU mapping; gaps are reported below
if hasattr(raw, "gt_exp_trace_dump_waves"):   # the 16 waves of a phase-1 workgroup: when each was done
    ww = np.zeros(16 * 8192, np.uint64); assert raw.gt_exp_trace_dump_waves(ww.ctypes.data_as(C.c_void_p)) == 0
    ww = ww.reshape(-1, 16)[: w1.shape[0]][ok1[: 8192]].astype(np.int64)
    good = (ww > 0).all(1); ww = ww[good]
    st = w1[ok1][good][:, 0].astype(np.int64)
    rel = (ww - st[:, None]) * TICK                       # each wave's end, from the chunk's entry
    dur = rel.max(1)
    big = dur > np.percentile(dur, 50)
    print("== the 16 waves of a workgroup (chunks above the median duration, %d of them): last wave done at %.1f us (median); the others, sorted, are done at %s %% of it" % (
        big.sum(), np.median(dur[big]), np.round(np.median(np.sort(rel[big], 1) / dur[big][:, None], 0) * 100).astype(int).tolist()))
    print("   by wave index (median end / last end, %%): %s" % np.round(np.median(rel[big] / dur[big][:, None], 0) * 100).astype(int).tolist())
    print("   wave-time lost to the drain: %.1f %% of the workgroups' wave-time" % (100 * (dur[:, None] - rel).sum() / (16 * dur.sum())))
report("phase 1 (k_pb_scatter)", w1[:, 0], w1[:, 1], w1[:, 3], (w1[:, 2] & np.uint64(0xFFFFFFFF)) * np.uint64(4))
w2 = p2.reshape(-1, 8)


def p2_extra(ok, s, e, t0):
    mid = (w2[:, 1][ok].astype(np.int64) - t0) * TICK
    single = w2[:, 5][ok] != 0
    print("   streaming part: median %.1f us; flush: median %.1f us (single-workgroup bins %d: %.1f us, split bins %d: %.1f us)" % (
        np.median(mid - s), np.median(e - mid), single.sum(), np.median((e - mid)[single]) if single.any() else 0, (~single).sum(), np.median((e - mid)[~single]) if (~single).any() else 0))


report("phase 2 (k_pb_gather)", w2[:, 0], w2[:, 2], w2[:, 3], w2[:, 4] & np.uint64(0xFFFFFFFF), p2_extra)
P.free(); V.free(); G.free()
