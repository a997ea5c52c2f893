#!/usr/bin/env python3
"""Bytes per rank and iteration of the two partitions SURVEY 8(e) allows, counted on the actual R-MAT edge list (CPU, numpy):

  1-D  tile-rows (a = 1, b = p; what this build ships): rank k owns vertex segment k and every entry whose ROW lies in it; it
       receives the messages of the columns it has an entry in (needed columns) from their owners; y is complete locally.
  2-D  the reference's grid (a x b = 2 x 4 at p = 8, src/mat/tiling.hpp:36-72, matrix.hpp:273-341): a rank holds p/b row groups x
       p/a column groups of the p x p tile grid; it receives the messages of its p/a column groups (broadcast in the column group of
       b ranks) and, for the row groups it does not lead, sends its partial accumulators to the leader (reduce in the row group of
       a ranks), who applies.

Both with the hashed internal ids of the build (u = v * 0x9E3779B1 mod 2^k), messages of F_x bytes, accumulators of F_y bytes,
and -- for the 2-D case -- both the dense form the reference ships (whole segments) and a needed-columns / touched-rows form.
  python tools/partition_model.py --scale 22 --p 8"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphtap_amd.rmat import rmat_edges

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22); ap.add_argument("--p", type=int, default=8)
ap.add_argument("--fx", type=int, default=4); ap.add_argument("--fy", type=int, default=8)
a = ap.parse_args()
p, nv = a.p, 1 << a.scale
e = rmat_edges(a.scale, 16, 1)
mask = nv * 2 - 1 if (nv + 1) > nv else nv - 1   # the build hashes over the next power of two >= nrows = nv + 1
M = 1 << int(np.ceil(np.log2(nv + 1)))
u = lambda v: (v.astype(np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(M - 1)
src, dst = u(e[:, 0]), u(e[:, 1])          # PageRank runs on the transpose: row = dst, col = src
H = M // p + 1
rseg, cseg = (dst // H).astype(np.int64), (src // H).astype(np.int64)
out = {"scale": a.scale, "p": p, "F_x": a.fx, "F_y": a.fy}
# ---- 1-D: needed columns per tile-row
need1 = []
for k in range(p):
    cols = np.unique(src[rseg == k])
    own = (cols // H) == k
    need1.append(int((~own).sum()))
out["1D_recv_bytes_per_rank_mean"] = int(np.mean(need1) * a.fx)
# ---- 2-D: a x b with a * b = p, a = floor(sqrt(p)) adjusted like integer_factorize (tiling.hpp:65-73)
ar = int(np.sqrt(p)); br = ar
while ar * br != p:
    br += 1; ar = p // br
# rank r = (i mod b) * a + (j mod a) holds tiles (i, j): row groups {i: i mod b == r // a}, column groups {j: j mod a == r % a}
recv_x_dense, recv_x_need, red_y_dense, red_y_touch = [], [], [], []
nnzcols_seg = [len(np.unique(src[cseg == j])) for j in range(p)]
nnzrows_seg = [len(np.unique(dst[rseg == i])) for i in range(p)]
for r in range(p):
    rows = [i for i in range(p) if i % br == r // ar]
    cols = [j for j in range(p) if j % ar == r % ar]
    inr = np.isin(rseg, rows); inc = np.isin(cseg, cols)
    mine = inr & inc
    # x: the column groups it holds minus the one segment it leads (it computes that one itself)
    lead = r   # every rank leads one diagonal segment (matrix.hpp:327-341); which one does not change the volumes
    recv_x_dense.append(sum(nnzcols_seg[j] for j in cols) - nnzcols_seg[cols[0]])
    need = np.unique(src[mine]); recv_x_need.append(int(len(need) * (len(cols) - 1) / len(cols)))
    # y: partial accumulators of the row groups it holds but does not lead go to their leaders
    red_y_dense.append(sum(nnzrows_seg[i] for i in rows) - nnzrows_seg[rows[0]])
    touched = np.unique(dst[mine]); red_y_touch.append(int(len(touched) * (len(rows) - 1) / len(rows)))
out["2D_grid"] = "%d x %d" % (ar, br)
out["2D_recv_x_bytes_dense_segments"] = int(np.mean(recv_x_dense) * a.fx)
out["2D_recv_x_bytes_needed_columns"] = int(np.mean(recv_x_need) * a.fx)
out["2D_reduce_y_bytes_dense_segments"] = int(np.mean(red_y_dense) * a.fy)
out["2D_reduce_y_bytes_touched_rows"] = int(np.mean(red_y_touch) * a.fy)
out["2D_total_dense"] = out["2D_recv_x_bytes_dense_segments"] + out["2D_reduce_y_bytes_dense_segments"]
out["2D_total_needed"] = out["2D_recv_x_bytes_needed_columns"] + out["2D_reduce_y_bytes_touched_rows"]
out["ratio_2D_needed_over_1D"] = round(out["2D_total_needed"] / max(out["1D_recv_bytes_per_rank_mean"], 1), 3)
print(json.dumps(out))
