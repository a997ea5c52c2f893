"""Offline statistics (CPU, numpy): how many value-stream slots -- distinct (column window, row) pairs -- the
propagation-blocking SpMV needs for R-MAT `scale` under different COLUMN layouts of the message vector.
A 1/`samp` sample of the rows is enough (pairs of different rows never coincide).

  python tools/layout_stats.py --scale 26 --procs 8
"""
import argparse
import multiprocessing as mp
import sys
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from graphtap_amd.rmat import rmat_edges  # noqa: E402


def _part(args):
    scale, first, count, samp = args
    e = rmat_edges(scale, first=first, count=count)
    src, dst = e[:, 0], e[:, 1]
    n = 1 << scale
    outdeg = np.bincount(src, minlength=n).astype(np.uint32)
    indeg = np.bincount(dst, minlength=n).astype(np.uint32)
    h = (dst.astype(np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
    keep = (h >> np.uint64(16)) % np.uint64(samp) == 0
    return outdeg, indeg, src[keep].copy(), dst[keep].copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--samp", type=int, default=16)
    a = ap.parse_args()
    m = 16 << a.scale
    step = 1 << 24
    jobs = [(a.scale, f, min(step, m - f), a.samp) for f in range(0, m, step)]
    n = 1 << a.scale
    outdeg = np.zeros(n, np.uint64)
    indeg = np.zeros(n, np.uint64)
    S, D = [], []
    with mp.Pool(a.procs) as pool:
        for i, (o, d, s, t) in enumerate(pool.imap_unordered(_part, jobs)):
            outdeg += o
            indeg += d
            S.append(s)
            D.append(t)
            if i % 8 == 0:
                print(f"  part {i}/{len(jobs)}", flush=True)
    src = np.concatenate(S)
    dst = np.concatenate(D)
    nnzcols = int((outdeg > 0).sum())
    nnzrows = int((indeg > 0).sum())
    print(f"scale {a.scale}: m={m} nnzcols={nnzcols} nnzrows={nnzrows} sampled entries={len(src)}")
    colid = np.cumsum(outdeg > 0, dtype=np.int64) - 1          # natural compressed column id
    rowid = np.cumsum(indeg > 0, dtype=np.int64) - 1
    r = rowid[dst].astype(np.uint64)
    scale_up = m / len(src)

    def slots(cmap, W, label):
        w = (cmap[src] // W).astype(np.uint64)
        key = (w << np.uint64(32)) | r
        u = np.unique(key).size
        print(f"{label:50s} W={W:6d}: slots ~{u * scale_up / 1e6:8.1f} M  D={len(src) / u:.3f}", flush=True)
        return u

    for W in (8192, 16384, 32768):
        slots(colid, W, "natural")
    # hubs first: the K columns of largest out-degree moved to the front (in degree order), the rest natural
    order = np.argsort(-outdeg.astype(np.int64), kind="stable")   # vertex ids by out-degree descending
    csum = np.cumsum(outdeg[order])
    for K in (8192, 32768, 65536, 131072, 262144, 524288, 1 << 20, 1 << 21, 1 << 22):
        if K > nnzcols:
            break
        print(f"top {K:8d} columns hold {100.0 * csum[K - 1] / m:5.1f}% of the entries; degree at K = {int(outdeg[order[K - 1]])}")
    for K in (8192, 65536, 262144, 1 << 20, 1 << 22):
        if K > nnzcols:
            break
        cmap = np.full(n, -1, np.int64)
        cmap[order[:K]] = np.arange(K)
        rest = np.ones(n, bool)
        rest[order[:K]] = False
        rest &= outdeg > 0
        cmap[rest] = K + np.cumsum(rest, dtype=np.int64)[rest] - 1
        for W in (8192, 16384):
            slots(cmap, W, f"top {K} hubs first")
    cmap = np.full(n, -1, np.int64)
    cmap[order[:nnzcols]] = np.arange(nnzcols)
    for W in (8192, 16384):
        slots(cmap, W, "columns fully degree-sorted")
    # row-degree classes: which rows pay for the slots
    deg_of = indeg[dst]
    for lo, hi in ((1, 4), (4, 16), (16, 64), (64, 256), (256, 1024), (1024, 1 << 30)):
        sel = (deg_of >= lo) & (deg_of < hi)
        print(f"rows of in-degree [{lo},{hi}): {100.0 * sel.sum() / len(src):5.1f}% of the entries")


if __name__ == "__main__":
    main()
