"""Deterministic synthetic R-MAT edge lists (host side, numpy).

The reference ships no generator (its graphs came from Graph500 tooling,
/root/reference/graphtap.slurm:44-45); SURVEY.md section 8d fixes the
parameters: (a,b,c,d) = (0.57,0.19,0.19,0.05), edge factor 16, `scale` bit
levels drawn independently per edge, no permutation, ids 0..2^scale-1,
record = <u4 src, u4 dst[, u4 w]>, weights uniform 1..128 (what
/root/reference/src/misc/converter.cpp:81 produces).

The generator is COUNTER BASED -- edge e of (seed, scale) is a pure function
of (seed, e) through the splitmix64 finaliser -- so that the numpy version
here, and the HIP kernel `gt_rmat_generate` in csrc/ produce bit-identical
edge lists at any size, in any chunking, on any number of ranks.
"""
import numpy as np

# floor(p * 2**32) for the cumulative quadrant probabilities 0.57 / 0.76 / 0.95
T_A = 2448131358
T_AB = 3264175144
T_ABC = 4080218931
GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def rmat_edges(scale, edge_factor=16, seed=1, weighted=False, first=0, count=None):
    """Edges [first, first+count) of the (scale, edge_factor, seed) R-MAT graph as a uint32 array
    of shape (count, 2) or (count, 3)."""
    m = edge_factor << scale
    if count is None:
        count = m - first
    with np.errstate(over="ignore"):
        e = np.arange(first, first + count, dtype=np.uint64)
        base = _mix64(np.uint64(seed) * GOLDEN + e)
        src = np.zeros(count, np.uint64)
        dst = np.zeros(count, np.uint64)
        for level in range(scale):
            u = _mix64(base + np.uint64(level + 1) * GOLDEN) >> np.uint64(32)
            rbit = (u >= np.uint64(T_AB)).astype(np.uint64)
            cbit = (((u >= np.uint64(T_A)) & (u < np.uint64(T_AB))) | (u >= np.uint64(T_ABC))).astype(np.uint64)
            src = (src << np.uint64(1)) | rbit
            dst = (dst << np.uint64(1)) | cbit
        cols = [src.astype(np.uint32), dst.astype(np.uint32)]
        if weighted:
            w = (_mix64(base ^ np.uint64(0xD1B54A32D192ED03)) % np.uint64(128)) + np.uint64(1)
            cols.append(w.astype(np.uint32))
    return np.ascontiguousarray(np.stack(cols, axis=1))
