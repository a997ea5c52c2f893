"""Host-side (numpy) restatements of single steps, used by the full-size GPU tests to check the engine against something
that is neither the engine nor -- at sizes where the C oracle would need tens of GB -- the oracle: they work on the raw RECORD
STREAM in slices. Each one is itself pinned to the oracle by a CPU test (tests/test_oracle_golden.py)."""
import numpy as np


def pagerank_step_from_records(record_slices, rank_prev, degree, alpha=0.15):
    """One PageRank iteration (src/apps/pr.h:31-47) straight from edge records <u4 a, u4 b>, under apps/pr.cpp's flags
    (directed, TRANSPOSE, self loops and parallel edges kept: mat/graph.hpp:337-356 swaps the pair): the stored entry of a
    record (a, b) is (row b, col a) -- vertex b gathers from vertex a -- so
        rank_new[b] = alpha + (1 - alpha) * sum over records (a, b) of rank_prev[a] / degree[a],
    with the messenger's 0 for a column of degree 0 (pr.h:33-37). `degree` is what the Degree program leaves in the PageRank
    state: the entries of the vertex's column IF the vertex has a row, else 0 (SURVEY 8a traps 1, 2) -- taken from the run under
    test, and equal to the oracle's wherever the two are compared. Sums are taken in f64, record by record (np.bincount),
    slice after slice. A vertex that is no record's `b` gets alpha + 0 = alpha, which is also what the reference leaves
    there (it never applies a vertex without a row, and alpha is the initial rank)."""
    n = rank_prev.size
    with np.errstate(divide="ignore", invalid="ignore"):
        msg = np.where(degree > 0, rank_prev / np.maximum(degree, 1).astype(np.float64), 0.0)
    y = np.zeros(n, np.float64)
    for h in record_slices:
        a, b = h[:, 0].astype(np.int64), h[:, 1].astype(np.int64)
        y += np.bincount(b, weights=msg[a], minlength=n)[:n]
    return alpha + (1.0 - alpha) * y
