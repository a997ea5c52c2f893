#!/usr/bin/env python3
"""Generate tests/golden/tcsc_cf.npz by RUNNING THE UNMODIFIED REFERENCE (np = 1): the tile's TCSC_CF_BASE arrays
(/root/reference/src/ds/compressed_column.hpp:419-470 -- IA with the source rows of every column swapped to its tail,
:671-708, and the four pair lists, :710-1120) as oracle/_ref/dump_tcsc_cf[_w] (oracle/ref/dump_main.cpp, -DAPP_TCSC_CF)
writes them for a graph loaded with the flags of src/apps/pr.cpp.

Runs only in the build container (needs /root/reference and `make -C oracle/ref`). The fixture (inputs + the reference's
arrays) is committed; nothing here is used at test time."""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
MPIRUN = "/opt/conda/bin/mpirun"
ENV = dict(os.environ, PATH=os.path.join(REF, "fileshim") + ":" + os.environ["PATH"])
LISTS = ("REG_R_REG_C", "REG_R_SNK_C", "SRC_R_REG_C", "SRC_R_SNK_C")


def mixed_graph():
    """48 vertices in three groups -- only in-edges (ids 0-15: the source ROWS of the transposed matrix, ahead of the
    regular rows in every column, so that the tail swap has work to do), both (16-31), only out-edges (32-47: sink
    columns) -- with enough parallel picks that columns mix the two kinds of rows in every order; 47 = N has an edge."""
    rng = np.random.RandomState(11)
    src = rng.randint(16, 48, 420); dst = rng.randint(0, 32, 420)
    return np.stack([src, dst], 1).astype(np.uint32), None, 47


def read_dump(path):
    raw = open(path, "rb").read()
    magic, weighted, nnz, nc, nr = struct.unpack("<IIQII", raw[:24])
    assert magic == 0x46435447
    pos = [24]
    def take(n):
        a = np.frombuffer(raw, "<u4", n, pos[0]).copy(); pos[0] += 4 * n; return a
    out = {"IA": take(nnz)}
    if weighted: out["A"] = take(nnz)
    out["JA"] = take(nc + 1); out["JC"] = take(nc); out["IR"] = take(nr); out["JA_REG_R_NNZ_C"] = take(2 * nc)
    for name in LISTS:
        n = int(take(1)[0])
        out["NC_" + name] = np.uint32(n); out["JA_" + name] = take(2 * n); out["JC_" + name] = take(n)
    assert pos[0] == len(raw)
    return out


def main():
    tmp = tempfile.mkdtemp(prefix="gtcf")
    cases = {}
    for name in ("tiny", "rmat8", "rmat12"):
        z = np.load(os.path.join(HERE, name + ".npz"))
        cases[name] = (z["edges"], z["wedges"], int(z["num_vertices"]))
    cases["rmat10"] = (np.fromfile(os.path.join(HERE, "rmat10_1024.bin"), "<u4").reshape(-1, 2),
                       np.fromfile(os.path.join(HERE, "rmat10_1024_w.bin"), "<u4").reshape(-1, 3), 1024)
    cases["mixed"] = mixed_graph()
    out = {}
    for name, (e, w, nv) in cases.items():
        if name == "mixed":
            out["mixed_edges"] = e; out["mixed_num_vertices"] = np.uint32(nv)
        # unweighted only: with weights the reference orders a column by (weight, unstable), ds/triple.hpp:83-92 -- the
        # order of IA inside a column is then libstdc++'s, see the header of oracle/gt_oracle.c
        for tag, arr, exe in (("u", e, "dump_tcsc_cf"),):
            f = os.path.join(tmp, "%s_%s.bin" % (name, tag)); np.ascontiguousarray(arr, "<u4").tofile(f)
            o = os.path.join(tmp, "%s_%s" % (name, tag))
            subprocess.run([MPIRUN, "-np", "1", os.path.join(REF, exe), f, str(nv), o], env=ENV, check=True, capture_output=True)
            d = read_dump(o + ".cf.bin")
            for k, v in d.items():
                out["%s_%s_%s" % (name, tag, k)] = v
            print(name, tag, "nnz", len(d["IA"]), {k: int(d["NC_" + k]) for k in LISTS})
    np.savez_compressed(os.path.join(HERE, "tcsc_cf.npz"), **out)


if __name__ == "__main__":
    main()
