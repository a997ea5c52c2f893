"""ctypes wrapper around oracle/libgt_oracle.so -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module. The product (graphtap_amd/) must never import it: it fails loudly
when the HIP library is missing instead of falling back to the CPU.

The functions restate the GraphTap reference at np=1; see gt_oracle.c for the
reference file:line each one follows.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgt_oracle.so")
INF = 2147483647

_u32p = C.POINTER(C.c_uint32)
_f64p = C.POINTER(C.c_double)


def build_library():
    """Compile gt_oracle.c with gcc (seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libgt_oracle.so"])


def _load():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "gt_oracle.c")):
        build_library()
    lib = C.CDLL(_SO)
    lib.gto_graph_build.restype = C.c_void_p
    lib.gto_graph_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint32] + [C.c_int] * 5
    lib.gto_graph_free.argtypes = [C.c_void_p]
    for name in ("gto_nnzrows", "gto_nnzcols", "gto_height", "gto_nrows"):
        getattr(lib, name).restype = C.c_uint32
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.gto_nnz.restype = C.c_uint64
    lib.gto_nnz.argtypes = [C.c_void_p]
    for name in ("gto_JA", "gto_IA", "gto_A", "gto_JC", "gto_IR"):
        getattr(lib, name).restype = _u32p
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.gto_class_counts.argtypes = [C.c_void_p, _u32p, _u32p, _u32p]
    lib.gto_spmv_plus_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gto_spmv_min_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gto_degree.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.gto_tcsc_cf_build.restype = C.c_void_p
    lib.gto_tcsc_cf_build.argtypes = [C.c_void_p]
    lib.gto_tcsc_cf_free.argtypes = [C.c_void_p]
    for name in ("gto_cf_IA", "gto_cf_A", "gto_cf_nnz_pairs"):
        getattr(lib, name).restype = _u32p
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.gto_cf_count.restype = C.c_uint32
    lib.gto_cf_count.argtypes = [C.c_void_p, C.c_int]
    for name in ("gto_cf_pairs", "gto_cf_cols"):
        getattr(lib, name).restype = _u32p
        getattr(lib, name).argtypes = [C.c_void_p, C.c_int]
    lib.gto_spmv_cf_plus_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.gto_pagerank.restype = C.c_uint32
    lib.gto_pagerank.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_double, C.c_double,
                                 C.c_void_p, C.c_void_p]
    lib.gto_bfs.restype = C.c_uint32
    lib.gto_bfs.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.gto_sssp.restype = C.c_uint32
    lib.gto_sssp.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.gto_cc.restype = C.c_uint32
    lib.gto_cc.argtypes = [C.c_void_p, C.c_void_p]
    lib.gto_checksum_u32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gto_checksum_f64.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleGraph:
    """Edge list -> TCSC, following mat/graph.hpp:329-356, mat/matrix.hpp:546-555,
    861-1122 and ds/compressed_column.hpp:371-417 (see gt_oracle.c)."""

    def __init__(self, edges, num_vertices, *, weighted=False, directed=True, transpose=False,
                 self_loops=True, acyclic=False, parallel_edges=True):
        edges = np.ascontiguousarray(edges, dtype=np.uint32)
        stride = 3 if weighted else 2
        assert edges.ndim == 2 and edges.shape[1] == stride, edges.shape
        self._h = lib().gto_graph_build(_ptr(edges), edges.shape[0], int(weighted), int(num_vertices),
                                        int(directed), int(transpose), int(self_loops), int(acyclic),
                                        int(parallel_edges))
        if not self._h:
            raise ValueError("vertex id out of range for num_vertices=%d" % num_vertices)
        L = lib()
        self.weighted = weighted
        self.nnz = L.gto_nnz(self._h)
        self.nnzrows = L.gto_nnzrows(self._h)
        self.nnzcols = L.gto_nnzcols(self._h)
        self.H = L.gto_height(self._h)
        self.nrows = L.gto_nrows(self._h)

    def _arr(self, fn, n):
        p = fn(self._h)
        if n == 0 or not p:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    @property
    def JA(self): return self._arr(lib().gto_JA, self.nnzcols + 1)
    @property
    def IA(self): return self._arr(lib().gto_IA, self.nnz)
    @property
    def A(self): return self._arr(lib().gto_A, self.nnz) if self.weighted else None
    @property
    def JC(self): return self._arr(lib().gto_JC, self.nnzcols)
    @property
    def IR(self): return self._arr(lib().gto_IR, self.nnzrows)

    def class_counts(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib().gto_class_counts(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"regular": a.value, "source_rows": b.value, "sink_cols": c.value}

    CF_LISTS = ("REG_R_REG_C", "REG_R_SNK_C", "SRC_R_REG_C", "SRC_R_SNK_C")

    def tcsc_cf(self):
        """The tile in TCSC_CF_BASE form (ds/compressed_column.hpp:419-470, 671-1120): dict of arrays named like the
        reference's members (IA, A, JA_REG_R_NNZ_C, NC_*/JA_*/JC_* of the four lists)."""
        L = lib()
        h = L.gto_tcsc_cf_build(self._h)
        def arr(p, n):
            return np.ctypeslib.as_array(p, shape=(n,)).copy() if n and p else np.zeros(0, np.uint32)
        out = {"IA": arr(L.gto_cf_IA(h), self.nnz), "A": arr(L.gto_cf_A(h), self.nnz) if self.weighted else None,
               "JA_REG_R_NNZ_C": arr(L.gto_cf_nnz_pairs(h), 2 * self.nnzcols)}
        for k, name in enumerate(self.CF_LISTS):
            n = L.gto_cf_count(h, k)
            out["NC_" + name] = n
            out["JA_" + name] = arr(L.gto_cf_pairs(h, k), 2 * n); out["JC_" + name] = arr(L.gto_cf_cols(h, k), n)
        L.gto_tcsc_cf_free(h)
        return out

    def spmv_cf_plus_f64(self, x, y, *, first, running, last):
        """spmv_stationary over the TCSC_CF pair lists (vp:1243-1317); y is accumulated into."""
        x = np.ascontiguousarray(x, np.float64); assert x.shape == (self.nnzcols,)
        assert y.dtype == np.float64 and y.shape == (self.nnzrows,) and y.flags.c_contiguous
        L = lib()
        h = L.gto_tcsc_cf_build(self._h)
        L.gto_spmv_cf_plus_f64(h, _ptr(x), _ptr(y), int(first), int(running), int(last))
        L.gto_tcsc_cf_free(h)
        return y

    def spmv_plus_f64(self, x, y):
        x = np.ascontiguousarray(x, np.float64); assert x.shape == (self.nnzcols,)
        assert y.dtype == np.float64 and y.shape == (self.nnzrows,) and y.flags.c_contiguous
        lib().gto_spmv_plus_f64(self._h, _ptr(x), _ptr(y))
        return y

    def spmv_min_u32(self, x, y):
        x = np.ascontiguousarray(x, np.uint32); assert x.shape == (self.nnzcols,)
        assert y.dtype == np.uint32 and y.shape == (self.nnzrows,) and y.flags.c_contiguous
        lib().gto_spmv_min_u32(self._h, _ptr(x), _ptr(y))
        return y

    def degree(self, order_col):
        d = np.zeros(self.H, np.uint32)
        lib().gto_degree(self._h, int(order_col), _ptr(d))
        return d

    def pagerank(self, degree, iters, *, cf=True, alpha=0.15, tol=1e-5):
        degree = np.ascontiguousarray(degree, np.uint32); assert degree.shape == (self.H,)
        rank = np.zeros(self.H, np.float64); deg = np.zeros(self.H, np.uint32)
        it = lib().gto_pagerank(self._h, _ptr(degree), int(iters), int(cf), alpha, tol, _ptr(rank), _ptr(deg))
        return rank, deg, it

    def bfs(self, root):
        parent = np.zeros(self.H, np.uint32); hops = np.zeros(self.H, np.uint32)
        it = lib().gto_bfs(self._h, int(root), _ptr(parent), _ptr(hops))
        return parent, hops, it

    def sssp(self, root):
        dist = np.zeros(self.H, np.uint32)
        it = lib().gto_sssp(self._h, int(root), _ptr(dist))
        return dist, it

    def cc(self):
        label = np.zeros(self.H, np.uint32)
        it = lib().gto_cc(self._h, _ptr(label))
        return label, it

    def close(self):
        if self._h:
            lib().gto_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def checksum_u32(state, nrows, infinity):
    state = np.ascontiguousarray(state, np.uint32)
    s, c = C.c_uint64(), C.c_uint64()
    lib().gto_checksum_u32(_ptr(state), state.size, int(nrows), int(infinity), C.byref(s), C.byref(c))
    return s.value, c.value


def checksum_f64(state, nrows):
    state = np.ascontiguousarray(state, np.float64)
    s, c = C.c_uint64(), C.c_uint64()
    lib().gto_checksum_f64(_ptr(state), state.size, int(nrows), C.byref(s), C.byref(c))
    return s.value, c.value


# ---------------------------------------------------------------- app drivers
# Flag sets of the reference mains (apps/*.cpp), so tests read like them.
APP_FLAGS = {
    "deg":  dict(directed=True, transpose=False, self_loops=True, acyclic=False, parallel_edges=True),   # deg.cpp:27-31
    "pr":   dict(directed=True, transpose=True, self_loops=True, acyclic=False, parallel_edges=True),    # pr.cpp:26-30
    "bfs":  dict(directed=False, transpose=False, self_loops=False, acyclic=False, parallel_edges=False),  # bfs.cpp:26-30
    "sssp": dict(directed=True, transpose=True, self_loops=False, acyclic=False, parallel_edges=False),   # sssp.cpp:27-38
    "cc":   dict(directed=False, transpose=False, self_loops=True, acyclic=False, parallel_edges=False),   # cc.cpp:24-28
}


def run_app(app, edges, num_vertices, *, iters=20, root=0, cf=True):
    """Run one reference app end to end on the oracle; returns a dict of H-sized arrays."""
    if app == "deg":
        g = OracleGraph(edges, num_vertices, **APP_FLAGS["deg"])
        return {"degree": g.degree(0), "iterations": 1, "graph": g}
    if app == "pr":
        g = OracleGraph(edges, num_vertices, **APP_FLAGS["pr"])
        d = g.degree(1)
        rank, deg, it = g.pagerank(d, iters, cf=cf)
        return {"rank": rank, "degree": deg, "iterations": it, "graph": g}
    if app == "bfs":
        g = OracleGraph(edges, num_vertices, **APP_FLAGS["bfs"])
        parent, hops, it = g.bfs(root)
        return {"parent": parent, "hops": hops, "iterations": it, "graph": g}
    if app == "sssp":
        g = OracleGraph(edges, num_vertices, weighted=True, **APP_FLAGS["sssp"])
        dist, it = g.sssp(root)
        return {"distance": dist, "iterations": it, "graph": g}
    if app == "cc":
        g = OracleGraph(edges, num_vertices, **APP_FLAGS["cc"])
        label, it = g.cc()
        return {"label": label, "iterations": it, "graph": g}
    raise ValueError(app)
