"""Host-side mirror of the reference's Graph<> (src/mat/graph.hpp) over the C ABI.

`Graph.load` keeps the reference's argument list (graph.hpp:41-43) so application code reads
like src/apps/*.cpp; the graph itself is built and kept in HBM by libgraphtap_amd.so."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import ExchangePlan, GraphFlags, GraphInfo, GraphTapError, TileArrays, TileCFArrays, check, lib

# Tiling_type (src/mat/tiling.hpp:13-16) / Compression_type (src/ds/compressed_column.hpp)
_2D_, _2DT_ = 0, 1
_CSC_, _DCSC_, _TCSC_, _TCSC_CF_ = 0, 1, 2, 3


def world():
    """(rank, nranks) of the torch.distributed job if one is initialised, else (0, 1)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def read_edge_file(path, weighted):
    """Binary edge list as the reference reads it (graph.hpp:308-336): records of two (three with
    weights) little-endian u32. The reference tells text from binary with popen("file -b")
    (graph.hpp:119-145); here a file is text when it decodes as ASCII digits/space/comment lines."""
    size = os.path.getsize(path)
    stride = 3 if weighted else 2
    with open(path, "rb") as f:
        head = f.read(4096)
    is_text = len(head) > 0 and all((32 <= b < 127) or b in (9, 10, 13) for b in head)
    if is_text:
        # parread_text (graph.hpp:195-304), the same rule as include/graphtap_amd.hpp parse_text: leading '#' / '%' / empty
        # lines are skipped, every other line is "row col[ weight]" separated by single spaces (another column count, a
        # non-digit or an id above 2^32 - 1 is "read() failure", :250-257), the list ends at the first empty line; comment
        # lines further down are skipped too.
        rows, started = [], False
        with open(path, "rb") as f:
            for raw in f.read().split(b"\n"):
                line = raw[:-1] if raw.endswith(b"\r") else raw
                if not line:
                    if started:
                        break
                    continue
                if line[:1] in (b"#", b"%"):
                    continue
                started = True
                parts = line.split(b" ")
                if len(parts) != stride or not all(t.isdigit() and int(t) <= 0xFFFFFFFF for t in parts):
                    raise GraphTapError('read() failure "%s"' % line.decode("ascii", "replace"))
                rows.append([int(t) for t in parts])
        return np.asarray(rows, dtype=np.uint32).reshape(-1, stride)
    if size % (4 * stride) != 0:
        raise GraphTapError("read() failure: %s is not a whole number of %d-byte records" % (path, 4 * stride))
    return np.fromfile(path, dtype="<u4").reshape(-1, stride)


class Graph:
    """Graph<Weight, Integer_Type, Fractional_Type> (src/mat/graph.hpp:31-71)."""

    def __init__(self, weighted=False, options=None):
        self.weighted = bool(weighted)   # the reference fixes this at compile time (-DHAS_WEIGHT)
        self.options = options           # _lib.GraphOptions, or None: the environment variables / defaults
        self._h = None
        self.info = None
        self.flags = None
        self.compression_type = _TCSC_
        self.rank, self.nranks = 0, 1
        self.exchange = False   # the message vector is filled by an exchange (several ranks, or GRAPHTAP_FORCE_EXCHANGE on one)

    # -- Graph::load (graph.hpp:105-148)
    def load(self, filepath, nrows, ncols, directed=True, transpose=False, self_loops=True, acyclic=False,
             parallel_edges=True, tiling_type=_2DT_, compression_type=_CSC_):
        edges = read_edge_file(filepath, self.weighted)
        return self.load_edges(edges, nrows, ncols, directed, transpose, self_loops, acyclic, parallel_edges,
                               tiling_type, compression_type)

    def load_edges(self, edges, nrows, ncols, directed=True, transpose=False, self_loops=True, acyclic=False,
                   parallel_edges=True, tiling_type=_2DT_, compression_type=_CSC_, rank=None, nranks=None):
        """Same as load() for an edge array already in host memory (uint32, shape (m, 2|3))."""
        edges = np.ascontiguousarray(edges, dtype=np.uint32)
        stride = 3 if self.weighted else 2
        if edges.ndim != 2 or edges.shape[1] != stride:
            raise GraphTapError("edge array must have shape (m, %d)" % stride)
        return self._build(edges.ctypes.data_as(C.c_void_p), edges.shape[0], 0, nrows, ncols, directed, transpose,
                           self_loops, acyclic, parallel_edges, compression_type, rank, nranks)

    def load_device(self, dev_ptr, m, nrows, ncols, directed=True, transpose=False, self_loops=True, acyclic=False,
                    parallel_edges=True, tiling_type=_2DT_, compression_type=_CSC_, rank=None, nranks=None):
        """Edge records already resident in HBM (bench.py: the synthetic generator writes them there)."""
        return self._build(C.c_void_p(dev_ptr), m, 1, nrows, ncols, directed, transpose, self_loops, acyclic,
                           parallel_edges, compression_type, rank, nranks)

    def load_share(self, dist_handle, rank, nranks, edges_share, nrows, ncols, directed=True, transpose=False, self_loops=True, acyclic=False,
                   parallel_edges=True, tiling_type=_2DT_, compression_type=_TCSC_, m_share=None):
        """Matrix::distribute (mat/matrix.hpp:693-810): this rank's SHARE of the edge list (any split; host array [m, 2 or 3] of
        uint32, or -- with `m_share` -- the address of that many records already in HBM); the build shuffles the records to the
        owners of their rows over the communicator. Collective: every rank of `dist_handle` (a gt_dist; `rank` / `nranks`
        are its) calls it."""
        self._validate(nrows, ncols, compression_type)
        if m_share is None:
            edges = np.ascontiguousarray(edges_share, dtype=np.uint32)
            ptr, m, on_device = edges.ctypes.data_as(C.c_void_p), edges.shape[0], 0
        else:
            ptr, m, on_device = C.c_void_p(int(edges_share)), int(m_share), 1
        self.free()
        self.rank, self.nranks = rank, nranks
        self.exchange = nranks > 1 or os.environ.get("GRAPHTAP_FORCE_EXCHANGE", "0") not in ("", "0")
        self.tiling_type, self.compression_type = tiling_type, compression_type
        self.flags = GraphFlags(int(directed), int(transpose), int(self_loops), int(acyclic), int(parallel_edges))
        h = C.c_void_p()
        check(lib().gt_graph_build_distributed(C.byref(h), dist_handle, ptr, m, on_device, int(self.weighted),
                                               int(nrows), C.byref(self.flags)))
        self._h = h
        self.info = GraphInfo()
        check(lib().gt_graph_info_get(self._h, C.byref(self.info)))
        # the TEPS denominators and display() are global figures: summed over the communicator the build ran on
        w = (C.c_uint64 * 2)(int(self.info.nnz_local), int(self.info.nnzrows))
        check(lib().gt_dist_all_reduce_u64(dist_handle, w, 2))
        self.nnz_global, self.nnzrows_global = int(w[0]), int(w[1])
        return self

    @staticmethod
    def _validate(nrows, ncols, compression_type):
        if nrows != ncols:
            raise GraphTapError("square matrices only (every reference app passes num_vertices twice)")
        if compression_type not in (_TCSC_, _TCSC_CF_):
            raise GraphTapError("only TCSC / TCSC_CF tiles exist in this engine (all reference apps use them)")

    def _build(self, ptr, m, on_device, nrows, ncols, directed, transpose, self_loops, acyclic, parallel_edges,
               compression_type, rank, nranks):
        self._validate(nrows, ncols, compression_type)
        if rank is None:
            rank, nranks = world()
        self.free()
        self.rank, self.nranks = rank, nranks
        self.exchange = nranks > 1 or (bool(self.options.force_exchange) if self.options is not None and self.options.force_exchange >= 0
                                       else os.environ.get("GRAPHTAP_FORCE_EXCHANGE", "0") not in ("", "0"))
        self.compression_type = compression_type
        self.flags = GraphFlags(int(directed), int(transpose), int(self_loops), int(acyclic), int(parallel_edges))
        h = C.c_void_p()
        if self.options is not None:   # handle-level configuration (gt_graph_options) instead of the environment
            check(lib().gt_graph_build_opt(C.byref(h), None, ptr, m, on_device, int(self.weighted), int(nrows),
                                           C.byref(self.flags), rank, nranks, C.byref(self.options)))
        else:
            check(lib().gt_graph_build(C.byref(h), ptr, m, on_device, int(self.weighted), int(nrows),
                                       C.byref(self.flags), rank, nranks))
        self._h = h
        self.info = GraphInfo()
        check(lib().gt_graph_info_get(self._h, C.byref(self.info)))
        self.nnz_global = int(self.info.nnz_local)
        self.nnzrows_global = int(self.info.nnzrows_global)
        if nranks > 1 and world()[1] == nranks:   # explicit rank/nranks without a process group: caller sums
            import torch
            import torch.distributed as dist
            t = torch.tensor([self.nnz_global, int(self.info.nnzrows)], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t)
            self.nnz_global, self.nnzrows_global = int(t[0]), int(t[1])
        return self

    def vertex_ids(self):
        """Original vertex id of every state slot of this rank (0xFFFFFFFF = padding slot). Identity on one rank; on
        several ranks the owned segment is a range of a hashed internal id space (gt_graph_vertex_ids)."""
        H = self.info.tile_height
        a = np.zeros(H, np.uint32)
        check(lib().gt_graph_vertex_ids(self._h, a.ctypes.data_as(C.c_void_p), H))
        return a

    def tile(self):
        t = TileArrays()
        check(lib().gt_graph_tile(self._h, C.byref(t)))
        return t

    def tile_to_host(self):
        """Copies the owned tile-row's TCSC arrays to numpy (tests compare them with the oracle's)."""
        t, i = self.tile(), self.info
        ncols_total = i.ncols_local

        def grab(ptr, n):
            a = np.zeros(n, np.uint32)
            if n and ptr:
                check(lib().gt_memcpy_d2h(a.ctypes.data_as(C.c_void_p), ptr, n * 4))
            return a
        return dict(JA=grab(t.JA, ncols_total + 1), IA=grab(t.IA, i.nnz_local),
                    A=grab(t.A, i.nnz_local) if t.A else None, JC=grab(t.JC, i.nnzcols), IR=grab(t.IR, i.nnzrows),
                    L2G=grab(t.L2G, ncols_total) if t.L2G else None)

    CF_LISTS = ("REG_R_REG_C", "REG_R_SNK_C", "SRC_R_REG_C", "SRC_R_SNK_C")   # gt_cf_list

    def tile_cf(self):
        """gt_graph_tile_cf: the tile in TCSC_CF form (device pointers; built on first use)."""
        t = TileCFArrays()
        check(lib().gt_graph_tile_cf(self._h, C.byref(t)))
        return t

    def tile_cf_to_host(self):
        """The TCSC_CF arrays as numpy, named like TCSC_CF_BASE's members (ds/compressed_column.hpp:419-470)."""
        t, i = self.tile_cf(), self.info

        def grab(ptr, n):
            a = np.zeros(n, np.uint32)
            if n and ptr:
                check(lib().gt_memcpy_d2h(a.ctypes.data_as(C.c_void_p), ptr, n * 4))
            return a
        out = dict(IA=grab(t.IA, i.nnz_local), A=grab(t.A, i.nnz_local) if t.A else None, JA_REG_R_NNZ_C=grab(t.JA_REG_R_NNZ_C, 2 * i.nnzcols))
        for k, name in enumerate(self.CF_LISTS):
            out["NC_" + name] = int(t.NC[k])
            out["JA_" + name] = grab(t.JA[k], 2 * int(t.NC[k])); out["JC_" + name] = grab(t.JC[k], int(t.NC[k]))
        return out

    def exchange_plan(self):
        """gt_graph_exchange_plan as python lists: (send_offset[K+1], recv_offset[K+1], send_counts[K][p], recv_counts[K][p])."""
        pl = ExchangePlan()
        check(lib().gt_graph_exchange_plan(self._h, C.byref(pl)))
        K, p = pl.x_slices, pl.nranks
        return ([int(pl.send_offset[k]) for k in range(K + 1)], [int(pl.recv_offset[k]) for k in range(K + 1)],
                [[int(pl.send_counts[k * p + d]) for d in range(p)] for k in range(K)],
                [[int(pl.recv_counts[k * p + d]) for d in range(p)] for k in range(K)])

    # -- Graph::free (graph.hpp:76-81)
    def free(self):
        if self._h:
            check(lib().gt_graph_free(self._h))
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
