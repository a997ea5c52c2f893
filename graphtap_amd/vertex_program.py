"""Host-side mirror of the reference's Vertex_Program<> surface (src/vp/vertex_program.hpp:23-62)
and of the five programs in src/apps/{deg,pr,bfs,sssp,cc}.h, over the C ABI.

Same names, argument meaning and call order as the reference, so application code and tests read
like src/apps/*.cpp:

    G = Graph(); G.load(path, n, n, directed, transpose, self_loops, acyclic, parallel_edges, TT, CT)
    V = Deg_Program(G, stationary, gather_depends_on_apply, apply_depends_on_iter, _COL_); V.execute(1)
    VR = PR_Program(G, stationary, ..., _ROW_); VR.initialize(V); VR.execute(20); VR.checksum(); VR.display()

The virtual per-vertex / per-edge hooks of the reference cannot run on a device, so each program is
an op-code (`kind`) whose hooks are HIP kernels in csrc/engine.hip; there is no CPU path."""
import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import (GT_BFS, GT_CC, GT_COL, GT_DEG, GT_INF, GT_PR, GT_ROW, GT_SSSP, GT_TCSC, GT_TCSC_CF, ExecStats,
                   GraphTapError, ProgramParams, check, lib)
from .graph import _TCSC_CF_

_ROW_, _COL_ = GT_ROW, GT_COL   # Ordering_type, vp:17-21


def _native_handle():
    from . import dist_native
    return dist_native.handle()
INF = GT_INF


class _CudaArray:
    """Zero-copy view of engine-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class _HipEngine:
    """The five phase calls of graphtap_amd.dist.run over the C ABI (device buffers stay in HBM)."""

    def __init__(self, prog):
        self.prog = prog
        i = prog.G.info
        self.rank, self.nranks, self.seg_stride = i.rank, i.nranks, i.seg_stride
        self.x_slices, self.slice_width = i.x_slices, i.slice_width
        self.force_exchange = prog.G.exchange and i.nranks == 1
        self.column_accumulators = (prog.ordering_type == _COL_)
        self.needs_x_exchange = not self.column_accumulators
        self._x = self._y = self._send = None
        self.stream = None          # the HIP stream the program launches on (set by Vertex_Program.execute)

    def _tensor(self, getter):
        import torch
        ptr, n, w = C.c_void_p(), C.c_uint64(), C.c_uint32()
        check(getter(self.prog._h, C.byref(ptr), C.byref(n), C.byref(w)))
        return torch.as_tensor(_CudaArray(ptr.value, n.value, self._typestr(w.value)), device="cuda")

    def _typestr(self, width):
        # PageRank messages are floating point (f64, or f32 under GT_SPMV_PB_F32MSG); everything else is u32 viewed as i32
        if self.prog.kind == GT_PR:
            return "<f8" if width == 8 else "<f4"
        return "<i4"

    def _installed(self, getter, setter):
        """A torch allocation with the engine buffer's contents, installed into the engine, so that RCCL sees
        ordinary caching-allocator memory."""
        import torch
        ptr, n, w = C.c_void_p(), C.c_uint64(), C.c_uint32()
        check(getter(self.prog._h, C.byref(ptr), C.byref(n), C.byref(w)))
        if not n.value:
            return torch.zeros(0, dtype={"<f8": torch.float64, "<f4": torch.float32, "<i4": torch.int32}[self._typestr(w.value)], device="cuda")
        t = torch.as_tensor(_CudaArray(ptr.value, n.value, self._typestr(w.value)), device="cuda").clone()
        check(setter(self.prog._h, C.c_void_p(t.data_ptr())))
        return t

    def prepare(self, iters):
        """gt_program_prepare: the first call of an execute(iters). It may change the element width of the message buffers
        (PageRank with f32 messages runs f64 ones in converge mode): tensors installed for the other width are taken back."""
        def width():
            w = C.c_uint32()
            check(lib().gt_program_x(self.prog._h, None, None, C.byref(w)))
            return w.value
        before = width()
        if self._x is not None and self.prog.G.exchange:
            check(lib().gt_program_set_x(self.prog._h, None))
        if self._send is not None:
            check(lib().gt_program_set_send(self.prog._h, None))
        check(lib().gt_program_prepare(self.prog._h, iters))
        if width() != before:
            self._x = self._send = None
        else:   # same width: the installed tensors stay
            if self._x is not None and self.prog.G.exchange:
                check(lib().gt_program_set_x(self.prog._h, C.c_void_p(self._x.data_ptr())))
            if self._send is not None and self._send.numel():
                check(lib().gt_program_set_send(self.prog._h, C.c_void_p(self._send.data_ptr())))

    def x_tensor(self):
        """The message vector the local SpMV reads (receive side of the exchange on several ranks); the
        single-rank case views the engine's own buffer."""
        if self._x is None:
            if self.prog.G.exchange:
                self._x = self._installed(lib().gt_program_x, lib().gt_program_set_x)
            else:
                self._x = self._tensor(lib().gt_program_x)
        return self._x

    def send_tensor(self):
        """Several ranks: the per-destination packing of the owned columns' messages (gt_program_send)."""
        if self._send is None:
            self._send = self._installed(lib().gt_program_send, lib().gt_program_set_send)
        return self._send

    def exchange_plan(self):
        return self.prog.G.exchange_plan()

    def y_tensor(self):
        if self._y is None:
            self._y = self._tensor(lib().gt_program_y)
        return self._y

    @property
    def iteration(self):
        return self.prog.iteration

    def scatter_gather(self):
        check(lib().gt_program_scatter_gather(self.prog._h))

    def combine(self):
        check(lib().gt_program_combine(self.prog._h))

    def combine_slice(self, k):
        check(lib().gt_program_combine_slice(self.prog._h, k))

    def arm_fused_apply(self, iters, want_active):
        """apply(iters, want_active) follows the next combine at once: phase 2 may run PageRank's applicator itself"""
        check(lib().gt_program_fuse_apply(self.prog._h, iters, int(bool(want_active))))

    def apply(self, iters, want_active):
        a = C.c_uint64(0)
        check(lib().gt_program_apply(self.prog._h, iters, C.byref(a) if want_active else None))
        return a.value

    def finish_converged(self):
        check(lib().gt_program_finish_converged(self.prog._h))


class Vertex_Program:
    """Vertex_Program(Graph&, stationary, gather_depends_on_apply, apply_depends_on_iter, Ordering_type), vp:27-29."""
    kind = None
    fields = ()            # (name, gt_field, dtype) of the Vertex_State struct
    state_field = None     # get_state()
    infinity = 0           # infinity(), vp:40

    def __init__(self, Graph, stationary=False, gather_depends_on_apply=False, apply_depends_on_iter=False,
                 ordering_type=_ROW_):
        if type(self) is Vertex_Program:
            raise GraphTapError("subclass one of the five programs: device code cannot call user virtuals")
        self.G = Graph
        self.stationary = stationary
        self.gather_depends_on_apply = gather_depends_on_apply
        self.apply_depends_on_iter = apply_depends_on_iter
        self.ordering_type = ordering_type
        self.num_iterations = 0
        self.converged = False
        self.root = 0
        self.alpha, self.tol = 0.15, 1e-5          # pr.h:12-13
        self.stats = None
        self._h = None
        self._already_initialized = False
        want = self.kind in (GT_DEG, GT_PR)
        if bool(stationary) != want:
            raise GraphTapError("%s is %sstationary in the reference (src/apps)" % (type(self).__name__, "" if want else "non-"))

    # -- lazily create the device program (root / alpha / tol are plain members in the reference,
    #    set after construction: bfs.cpp:46)
    def _handle(self):
        if self._h is None:
            prm = ProgramParams(self.kind, self.ordering_type,
                                GT_TCSC_CF if self.G.compression_type == _TCSC_CF_ else GT_TCSC,
                                int(self.root), float(self.alpha), float(self.tol))
            h = C.c_void_p()
            check(lib().gt_program_create(C.byref(h), self.G._h, C.byref(prm)))
            self._h = h
        return self._h

    @property
    def iteration(self):
        it = C.c_uint32()
        check(lib().gt_program_iteration(self._handle(), C.byref(it)))
        return it.value

    def set_options(self, options=None, **kw):
        """gt_program_set_options: per-program configuration (a _lib.ProgramOptions, or its fields as keywords)."""
        from ._lib import ProgramOptions
        o = options if options is not None else ProgramOptions(**kw)
        check(lib().gt_program_set_options(self._handle(), C.byref(o)))

    # -- initialize(), vp:443-464 / initialize(other), vp:466-501
    def initialize(self, other=None):
        eng = getattr(self, "_engine", None)
        h = self._handle()
        w0, w1 = C.c_uint32(), C.c_uint32()
        check(lib().gt_program_x(h, None, None, C.byref(w0)))
        if other is None:
            check(lib().gt_program_initialize(h))
        else:
            check(lib().gt_program_initialize_from(h, other._handle()))
        check(lib().gt_program_x(h, None, None, C.byref(w1)))
        self._already_initialized = True
        if eng is not None:
            eng.check_sticky = False
            if w0.value != w1.value:
                # initialize() took a PageRank program back from the f64 messages of a converge-mode run to its f32 ones and
                # un-installed the driver's buffers (engine.hip, init_common): tensors of the other width must not be exchanged
                eng._x = eng._send = None

    # -- execute(num_iterations = 0), vp:408-441
    def execute(self, num_iterations=0):
        self.num_iterations = int(num_iterations)
        h = self._handle()
        if not self._already_initialized:
            self.initialize()
        if not self.G.exchange:
            st = ExecStats()
            check(lib().gt_program_execute(h, self.num_iterations, C.byref(st)))
            self.stats = st
            self.converged = bool(st.converged)
        elif _native_handle() is not None:   # the C++ driver (csrc/dist.hip): RCCL called from the library
            from . import dist_native
            _, self.converged, self.stats = dist_native.execute(self, self.num_iterations)
        else:
            import torch
            from . import dist as gdist
            check(lib().gt_program_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            if getattr(self, "_engine", None) is None:
                self._engine = _HipEngine(self)
            self._engine.stream = torch.cuda.current_stream().cuda_stream
            _, self.converged = gdist.run(self._engine, self.num_iterations)
        return self

    # -- V, vp:61 (owned segment; struct-of-arrays)
    @property
    def V(self):
        H = self.G.info.tile_height
        out = {}
        for name, field, dtype in self.fields:
            a = np.zeros(H, dtype)
            check(lib().gt_program_copy_state(self._handle(), field, a.ctypes.data_as(C.c_void_p), H))
            out[name] = a
        return out

    def get_vid(self, index):   # vp:1805-1808 (one rank); several ranks: through the graph's slot -> vertex map
        if self.G.nranks == 1:
            return index + self.G.info.rank * self.G.info.tile_height
        return int(self.G.vertex_ids()[index])

    def gather_global(self):
        """{field: array over original vertex ids 0..nrows-1} assembled from every rank's V (all ranks get it)."""
        V, vids = self.V, self.G.vertex_ids()
        keep = vids != 0xFFFFFFFF
        n = self.G.info.nrows
        parts = [(vids[keep], {k: v[keep] for k, v in V.items()})]
        if self.G.nranks > 1:
            import torch.distributed as dist
            allp = [None] * self.G.nranks
            dist.all_gather_object(allp, parts[0])
            parts = allp
        out = {name: np.zeros(n, dtype) for name, _, dtype in self.fields}
        for ids, vals in parts:
            for k in out:
                out[k][ids] = vals[k]
        return out

    # -- checksum(), vp:1927-1960 (the accumulator is uint64_t: an fp state truncates at every add)
    def checksum(self, out=sys.stdout):
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().gt_program_checksum(self._handle(), C.byref(a), C.byref(b)))
        s, cnt = a.value, b.value
        if self.G.nranks > 1:
            import torch
            import torch.distributed as dist
            t = torch.tensor([s, cnt], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t)
            s, cnt = int(t[0]), int(t[1])
        self.value_checksum, self.reachable = s, cnt
        if self.G.rank == 0 and out is not None:
            print("Iterations: %d" % self.iteration, file=out)
            print("Value checksum: %d" % s, file=out)
            print("Reachable vertices: %d" % cnt, file=out)
        return s, cnt

    def print_state(self, V, i):
        raise NotImplementedError

    # -- display(count = 31), vp:2124-2181
    def display(self, count=31, out=sys.stdout):
        if self.G.nranks == 1:
            V = self.V
            count = min(count, self.G.info.tile_height)
        else:   # the reference prints rank 0's first states (vertices 0..30 at np=1); here: vertices 0..count-1
            V = self.gather_global()
            count = min(count, self.G.info.nrows)
        lines = ["vertex[%d]:%s" % (i, self.print_state(V, i)) for i in range(count)]
        if self.G.rank == 0 and out is not None:
            print("\n".join(lines), file=out)
        return lines

    # -- free(), vp:335-405
    def free(self):
        if self._h:
            check(lib().gt_program_free(self._h))
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _inf_str(v):
    return "INF" if v == INF else str(int(v))


class Deg_Program(Vertex_Program):     # src/apps/deg.h:27-53
    kind = GT_DEG
    fields = (("degree", _lib.GT_F_DEGREE, np.uint32),)
    state_field = "degree"

    def print_state(self, V, i):       # deg.h:24
        return "Degree=%d" % V["degree"][i]


class PR_Program(Vertex_Program):      # src/apps/pr.h:21-48
    kind = GT_PR
    fields = (("degree", _lib.GT_F_DEGREE, np.uint32), ("rank", _lib.GT_F_RANK, np.float64))
    state_field = "rank"

    def print_state(self, V, i):       # pr.h:18
        return "Rank=%f,Degree=%d" % (V["rank"][i], V["degree"][i])


class BFS_Program(Vertex_Program):     # src/apps/bfs.h:33-82
    kind = GT_BFS
    fields = (("parent", _lib.GT_F_PARENT, np.uint32), ("hops", _lib.GT_F_HOPS, np.uint32))
    state_field = "hops"
    infinity = INF

    def print_state(self, V, i):       # bfs.h:29-30
        return "Parent=%d,Hops=%s" % (V["parent"][i], _inf_str(V["hops"][i]))


class SSSP_Program(Vertex_Program):    # src/apps/sssp.h:29-71
    kind = GT_SSSP
    fields = (("distance", _lib.GT_F_DISTANCE, np.uint32),)
    state_field = "distance"
    infinity = INF

    def print_state(self, V, i):       # sssp.h:26
        return "Distance=%s" % _inf_str(V["distance"][i])


class CC_Program(Vertex_Program):      # src/apps/cc.h:29-60
    kind = GT_CC
    fields = (("label", _lib.GT_F_LABEL, np.uint32),)
    state_field = "label"
    infinity = INF

    def print_state(self, V, i):       # cc.h:26
        return "Label=%d" % V["label"][i]
