"""ctypes binding of include/graphtap_amd.h. There is NO CPU fallback: a missing library or a
missing GPU raises, it never degrades to another implementation."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRAPHTAP_LIB") or os.path.join(HERE, "lib", "libgraphtap_amd.so")   # GRAPHTAP_LIB: an alternative build (A/B experiments)

GT_INF = 2147483647
GT_DEG, GT_PR, GT_BFS, GT_SSSP, GT_CC = range(5)
GT_ROW, GT_COL = 0, 1
GT_TCSC, GT_TCSC_CF = 0, 1
GT_PLUS_F64, GT_PLUS_U32, GT_MIN_U32, GT_MINPLUS_U32 = range(4)
GT_SPMV_EDGE, GT_SPMV_PB, GT_SPMV_PB_F32MSG = 0, 1, 2
GT_F_DEGREE, GT_F_RANK, GT_F_PARENT, GT_F_HOPS, GT_F_DISTANCE, GT_F_LABEL, GT_F_ACTIVE = range(7)


class GraphTapError(RuntimeError):
    pass


class GraphFlags(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("directed", "transpose", "self_loops", "acyclic", "parallel_edges")]


class GraphInfo(C.Structure):
    _fields_ = [("num_vertices", C.c_uint32), ("nrows", C.c_uint32), ("tile_height", C.c_uint32),
                ("rank", C.c_uint32), ("nranks", C.c_uint32), ("nnzrows", C.c_uint32), ("nnzcols", C.c_uint32),
                ("seg_stride", C.c_uint32), ("nnz_local", C.c_uint64), ("nnz_global", C.c_uint64),
                ("nnzrows_global", C.c_uint64), ("nnzcols_global", C.c_uint64), ("weighted", C.c_int32),
                ("regular", C.c_uint32), ("source_rows", C.c_uint32), ("sink_cols", C.c_uint32),
                ("x_slices", C.c_uint32), ("slice_width", C.c_uint32), ("ncols_local", C.c_uint32), ("send_elems", C.c_uint32)]


class TileArrays(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("JA", "IA", "A", "JC", "IR", "L2G")]


class TileCFArrays(C.Structure):
    _fields_ = [("IA", C.c_void_p), ("A", C.c_void_p), ("JA_REG_R_NNZ_C", C.c_void_p), ("NC", C.c_uint32 * 4),
                ("JA", C.c_void_p * 4), ("JC", C.c_void_p * 4)]


class ExchangePlan(C.Structure):
    _fields_ = [("nranks", C.c_uint32), ("x_slices", C.c_uint32),
                ("send_offset", C.POINTER(C.c_uint64)), ("recv_offset", C.POINTER(C.c_uint64)),
                ("send_counts", C.POINTER(C.c_uint32)), ("recv_counts", C.POINTER(C.c_uint32))]


class ProgramParams(C.Structure):
    _fields_ = [("kind", C.c_int32), ("order", C.c_int32), ("compression", C.c_int32), ("root", C.c_uint32),
                ("alpha", C.c_double), ("tol", C.c_double)]


class ExecStats(C.Structure):
    _fields_ = [("iterations", C.c_uint32), ("converged", C.c_uint32), ("seconds", C.c_double),
                ("spmv_ms", C.c_double), ("spmv_launches", C.c_uint32), ("fused_apply_rows", C.c_uint32),
                ("scatter_gather_ms", C.c_double), ("combine_ms", C.c_double), ("apply_ms", C.c_double),
                ("spmspv_iterations", C.c_uint32), ("phase_samples", C.c_uint32),
                ("scatter_gather_sq", C.c_double), ("combine_sq", C.c_double), ("apply_sq", C.c_double),
                ("cf_filtered_iterations", C.c_uint32), ("list_iterations", C.c_uint32),
                ("allocs_in_execute", C.c_uint32), ("reserved0", C.c_uint32)]


class GraphOptions(C.Structure):
    """gt_graph_options: per-handle configuration of a graph build (unset fields fall back to the environment)."""
    _fields_ = [("size", C.c_uint32), ("spmv_variant", C.c_int32), ("force_exchange", C.c_int32), ("x_slices", C.c_uint32),
                ("hubs_first", C.c_int32), ("hub_min_degree", C.c_uint32), ("exchange_hub_min", C.c_uint32), ("chunk_log2", C.c_uint32),
                ("wide_windows", C.c_int32), ("reserved", C.c_uint32 * 7)]

    def __init__(self, **kw):
        super().__init__()
        lib().gt_graph_options_init(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("gt_graph_options has no field " + k)
            setattr(self, k, v)


class ProgramOptions(C.Structure):
    """gt_program_options: per-handle configuration of a program (unset fields fall back to the graph's, then the environment)."""
    _fields_ = [("size", C.c_uint32), ("frontier_lists", C.c_int32), ("spmspv", C.c_int32), ("tail_kernel", C.c_int32),
                ("bfs_bottom_up", C.c_int32), ("cc_first", C.c_int32), ("fuse_apply", C.c_int32), ("lean_state", C.c_int32), ("hybrid", C.c_int32),
                ("spmspv_fraction", C.c_uint32), ("tail_list_max", C.c_uint32), ("tail_entries_max", C.c_uint32), ("reserved0", C.c_uint32),
                ("timeout_s", C.c_double), ("reserved", C.c_uint32 * 8)]

    def __init__(self, **kw):
        super().__init__()
        lib().gt_program_options_init(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("gt_program_options has no field " + k)
            setattr(self, k, v)


# every symbol include/graphtap_amd.h declares: (restype, argtypes)
_vp = C.c_void_p
SIGNATURES = {
    "gt_abi_version": (C.c_int, []),
    "gt_last_error": (C.c_char_p, []),
    "gt_device_count": (C.c_int, []),
    "gt_set_device": (C.c_int, [C.c_int]),
    "gt_graph_build": (C.c_int, [C.POINTER(_vp), _vp, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.POINTER(GraphFlags), C.c_int, C.c_int]),
    "gt_graph_build_distributed": (C.c_int, [C.POINTER(_vp), _vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.POINTER(GraphFlags)]),
    "gt_graph_options_init": (None, [C.POINTER(GraphOptions)]),
    "gt_program_options_init": (None, [C.POINTER(ProgramOptions)]),
    "gt_graph_build_opt": (C.c_int, [C.POINTER(_vp), _vp, _vp, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.POINTER(GraphFlags), C.c_int, C.c_int, C.POINTER(GraphOptions)]),
    "gt_program_set_options": (C.c_int, [_vp, C.POINTER(ProgramOptions)]),
    "gt_graph_info_get": (C.c_int, [_vp, C.POINTER(GraphInfo)]),
    "gt_graph_has_wide_build": (C.c_int, [_vp]),
    "gt_graph_select_spmv": (C.c_int, [_vp, C.c_int]),
    "gt_graph_vertex_ids": (C.c_int, [_vp, _vp, C.c_uint64]),
    "gt_graph_tile": (C.c_int, [_vp, C.POINTER(TileArrays)]),
    "gt_graph_exchange_plan": (C.c_int, [_vp, C.POINTER(ExchangePlan)]),
    "gt_graph_free": (C.c_int, [_vp]),
    "gt_program_create": (C.c_int, [C.POINTER(_vp), _vp, C.POINTER(ProgramParams)]),
    "gt_program_initialize": (C.c_int, [_vp]),
    "gt_program_initialize_from": (C.c_int, [_vp, _vp]),
    "gt_graph_phase_times": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_int]),
    "gt_program_prepare": (C.c_int, [_vp, C.c_uint32]),
    "gt_program_execute": (C.c_int, [_vp, C.c_uint32, C.POINTER(ExecStats)]),
    "gt_program_set_stream": (C.c_int, [_vp, _vp]),
    "gt_program_enable_timing": (C.c_int, [_vp, C.c_int]),
    "gt_program_timing": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_int]),
    "gt_program_x": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "gt_program_set_x": (C.c_int, [_vp, _vp]),
    "gt_program_send": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "gt_program_set_send": (C.c_int, [_vp, _vp]),
    "gt_program_scatter_gather": (C.c_int, [_vp]),
    "gt_program_combine": (C.c_int, [_vp]),
    "gt_program_combine_slice": (C.c_int, [_vp, C.c_uint32]),
    "gt_program_fuse_apply": (C.c_int, [_vp, C.c_uint32, C.c_int]),
    "gt_program_apply": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_uint64)]),
    "gt_program_finish_converged": (C.c_int, [_vp]),
    "gt_program_iteration": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "gt_program_y": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "gt_program_copy_state": (C.c_int, [_vp, C.c_int, _vp, C.c_uint64]),
    "gt_program_checksum": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gt_program_free": (C.c_int, [_vp]),
    "gt_dist_unique_id": (C.c_int, [_vp]),
    "gt_dist_create": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int]),
    "gt_dist_create_from_comm": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int]),
    "gt_dist_create_loopback": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "gt_dist_free": (C.c_int, [_vp]),
    "gt_dist_iteration_times": (C.c_int, [_vp, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_uint32)]),
    "gt_dist_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gt_dist_execute": (C.c_int, [_vp, _vp, C.c_uint32, C.POINTER(ExecStats)]),
    "gt_dist_all_reduce_u64": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_uint32]),
    "gt_dist_exchange_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    "gt_spmv": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "gt_spmv_cf": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "gt_graph_tile_cf": (C.c_int, [_vp, C.POINTER(TileCFArrays)]),
    "gt_rmat_generate": (C.c_int, [_vp, C.c_int, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, _vp]),
    "gt_malloc": (C.c_int, [C.POINTER(_vp), C.c_uint64]),
    "gt_free": (C.c_int, [_vp]),
    "gt_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_uint64]),
    "gt_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_uint64]),
    "gt_memset": (C.c_int, [_vp, C.c_int, C.c_uint64]),
    "gt_device_synchronize": (C.c_int, []),
    "gt_diag_hbm_ceiling": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_double)]),
}

_lib = None


def lib():
    """Loads the HIP library; raises GraphTapError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GraphTapError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(graphtap_amd has no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        if L.gt_abi_version() != 3:
            raise GraphTapError("ABI version mismatch")
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise GraphTapError("graphtap_amd error %d: %s" % (status, lib().gt_last_error().decode()))


def require_gpu():
    if lib().gt_device_count() <= 0:
        raise GraphTapError("no HIP device visible: graphtap_amd has no CPU fallback")
