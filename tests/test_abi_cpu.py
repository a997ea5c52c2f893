"""CPU-side checks of the drop-in boundary: the C-ABI library is built, loads, and exports every
symbol include/graphtap_amd.h declares; the product refuses to run without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "graphtap_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gt_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from graphtap_amd import _build, _lib
    _build.build()
    L = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "missing export " + name
    assert sorted(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ set(declared)
    assert _lib.lib().gt_abi_version() == 3


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offset of the last member of every struct of the header, as gcc sees them, against the ctypes mirror"""
    import subprocess
    from graphtap_amd import _lib
    pairs = [("gt_graph_flags", _lib.GraphFlags, "parallel_edges"), ("gt_graph_info", _lib.GraphInfo, "send_elems"),
             ("gt_tile_arrays", _lib.TileArrays, "L2G"), ("gt_exchange_plan", _lib.ExchangePlan, "recv_counts"),
             ("gt_program_params", _lib.ProgramParams, "tol"), ("gt_exec_stats", _lib.ExecStats, "reserved0"),
             ("gt_graph_options", _lib.GraphOptions, "reserved"), ("gt_program_options", _lib.ProgramOptions, "reserved")]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "graphtap_amd.h"\nint main(void){' +
                   "".join('printf("%%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (c, c, m) for c, _, m in pairs) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for (cname, cls, member), line in zip(pairs, lines):
        size, off = map(int, line.split())
        assert C.sizeof(cls) == size and getattr(cls, member).offset == off, cname
    assert C.sizeof(_lib.GraphInfo) == 96 and C.sizeof(_lib.ExecStats) == 104


def test_option_structs_start_unset():
    """gt_graph_options_init / gt_program_options_init (no GPU needed): every field at its "unset" value, size filled in."""
    from graphtap_amd import _lib
    g, p = _lib.GraphOptions(), _lib.ProgramOptions()
    assert g.size == C.sizeof(_lib.GraphOptions) and (g.spmv_variant, g.force_exchange, g.hubs_first, g.wide_windows) == (-1, -1, -1, -1)
    assert (g.x_slices, g.hub_min_degree, g.exchange_hub_min, g.chunk_log2) == (0, 0, 0, 0)
    assert p.size == C.sizeof(_lib.ProgramOptions) and p.timeout_s == 0.0
    assert all(getattr(p, f) == -1 for f in ("frontier_lists", "spmspv", "tail_kernel", "bfs_bottom_up", "cc_first", "fuse_apply", "lean_state", "hybrid"))
    with pytest.raises(TypeError):
        _lib.GraphOptions(no_such_field=1)


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import graphtap_amd as gt
    assert gt._lib.lib().gt_device_count() == 0
    e = np.array([[0, 1], [1, 2]], np.uint32)
    with pytest.raises(gt.GraphTapError, match="no HIP device"):
        gt.Graph().load_edges(e, 4, 4, compression_type=gt._TCSC_, rank=0, nranks=1)
    with pytest.raises(gt.GraphTapError):
        gt._lib.require_gpu()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under graphtap_amd/, include/ or apps/ may reference it."""
    bad = []
    for sub in ("graphtap_amd", "include", "apps"):
        for d, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                    s = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"gt_oracle|from oracle|import oracle|oracle/", s):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_rmat_generator_is_deterministic_and_chunkable():
    from graphtap_amd.rmat import rmat_edges
    a = rmat_edges(10, 16, seed=1)
    b = np.concatenate([rmat_edges(10, 16, seed=1, first=0, count=5000), rmat_edges(10, 16, seed=1, first=5000)])
    assert a.shape == (16 << 10, 2) and (a == b).all() and a.max() < 1024
    w = rmat_edges(10, 16, seed=1, weighted=True)
    assert (w[:, :2] == a).all() and w[:, 2].min() >= 1 and w[:, 2].max() <= 128
    assert not (rmat_edges(10, 16, seed=2) == a).all()
    # R-MAT skew: quadrant (0,0) of the top level carries ~57 % of the edges
    top = ((a[:, 0] < 512) & (a[:, 1] < 512)).mean()
    assert 0.54 < top < 0.60
