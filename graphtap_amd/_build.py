"""Builds graphtap_amd/lib/libgraphtap_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per translation unit (compiled side by side, only the stale ones), then one link."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libgraphtap_amd.so")
OBJ = os.path.join(HERE, "lib", "obj")
SOURCES = ["engine.hip", "ingest.hip", "kernels.hip", "pb.hip", "dist.hip", "tcsc_cf.hip", "diag.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(HERE, "..", "include", "graphtap_amd.h")]


def _obj(src):
    return os.path.join(OBJ, src.replace(".hip", ".o"))


def _stale_sources(force):
    ht = max(os.path.getmtime(h) for h in _headers())
    out = []
    for s in SOURCES:
        o = _obj(s)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(ht, os.path.getmtime(os.path.join(CSRC, s))):
            out.append(s)
    return out


def build(force=False, verbose=False):
    stale = _stale_sources(force)
    if not stale and os.path.exists(LIB) and all(os.path.getmtime(_obj(s)) <= os.path.getmtime(LIB) for s in SOURCES):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def compile_one(s):
        cmd = [hipcc] + FLAGS + ["-c", "-o", _obj(s), os.path.join(CSRC, s)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(7, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, stale))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [_obj(s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True)
