#!/usr/bin/env python3
"""Captures the -DTIMING record of the UNMODIFIED reference (oracle/_ref/pr, built by oracle/ref/Makefile with -DTIMING)
on its bundled sample: the six lines display() prints before the vertex states (src/vp/vertex_program.hpp:2134-2152).
The numbers are this container's timings; tests/test_apps_cpp.py pins the FORMAT of our mains' record against it.
  python tests/golden/make_timing_golden.py"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
env = dict(os.environ, PATH=os.path.join(REF, "fileshim") + ":" + os.environ["PATH"])
out = subprocess.run(["/opt/conda/bin/mpirun", "-np", "1", os.path.join(REF, "pr"), os.path.join(HERE, "rmat10_1024.bin"), "1024", "20"],
                     env=env, capture_output=True, text=True, check=True).stdout.splitlines()
i = max(k for k, l in enumerate(out) if l.startswith("Init           time:"))   # the PageRank program's record (the second display)
rec = out[i:i + 6]
assert rec[5].startswith("TIMING ") and len(rec[5].split()) == 12, rec
open(os.path.join(HERE, "timing_record_rmat10_pr20.txt"), "w").write("\n".join(rec) + "\n")
print("\n".join(rec))
