"""The C++ application mains (apps/*.cpp over include/graphtap_amd.hpp) print the reference's lines.

Expected lines are the reference's own output on its bundled samples (regenerated in
tests/golden/known_answers.json by running the unmodified reference)."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "apps", "bin")


def run(app, *args):
    exe = os.path.join(BIN, app)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "apps"), "all"])
    return subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=300)


def test_usage_line_without_arguments():
    """Same usage text and non-zero exit as the reference mains (e.g. apps/pr.cpp:18-22); needs no GPU."""
    r = run("pr")
    assert r.returncode != 0 and "<file_path> <num_vertices> [<num_iterations=INF>]" in r.stdout
    r = run("bfs")
    assert r.returncode != 0 and "<file_path> <num_vertices> <root>" in r.stdout


@pytest.mark.gpu
def test_pr_main_prints_the_reference_lines(known_answers):
    r = run("pr", os.path.join(GOLDEN, "rmat10_1024.bin"), 1024, 20)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    # Degree pass then PageRank pass, each with the three checksum() lines (vp:1944-1958)
    assert "Iterations: 1" in out and "Value checksum: 16384" in out and "Reachable vertices: 571" in out
    assert "Iterations: 20" in out and "Value checksum: 70" in out and "Reachable vertices: 1025" in out
    assert "vertex[0]:Rank=0.165455,Degree=10" in out and "vertex[1]:Rank=0.426287,Degree=2" in out
    assert "vertex[3]:Rank=0.151325,Degree=0" in out and "vertex[4]:Rank=1.238176,Degree=2" in out
    assert "vertex[5]:Rank=0.150000,Degree=0" in out
    assert sum(l.startswith("vertex[") for l in out) == 31
    assert any(l.startswith("Execute time:") for l in out) and any(l.startswith("Ingress time:") for l in out)


@pytest.mark.gpu
def test_bfs_cc_sssp_deg_mains(known_answers):
    k = known_answers["rmat10"]
    r = run("bfs", os.path.join(GOLDEN, "rmat10_1024.bin"), 1024, 0)
    assert r.returncode == 0, r.stderr
    assert "Iterations: %d" % k["np1_bfs"]["iterations"] in r.stdout and "Value checksum: 1912" in r.stdout
    assert "Reachable vertices: 887" in r.stdout and "vertex[1]:Parent=317,Hops=2" in r.stdout and "vertex[5]:Parent=866,Hops=3" in r.stdout
    r = run("cc", os.path.join(GOLDEN, "rmat10_1024.bin"), 1024)
    assert r.returncode == 0, r.stderr
    assert "Iterations: 4" in r.stdout and "Value checksum: 69590" in r.stdout and "vertex[5]:Label=0" in r.stdout
    r = run("sssp", os.path.join(GOLDEN, "rmat10_1024_w.bin"), 1024, 0)
    assert r.returncode == 0, r.stderr
    assert "Iterations: 7" in r.stdout and "Value checksum: 53366" in r.stdout and "Reachable vertices: 471" in r.stdout
    assert "vertex[1]:Distance=43" in r.stdout and "vertex[2]:Distance=INF" in r.stdout and "vertex[4]:Distance=51" in r.stdout
    r = run("deg", os.path.join(GOLDEN, "rmat10_1024.bin"), 1024)
    assert r.returncode == 0, r.stderr
    assert "Value checksum: 16384" in r.stdout and "vertex[0]:Degree=10" in r.stdout and "vertex[5]:Degree=1" in r.stdout


@pytest.mark.gpu
def test_missing_file_is_an_error_not_a_crash():
    r = run("pr", "/nonexistent/file.bin", 1024, 20)
    assert r.returncode == 1 and "Unable to open input file" in r.stderr


@pytest.mark.gpu
def test_deg_checksum1_and_timing_record():
    """checksum1 statistics (vp:1963-2119) as printed by the reference's `deg` on its bundled sample, and the
    -DTIMING-style record."""
    env = dict(os.environ, GRAPHTAP_TIMING="1")
    exe = os.path.join(BIN, "deg")
    r = subprocess.run([exe, os.path.join(GOLDEN, "rmat10_1024.bin"), "1024"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    # lines printed by the unmodified reference (oracle/_ref/deg tests/golden/rmat10_1024.bin 1024)
    assert "Sum: mean +/- std_dev: 16384.000000: 15.984390 +/- 81.271468" in r.stdout
    assert "Mode & skew : 0 & 0.196679" in r.stdout
    assert "Max index : 613" in r.stdout and "Max value : 1983" in r.stdout
    assert "TIMING scatter_gather" in r.stdout
