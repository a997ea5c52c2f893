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
    assert "Combine        time (sum: avg +/- std_dev):" in r.stdout and "\nTIMING " in r.stdout   # format pinned by the test below


# ------------------------------------------------------------------------------- edge-list converter (host only)
CONVERTER_CASES = {  # as in tests/golden/make_converter_golden.py
    "txt_to_bin": ("converter_in.txt", 0, 0, 1, 0, None),
    "txt_to_wbin": ("converter_in.txt", 0, 0, 1, 1, None),
    "wtxt_to_wbin_off3": ("converter_in_w.txt", 0, 1, 1, 1, 3),
    "wtxt_to_txt": ("converter_in_w.txt", 0, 1, 0, 0, None),
    "bin_to_wtxt": ("converter_txt_to_bin.out", 1, 0, 0, 1, None),
    "wbin_to_bin": ("converter_txt_to_wbin.out", 1, 1, 1, 0, 1),
}


@pytest.mark.parametrize("case", sorted(CONVERTER_CASES))
def test_converter_matches_the_reference_converter(case, tmp_path):
    """apps/converter.cpp against the files and statistics lines the unmodified reference converter produced
    (tests/golden/converter_*; weights of unweighted inputs come from the C library's default-seeded rand())."""
    inp, ib, iw, ob, ow, off = CONVERTER_CASES[case]
    out = tmp_path / "out"
    args = [inp, ib, iw, out, ob, ow] + ([off] if off is not None else [])
    exe = os.path.join(BIN, "converter")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "apps"), "bin/converter"])
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=60, cwd=GOLDEN)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == open(os.path.join(GOLDEN, "converter_%s.out" % case), "rb").read()
    assert r.stdout == open(os.path.join(GOLDEN, "converter_%s.stdout" % case)).read()
    ref = os.path.join(ROOT, "oracle", "_ref", "converter")
    if os.path.exists(ref):   # live comparison where the reference build is present
        out2 = tmp_path / "out_ref"
        args[3] = out2
        r2 = subprocess.run([ref, *map(str, args)], capture_output=True, text=True, timeout=60, cwd=GOLDEN)
        assert r2.returncode == 0 and out2.read_bytes() == out.read_bytes() and r2.stdout == r.stdout


def test_converter_errors():
    exe = os.path.join(BIN, "converter")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "apps"), "bin/converter"])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.startswith("Usage: ")
    r = subprocess.run([exe, "/nonexistent", "0", "0", "/tmp/x", "1", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unable to open input file" in r.stderr
    r = subprocess.run([exe, os.path.join(GOLDEN, "converter_in.txt"), "0", "1", "/dev/null", "1", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "read() failure" in r.stderr     # two columns where three were announced


@pytest.mark.gpu
def test_mains_read_text_edge_lists(tmp_path):
    """ASCII edge lists (graph.hpp:195-304) through the C++ shim: the same printed results as for the binary file."""
    conv = os.path.join(BIN, "converter")
    txt, wtxt = tmp_path / "rmat10.txt", tmp_path / "rmat10_w.txt"
    subprocess.check_call([conv, os.path.join(GOLDEN, "rmat10_1024.bin"), "1", "0", str(txt), "0", "0"], stdout=subprocess.DEVNULL)
    subprocess.check_call([conv, os.path.join(GOLDEN, "rmat10_1024_w.bin"), "1", "1", str(wtxt), "0", "1"], stdout=subprocess.DEVNULL)
    txt.write_text("# R-MAT scale 10\n% edge factor 16\n" + txt.read_text())
    keep = lambda s: [l for l in s.splitlines() if l.startswith(("Iterations", "Value checksum", "Reachable", "vertex["))]
    for app, f_bin, f_txt, arg in (("cc", "rmat10_1024.bin", txt, None), ("sssp", "rmat10_1024_w.bin", wtxt, 0)):
        extra = [] if arg is None else [arg]
        a = run(app, os.path.join(GOLDEN, f_bin), 1024, *extra); b = run(app, f_txt, 1024, *extra)
        assert a.returncode == 0 and b.returncode == 0, b.stderr
        assert keep(a.stdout) == keep(b.stdout) and len(keep(a.stdout)) == 34
        assert "Read %d edges" % (16384) in b.stdout
    r = run("cc", wtxt, 1024)      # three columns where cc expects two
    assert r.returncode == 1 and "read() failure" in r.stderr


@pytest.mark.gpu
def test_timing_record_has_the_references_format():
    """GRAPHTAP_TIMING=1: display() prints the reference's -DTIMING record (vp:2134-2152) -- per-iteration mean +/- std_dev
    of the three phases, first init and execute times, and the one-line TIMING summary. The format is pinned against the
    record the unmodified reference printed for the same command (tests/golden/make_timing_golden.py); the numbers must be
    self-consistent (TIMING repeats the lines above it; sum = 20 x mean for 20 iterations)."""
    import re
    exe = os.path.join(BIN, "pr")
    r = subprocess.run([exe, os.path.join(GOLDEN, "rmat10_1024.bin"), "1024", "20"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GRAPHTAP_TIMING="1"))
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    i = max(k for k, l in enumerate(out) if l.startswith("Init           time:"))
    rec = out[i:i + 6]
    ref = open(os.path.join(GOLDEN, "timing_record_rmat10_pr20.txt")).read().splitlines()
    shape = lambda lines: [re.sub(r"-?[0-9]+\.[0-9]{6}", "#", l) for l in lines]
    assert shape(rec) == shape(ref), (rec, ref)
    assert out[i + 6].startswith("vertex[0]:")
    nums = [float(x) for x in rec[5].split()[1:]]
    assert len(nums) == 11
    flat = [float(x) for l in rec[:5] for x in re.findall(r"-?[0-9]+\.[0-9]{6}", l)]
    assert flat == nums
    for k in (1, 4, 7):   # sum, mean, std of scatter_gather / combine / apply over 20 iterations
        assert abs(nums[k] - 20 * nums[k + 1]) <= 1e-4 * max(1.0, nums[k]) + 2e-5 and nums[k + 2] >= 0
