"""Edge-file reading (the on-disk formats of /root/reference/src/mat/graph.hpp:195-372): 8/12-byte
little-endian binary records, and ASCII edge lists with '#' / '%' comment lines."""
import numpy as np
import pytest

from conftest import load_case


def test_binary_and_text_files_parse_to_the_same_records(tmp_path):
    from graphtap_amd.graph import read_edge_file
    from graphtap_amd import GraphTapError
    c = load_case("tiny")
    b = tmp_path / "e.bin"; c["edges"].tofile(b)
    t = tmp_path / "e.txt"
    t.write_text("# comment\n% another\n" + "".join("%d %d\n" % (a, b_) for a, b_ in c["edges"]))
    assert (read_edge_file(str(b), False) == c["edges"]).all()
    assert (read_edge_file(str(t), False) == c["edges"]).all()
    bw = tmp_path / "w.bin"; c["wedges"].tofile(bw)
    tw = tmp_path / "w.txt"; tw.write_text("".join("%d %d %d\n" % tuple(r) for r in c["wedges"]))
    assert (read_edge_file(str(bw), True) == c["wedges"]).all()
    assert (read_edge_file(str(tw), True) == c["wedges"]).all()
    with pytest.raises(GraphTapError, match="read\\(\\) failure"):      # wrong column count, graph.hpp:250-257
        read_edge_file(str(tw), False)
    bad = tmp_path / "bad.bin"; bad.write_bytes(b"\x00\x01\x02\xff" * 3)   # not a whole number of records
    with pytest.raises(GraphTapError, match="read\\(\\) failure"):
        read_edge_file(str(bad), False)


TRICKY = "# header\n\n% more\n0 1\n1 2\n# comment inside the list\n2 3\r\n3 4\n\n7 8\n8 9\n"   # the list ends at the first empty line


def test_text_rule_is_the_references_parread_text(tmp_path):
    """Same rule as the C++ front end (include/graphtap_amd.hpp parse_text), which follows parread_text (graph.hpp:195-304):
    leading comment / empty lines skipped, single spaces, the list ends at the first empty line, ids checked."""
    from graphtap_amd.graph import read_edge_file
    from graphtap_amd import GraphTapError
    t = tmp_path / "t.txt"; t.write_text(TRICKY)
    assert read_edge_file(str(t), False).tolist() == [[0, 1], [1, 2], [2, 3], [3, 4]]
    for bad in ("0  1\n", "0 1 \n", "0 x\n", "0 4294967296\n", "0 -1\n"):
        b = tmp_path / "b.txt"; b.write_text("0 1\n" + bad)
        with pytest.raises(GraphTapError, match="read\\(\\) failure"):
            read_edge_file(str(b), False)
    ok = tmp_path / "ok.txt"; ok.write_text("0 4294967295\n")
    assert read_edge_file(str(ok), False).tolist() == [[0, 4294967295]]


@pytest.mark.gpu
def test_both_front_ends_read_a_text_list_the_same_way(tmp_path):
    """The tricky list above through graphtap_amd.Graph.load (Python) and through apps/bin/cc (C++ shim): same graph."""
    import os
    import subprocess
    import graphtap_amd as gt
    from conftest import ROOT
    gt._lib.require_gpu()
    t = tmp_path / "t.txt"; t.write_text(TRICKY)
    G = gt.Graph(); G.load(str(t), 16, 16, False, False, True, False, False, gt._2DT_, gt._TCSC_)
    P = gt.CC_Program(G, False, True, False, gt._ROW_); P.execute()
    cs = P.checksum(out=None); it = P.iteration
    assert P.V["label"][:10].tolist() == [0, 0, 0, 0, 0, 5, 6, 7, 8, 9]     # 7-8-9 were cut off with the empty line
    P.free(); G.free()
    r = subprocess.run([os.path.join(ROOT, "apps", "bin", "cc"), str(t), "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Read 4 edges" in r.stdout and "Iterations: %d" % it in r.stdout and "Value checksum: %d" % cs[0] in r.stdout
    b = tmp_path / "b.txt"; b.write_text("0 1\n0  2\n")
    r = subprocess.run([os.path.join(ROOT, "apps", "bin", "cc"), str(b), "16"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "read() failure" in r.stderr
    with pytest.raises(gt.GraphTapError, match="read\\(\\) failure"):
        gt.Graph().load(str(b), 16, 16, False, False, True, False, False, gt._2DT_, gt._TCSC_)


@pytest.mark.gpu
def test_graph_load_from_files(tmp_path):
    import graphtap_amd as gt
    gt._lib.require_gpu()
    c = load_case("rmat8")
    t = tmp_path / "e.txt"; t.write_text("".join("%d %d\n" % (a, b) for a, b in c["edges"]))
    b = tmp_path / "e.bin"; c["edges"].tofile(b)
    outs = []
    for path in (t, b):
        G = gt.Graph(); G.load(str(path), 256, 256, False, False, True, False, False, gt._2DT_, gt._TCSC_)
        P = gt.CC_Program(G, False, True, False, gt._ROW_); P.execute()
        outs.append(P.V["label"]); P.free(); G.free()
    assert (outs[0] == outs[1]).all() and (outs[0][:257] == c["np1_cc_a"]).all()
