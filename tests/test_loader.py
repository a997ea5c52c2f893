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
