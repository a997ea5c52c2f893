import os
import sys

import numpy as np
import pytest

# PyTorch bundles its own HIP runtime (same SONAME as /opt/rocm's). Whichever copy a process loads first serves both
# PyTorch and libgraphtap_amd.so; PyTorch's refuses to see the GPU behind the system copy ("No HIP GPUs are available"),
# the other order works. Tests that use both (the torch.distributed drivers) therefore need torch imported first, whatever
# subset of test files is collected.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_case(name):
    """A golden case: the edge list fed to the reference + the reference's outputs (tests/golden/make_golden.py)."""
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "edges" not in z:  # the reference's own bundled samples are kept as the original files
        z["edges"] = np.fromfile(os.path.join(GOLDEN, "rmat10_1024.bin"), dtype="<u4").reshape(-1, 2)
        z["wedges"] = np.fromfile(os.path.join(GOLDEN, "rmat10_1024_w.bin"), dtype="<u4").reshape(-1, 3)
    z["num_vertices"] = int(z["num_vertices"]); z["root"] = int(z["root"])
    return z


def load_cf_case(name):
    """(edges, num_vertices, the reference's TCSC_CF arrays) of a case of tests/golden/tcsc_cf.npz (make_tcsc_cf_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "tcsc_cf.npz"))
    if name == "mixed":
        edges, nv = z["mixed_edges"], int(z["mixed_num_vertices"])
    else:
        c = load_case(name); edges, nv = c["edges"], c["num_vertices"]
    pre = name + "_u_"
    return edges, nv, {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


CF_CASES = ["tiny", "rmat8", "rmat10", "rmat12", "mixed"]
CASES = ["tiny", "rmat8", "rmat10", "rmat12"]
FIRST_NP = {"tiny": 1, "rmat8": 1, "rmat10": 1, "rmat12": 1}
OTHER_NP = {"tiny": 2, "rmat8": 2, "rmat10": 4, "rmat12": 8}


@pytest.fixture(scope="session")
def known_answers():
    import json
    return json.load(open(os.path.join(GOLDEN, "known_answers.json")))
