#!/usr/bin/env python3
"""Golden vectors for apps/converter.cpp, made by the UNMODIFIED reference converter (oracle/_ref/converter, built by
`make -C oracle/ref` from /root/reference/src/misc/converter.cpp). Run in the build container:
    python tests/golden/make_converter_golden.py
Writes converter_in.txt (our input: comment lines, 48 edges), converter_in_w.txt and, per mode, the reference's output
file and stdout: converter_<case>.out / .stdout."""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "..", "oracle", "_ref", "converter")
CASES = {  # name: (input, in_bin, in_w, out_bin, out_w, offset)
    "txt_to_bin": ("converter_in.txt", 0, 0, 1, 0, None),
    "txt_to_wbin": ("converter_in.txt", 0, 0, 1, 1, None),
    "wtxt_to_wbin_off3": ("converter_in_w.txt", 0, 1, 1, 1, 3),
    "wtxt_to_txt": ("converter_in_w.txt", 0, 1, 0, 0, None),
    "bin_to_wtxt": ("converter_txt_to_bin.out", 1, 0, 0, 1, None),
    "wbin_to_bin": ("converter_txt_to_wbin.out", 1, 1, 1, 0, 1),
}


def main():
    rng = np.random.RandomState(7)
    e = rng.randint(0, 200, (48, 2))
    w = rng.randint(1, 129, 48)
    with open(os.path.join(HERE, "converter_in.txt"), "w") as f:
        f.write("# directed edge list\n% 48 edges\n" + "".join("%d %d\n" % (a, b) for a, b in e))
    with open(os.path.join(HERE, "converter_in_w.txt"), "w") as f:
        f.write("".join("%d %d %d\n" % (a, b, c) for (a, b), c in zip(e, w)))
    for name, (inp, ib, iw, ob, ow, off) in CASES.items():
        out = os.path.join(HERE, "converter_%s.out" % name)
        cmd = [REF, inp, str(ib), str(iw), os.path.basename(out), str(ob), str(ow)] + ([str(off)] if off is not None else [])
        r = subprocess.run(cmd, cwd=HERE, check=True, capture_output=True, text=True)
        open(os.path.join(HERE, "converter_%s.stdout" % name), "w").write(r.stdout)
        print(name, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
