#!/usr/bin/env python3
"""Registers, LDS and the workgroups per CU that follow from them, per kernel of a .hip source (no GPU needed):
  python tools/kernel_regs.py [graphtap_amd/csrc/pb.hip] [--grep k_pb_]
hipcc -S for gfx950, the amdhsa metadata of the assembly. A 1 024-thread workgroup is 4 waves per SIMD: two of them fit on a CU only
with <= 64 VGPRs (512 per SIMD lane / 8 waves) AND <= 80 KiB of LDS (160 KiB per CU). Round 4 lost an afternoon to a kernel body
inlined twice (39 -> 85 VGPRs): tests/test_kernel_resources.py keeps the two-per-CU kernels where they are."""
import argparse, os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def kernel_resources(src, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-S", "--cuda-device-only", *extra, "-o", out, src], check=True)
        txt = open(out).read()
    md = txt[txt.index("amdhsa.kernels"):]
    res = []
    for ent in md.split("  - .agpr_count")[1:]:
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", ent).group(1))
        name = re.search(r"\.name:\s+(\S+)", ent).group(1)
        res.append(dict(name=name, vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"), spill=g("vgpr_spill_count"), wg=g("max_flat_workgroup_size")))
    return res


def workgroups_per_cu(k):
    waves = (k["wg"] + 63) // 64                     # waves of one workgroup
    per_simd = max(1, (waves + 3) // 4)
    by_vgpr = (512 // max(8, (k["vgpr"] + 7) // 8 * 8)) // per_simd
    by_lds = (160 * 1024) // k["lds"] if k["lds"] else 99
    by_waves = 8 // per_simd                         # 8 waves per SIMD on gfx950 (the guide)
    return max(0, min(by_vgpr, by_lds, by_waves))


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("src", nargs="?", default=os.path.join(ROOT, "graphtap_amd", "csrc", "pb.hip")); ap.add_argument("--grep", default="")
    a, extra = ap.parse_known_args()
    for k in kernel_resources(a.src, extra):
        if a.grep and a.grep not in k["name"]: continue
        dem = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(anonymous namespace\)::", "", dem); short = re.sub(r">\(.*", ">", short)[:96]
        print("%-98s vgpr %3d sgpr %3d lds %6d spill %d -> %d workgroup(s) of %d per CU" % (short, k["vgpr"], k["sgpr"], k["lds"], k["spill"], workgroups_per_cu(k), k["wg"]))
