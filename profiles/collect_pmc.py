#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of bench.py into profiles/pmc_traffic.json, the file bench.py reads
for `roofline.traffic`.

On the GPU box (counters in their own runs, FETCH_SIZE and WRITE_SIZE in separate passes because they do
not fit the TCC slots together; MI355X_MICROARCH.md, "rocprofv3 PMC slots"):

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1
  python3 profiles/collect_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write scale26_gpus1

Units and gfx950 correction (same guide, "HBM"): the counters are in KiB; FETCH_SIZE reports exactly half of
the bytes of a coalesced streaming read on gfx950, so it is doubled. The doubling was checked against the
known byte counts of these kernels (k_pb_scatter reads LCOL + group records + windows, k_pb_gather reads
VAL + LROW): 2 x FETCH matches them within 3 %. WRITE_SIZE is taken as is.
"""
import collections
import csv
import datetime
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    """sha256 (16 hex digits) of the SpMV kernels' source: bench.py reports `traffic` only while it still matches."""
    return hashlib.sha256(open(os.path.join(ROOT, "graphtap_amd", "csrc", "pb.hip"), "rb").read()).hexdigest()[:16]


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            m = re.search(r"\b(k_[A-Za-z0-9_]+)", r["Kernel_Name"])
            acc[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}   # mean bytes per launch


def main():
    fetch_dir, write_dir, key = sys.argv[1:4]
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    spmv = [k for k in fetch if k.startswith("k_pb_") or k.startswith("k_spmv_edge")]
    rec = {"kernels": {k: {"hbm_read_bytes": 2 * fetch[k], "hbm_write_bytes": write.get(k, 0.0)} for k in sorted(fetch)
                       if k.startswith(("k_pb_", "k_spmv_edge", "k_pr_", "k_msg_", "k_apply_"))},
           "hbm_bytes_per_launch": sum(2 * fetch[k] + write.get(k, 0.0) for k in spmv),
           "note": "one SpMV launch = k_pb_scatter + k_pb_gather; FETCH_SIZE x2 (gfx950), WRITE_SIZE x1, KiB -> bytes",
           "collected": {"date": datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M UTC"), "pb_hip_sha16": kernel_source_sha(),
                         "commit": os.environ.get("GRAPHTAP_COMMIT", "unknown (the GPU box has no .git; set GRAPHTAP_COMMIT)")}}
    out = out_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    allrec = json.load(open(out)) if os.path.exists(out) else {}
    allrec[key] = rec
    json.dump(allrec, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
