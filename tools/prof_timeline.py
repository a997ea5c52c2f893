#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run (csv or rocpd .db): the launches between two marks, with start and duration.
  python tools/prof_timeline.py DIR [--grep k_bu_collect] [--last N]   prints the window around the last N matches"""
import argparse, csv, glob, os, re, sqlite3, sys
ap = argparse.ArgumentParser(); ap.add_argument("dir"); ap.add_argument("--grep", default=None); ap.add_argument("--last", type=int, default=1)
ap.add_argument("--before", type=int, default=4); ap.add_argument("--after", type=int, default=12)
a = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)): rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for f in glob.glob(os.path.join(a.dir, "**", "*.db"), recursive=True):
    rows += list(sqlite3.connect(f).execute("select name, start, end from kernels"))
rows.sort(key=lambda r: r[1])
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", ""))[:44]
idx = [i for i, r in enumerate(rows) if a.grep and a.grep in r[0]]
if not idx: idx = [len(rows) - 1]
for i in idx[-a.last:]:
    s, e = max(0, i - a.before), min(len(rows), i + a.after)
    t0 = rows[s][1]
    print("--")
    for r in rows[s:e]: print("%-46s start %9.1f us  dur %8.1f us" % (short(r[0]), (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3))
