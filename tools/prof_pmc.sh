#!/bin/bash
# rocprofv3 PMC pass of a short bench run: per-kernel means of the given counters (outputs under gpurun_out/pmc_<tag>/)
#   gpurun -- 'bash tools/prof_pmc.sh tag "SQ_INSTS_VALU SQ_INSTS_SALU ..." ["ENV=.."]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; ctrs=$2; envs=${3:-X=1}
O=gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O
( export $envs; rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/x -o p -- python3 bench.py --no-cpu-baseline --no-f64 --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err ) || exit 1
python3 - $O <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/x/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    m = re.search(r"\b(k_[A-Za-z0-9_]+)", r["Kernel_Name"])
    if not m or not m.group(1).startswith(("k_pb_", "k_pr_apply")): continue
    name = m.group(1) + ("<f64>" if "double, double" in r["Kernel_Name"] else "")
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
