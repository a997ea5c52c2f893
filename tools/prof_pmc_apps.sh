#!/bin/bash
# rocprofv3 PMC pass over tools/bench_apps.py: per-kernel means of the given counters for the propagation-blocking kernels
#   gpurun -- 'bash tools/prof_pmc_apps.sh tag "SQ_INSTS_VALU ..." "--scale 25 --apps cc" [GRAPHTAP_LIB=...]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; ctrs=$2; args=$3; envs=${4:-X=1}
O=gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O
( export $envs; rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/x -o p -- python3 tools/bench_apps.py $args > $O/out.jsonl 2> $O/err.txt ) || exit 1
python3 - $O <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/x/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"\b(k_pb_[A-Za-z0-9_]+)", r["Kernel_Name"])
    if not m: continue
    acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    # the heavy launches only (full passes): the top third by the first counter
    print(k, " ".join("%s=%.4g(max %.4g, n %d)" % (c, sum(v) / len(v), max(v), len(v)) for c, v in sorted(d.items())))
PY
