#!/bin/bash
# rocprofv3 kernel statistics of short bench runs under a few environment settings (outputs under gpurun_out/prof_<tag>/)
#   gpurun -- 'bash tools/prof_kernels.sh tag1 "ENV=.. ENV2=.." tag2 "..."'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
while [ $# -ge 2 ]; do
  tag=$1; envs=$2; shift 2
  O=gpurun_out/prof_$tag; rm -rf $O; mkdir -p $O
  ( export $envs; rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 1 ${BENCH_ARGS} > $O/bench.json 2> $O/bench.err ) || exit 1
  f=$(ls $O/s/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $tag ($envs)"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-60s calls %5s avg %10.1f us total %6.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  python3 -c "import json,sys; d=json.load(open('$O/bench.json')); print('GTEPS %.1f kernel_ms %.3f f64 %s' % (d['value'], d['roofline']['kernel_ms'], d.get('f64_messages',{}).get('kernel_ms')))"
done
