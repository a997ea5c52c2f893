#!/bin/bash
# tools/ab_env_apps.sh OUTDIR "ARGS of bench_apps.py" "ENV=.." ...: bench_apps.py under each environment setting ("X=1" = defaults), two rounds
out=$1; args=$2; shift 2
mkdir -p $out
for round in 1 2; do
  i=0
  for envs in "$@"; do
    i=$((i+1))
    ( export $envs; python tools/bench_apps.py $args > $out/e$i.$round.jsonl 2> $out/e$i.$round.err ) || exit 1
    python - $out/e$i.$round.jsonl "$envs" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print("%-28s %-5s scale %d: %.3f ms (warm %.3f), %d it; [msg, combine, apply] ms per iteration %s" % (sys.argv[2], d["app"], d["scale"], d["execute_s"]*1e3, d["execute_warm_s"]*1e3, d["iterations"], [p["stepped_ms"] for p in d["per_iteration"]][:6]), flush=True)
PY
  done
done
