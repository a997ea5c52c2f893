#!/bin/bash
# tools/ab_env.sh OUTDIR "ENV=.." ...: bench.py (no CPU baseline, no f64 leg) under each environment setting ("X=1" = defaults), AB_ROUNDS rounds
out=$1; shift
mkdir -p $out
for round in $(seq 1 ${AB_ROUNDS:-2}); do
  i=0
  for envs in "$@"; do
    i=$((i+1))
    ( export $envs; GRAPHTAP_PB_PHASE_TIMING=1 python bench.py --no-cpu-baseline ${AB_ARGS:---no-f64} > $out/e$i.$round.json 2> $out/e$i.$round.err ) || exit 1
    python - $out/e$i.$round.json "$envs" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
f=d.get("f64_messages") or {}
print("%-34s %7.1f GTEPS  kernel %.3f ms  phases %s | f64: %s GTEPS phases %s" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"], d["roofline"].get("phase_ms"), f.get("value"), f.get("phase_ms")), flush=True)
PY
  done
done
