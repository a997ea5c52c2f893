#!/bin/bash
# tools/ab_apps.sh OUTDIR "ARGS of bench_apps.py" NAME...: bench_apps.py under each variant library (NAME = base: the default build)
out=$1; args=$2; shift 2
mkdir -p $out
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" == "base" ]; then lib=""; else lib=$PWD/graphtap_amd/lib/variants/$v.so; fi
    GRAPHTAP_LIB=$lib python tools/bench_apps.py $args > $out/$v.$round.jsonl 2> $out/$v.$round.err || exit 1
    python - $out/$v.$round.jsonl $v <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print("%-10s %-5s scale %d: %.3f ms (warm %.3f), %d iterations; combine ms per iteration %s" % (sys.argv[2], d["app"], d["scale"], d["execute_s"]*1e3, d["execute_warm_s"]*1e3, d["iterations"], [p["stepped_ms"][1] for p in d["per_iteration"]]), flush=True)
PY
  done
done
