#!/bin/bash
# tools/ab.sh OUTDIR NAME...: bench.py (no CPU baseline) under each variant library of graphtap_amd/lib/variants (NAME = base: the default build),
# back to back on one box, twice round-robin (boxes drift by a few percent)
out=$1; shift
mkdir -p $out
ROUNDS=${AB_ROUNDS:-2}
for round in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ "$v" == "base" ]; then lib=""; else lib=$PWD/graphtap_amd/lib/variants/$v.so; fi
    GRAPHTAP_PB_PHASE_TIMING=1 GRAPHTAP_LIB=$lib python bench.py --no-cpu-baseline ${AB_ARGS} > $out/$v.$round.json 2> $out/$v.$round.err || exit 1
    python - $out/$v.$round.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
f=d.get("f64_messages") or {}
print("%-10s %7.1f GTEPS  kernel %.3f ms  step %.3f ms  frac %.3f phases %s | f64: %s GTEPS kernel %s ms phases %s" % (sys.argv[2], d["value"], d["roofline"]["kernel_ms"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("phase_ms"), f.get("value"), f.get("kernel_ms"), f.get("phase_ms")), flush=True)
PY
  done
done
