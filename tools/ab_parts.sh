#!/bin/bash
# A/B of the pipelined multi-rank PageRank loop on ONE GPU: RCCL at world size 1 with the exchange layout forced on
# (GRAPHTAP_FORCE_EXCHANGE=1), R-MAT-26, K = 4 slices. GRAPHTAP_P2_PARTS: 0 = round 3's loop (exchange after the whole
# applicator), unset = phase 2 in K concurrent parts, slice k sent behind part k, 2 = the parts one after the other.
# usage (on a GPU box): bash tools/ab_parts.sh OUTFILE [rounds] [scale]
out=${1:-gpurun_out/ab_parts.txt}; rounds=${2:-2}; scale=${3:-26}
for r in $(seq 1 $rounds); do
  for m in 0 1 2; do
    GRAPHTAP_P2_PARTS=$m GRAPHTAP_FORCE_EXCHANGE=1 python bench.py --no-cpu-baseline --no-f64 --scale $scale 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
pr = r['per_rank'][0]
print('P2_PARTS=$m  GTEPS %.1f  ms/step %.4f  spmv span (events) %.4f  checksum %s  ' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['value_checksum']), {k: v for k, v in pr.items() if k.endswith('_ms')})
" >> $out
  done
done
cat $out
