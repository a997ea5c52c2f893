#!/bin/bash
# A/B of the WIDE propagation-blocking build for the min programs (GRAPHTAP_PB_WIDE=0 / 1), round-robin, on one box.
# usage: bash tools/ab_wide_apps.sh OUT [rounds]
out=${1:-gpurun_out/ab_wide_apps.txt}; rounds=${2:-2}
for r in $(seq 1 $rounds); do
 for cfg in "24 sssp 16" "26 sssp 16" "26 cc 16" "25 cc 36" "26 bfs 16"; do set -- $cfg
  for h in 0 1; do
   GRAPHTAP_PB_WIDE=$h python tools/bench_apps.py --scale $1 --apps $2 --edge-factor $3 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PB_WIDE=$h %s scale %d ef %d: %d iterations, execute %.3f ms (warm %.3f), checksum %s' % (r['app'], r['scale'], r['edge_factor'], r['iterations'], r['execute_s'] * 1e3, r['execute_warm_s'] * 1e3, r['value_checksum']))" >> $out
  done
 done
done
cat $out
