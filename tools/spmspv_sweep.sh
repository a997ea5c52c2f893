#!/bin/bash
# BFS / CC / SSSP of R-MAT-26 under several switches of the sparse-frontier path (one process per app: a process that frees one
# large graph and builds the next one can stall for seconds inside hipMalloc on this pool)
for cfg in "4096 16384" "1024 16384" "1024 4000000" "256 4000000" "64 4000000" "16 40000000"; do set -- $cfg; echo "FRACTION=$1 MAX_ACTIVE=$2"
  for app in bfs cc sssp; do GRAPHTAP_SPMSPV_FRACTION=$1 GRAPHTAP_SPMSPV_MAX_ACTIVE=$2 python tools/bench_apps.py --scale 26 --apps $app 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['app'], round(d['execute_s']*1e3,2), 'ms', d['iterations'], 'it', d['sparse_iterations'], 'sparse', [r['stepped_ms'][1] for r in d['per_iteration']])
"; done; done
