#!/bin/bash
# BFS / CC / SSSP of R-MAT-26 under several switches of the sparse-frontier path (one process per app: a process that frees one
# large graph and builds the next one can stall for seconds inside hipMalloc on this pool)
for frac in 1024 256 64 32 16 8; do echo "FRACTION=$frac"
  for app in bfs cc sssp; do GRAPHTAP_SPMSPV_FRACTION=$frac python tools/bench_apps.py --scale 26 --apps $app 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['app'], round(d['execute_s']*1e3,2), 'ms cold', round(d['execute_warm_s']*1e3,2), 'warm', d['iterations'], 'it', d['sparse_iterations'], 'sparse', d['list_iterations'], 'list', [r['stepped_ms'][1] for r in d['per_iteration']])
"; done; done
