#!/bin/bash
# Collects what profiles/r04/ (round 3: profiles/r03/) holds, on a GPU box, in two parts of < 20 minutes (outputs under gpurun_out/final/; copy into profiles/ afterwards):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh A'
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh B'
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
if [ "$1" == "A" ]; then
  # first process of the lease: what a cold process pays where (round 2's first-process stall)
  GRAPHTAP_PB_STATS=1 python tools/cold_steps.py > $O/cold_steps_first_process.txt 2>&1
  python bench.py > $O/bench_default.json 2>$O/bench_default.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --no-cpu-baseline --no-f64 > $O/bench_line_under_rocprof.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch/x -o p -- python3 bench.py --no-cpu-baseline --no-f64 --steps 20 --warmup 0 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write/x -o p -- python3 bench.py --no-cpu-baseline --no-f64 --steps 20 --warmup 0 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq/x -o p -- python3 bench.py --no-cpu-baseline --no-f64 --steps 20 --warmup 0 > /dev/null 2>&1
  GRAPHTAP_COMMIT=${GRAPHTAP_COMMIT:-unknown} python3 profiles/collect_pmc.py $O/pmc_fetch $O/pmc_write scale26_gpus1 $O/pmc_traffic.json > /dev/null
  GRAPHTAP_PB_STATS=1 python bench.py --no-cpu-baseline --no-f64 --steps 2 --warmup 1 2>&1 >/dev/null | grep -E "^\[pb\]|^\[build\]" > $O/pb_build_stats_rmat26.txt || true
  GRAPHTAP_PB_PHASE_TIMING=1 python bench.py --no-cpu-baseline > $O/bench_with_phase_times.json 2>/dev/null
  python bench.py --no-cpu-baseline --scale 22 > $O/bench_scale22.json 2>/dev/null
  [ -x tools/hbm_ceiling2 ] && ./tools/hbm_ceiling2 > $O/hbm_ceiling2.txt 2>&1 || true
  echo collected A
else
  for app in bfs cc sssp; do python tools/bench_apps.py --scale 26 --apps $app >> $O/apps_scale26.jsonl 2>/dev/null; done   # one process per app
  python tools/bench_apps.py --scale 24 --apps sssp >> $O/baseline_configs_sssp24_cc_standin.jsonl 2>/dev/null
  python tools/bench_apps.py --scale 25 --edge-factor 36 --apps cc >> $O/baseline_configs_sssp24_cc_standin.jsonl 2>/dev/null
  for r in 0 3 7; do python tools/bench_tilerow.py --scale 26 --nranks 8 --rank $r >> $O/tilerow_of_8_compute_only.jsonl 2>/dev/null; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tilerow_stats -o t -- python3 tools/bench_tilerow.py --scale 26 --nranks 8 --rank 3 > /dev/null 2>&1
  python tools/bench_dist_iters.py --scale 22 --apps bfs,sssp,cc --transport rccl 2>/dev/null | grep "^{" > $O/dist_iterations_rmat22_rccl_world1.jsonl
  python tools/bench_dist_iters.py --scale 22 --nranks 8 --apps bfs --transport loopback 2>/dev/null | grep "^{" > $O/dist_iterations_bfs_rmat22_loopback8.jsonl
  GRAPHTAP_FORCE_EXCHANGE=1 python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_rmat26_forced_exchange_rccl_world1.json
  echo collected B
fi
