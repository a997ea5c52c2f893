#!/bin/bash
# Collects everything under profiles/r02_hubs_first/ on a GPU box (outputs under gpurun_out/final/; copy into profiles/ afterwards):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2>$O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --no-cpu-baseline --no-f64 > $O/bench_line_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch/x -o p -- python3 bench.py --no-cpu-baseline --no-f64 --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write/x -o p -- python3 bench.py --no-cpu-baseline --no-f64 --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq/x -o p -- python3 bench.py --no-cpu-baseline --no-f64 --steps 3 --warmup 1 > /dev/null 2>&1
GRAPHTAP_PB_STATS=1 python bench.py --no-cpu-baseline --no-f64 --steps 2 --warmup 1 2>&1 >/dev/null | grep -E "^\[pb\]|^\[build\]" > $O/pb_build_stats_rmat26.txt || true
python bench.py --no-cpu-baseline --scale 22 > $O/bench_scale22.json 2>/dev/null
for app in bfs cc sssp; do python tools/bench_apps.py --scale 26 --apps $app >> $O/apps_scale26.jsonl 2>/dev/null; done   # one process per app
python tools/bench_apps.py --scale 24 --apps sssp >> $O/baseline_configs_sssp24_cc_standin.jsonl 2>/dev/null
python tools/bench_apps.py --scale 25 --edge-factor 36 --apps cc >> $O/baseline_configs_sssp24_cc_standin.jsonl 2>/dev/null
for r in 0 3 7; do python tools/bench_tilerow.py --scale 26 --nranks 8 --rank $r >> $O/tilerow_of_8_compute_only.jsonl 2>/dev/null; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tilerow_stats -o t -- python3 tools/bench_tilerow.py --scale 26 --nranks 8 --rank 3 > /dev/null 2>&1
[ -x tools/hbm_ceiling ] && ./tools/hbm_ceiling > $O/hbm_ceiling.txt 2>&1 || true
[ -x tools/lds_atomic_bench ] && ./tools/lds_atomic_bench > $O/lds_atomic_bench.txt 2>&1 || true
python3 profiles/collect_pmc.py $O/pmc_fetch $O/pmc_write scale26_gpus1 $O/pmc_traffic.json > /dev/null
echo collected
