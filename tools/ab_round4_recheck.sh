#!/bin/bash
# tools/ab_round4_recheck.sh: everything the persistent phase 1 was compared on, again, after the register fix (one call site of the kernel body)
O=gpurun_out/recheck; mkdir -p $O
export GRAPHTAP_TIMEOUT_S=120
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q > $O/tests.txt 2>&1; tail -1 $O/tests.txt
echo "== PageRank R-MAT-26, GRAPHTAP_PB_PERSIST 0 / default / 3, with the f64 leg"
AB_ROUNDS=4 AB_ARGS=" " bash tools/ab_env.sh $O/pr26 "GRAPHTAP_PB_PERSIST=0" "X=1" "GRAPHTAP_PB_PERSIST=3" 2>&1 | cut -c1-260
echo "== through the exchange at world size 1: the library before this work / now / now with the persistent phase 1 forced"
( export GRAPHTAP_FORCE_EXCHANGE=1; AB_ROUNDS=2 AB_ARGS="--no-f64" bash tools/ab_env.sh $O/fx "GRAPHTAP_LIB=$PWD/graphtap_amd/lib/variants/old.so" "X=1" "GRAPHTAP_PB_PERSIST=1" 2>&1 | cut -c1-170 )
for sc in 20 21 22 24; do echo "== R-MAT-$sc narrow / wide"; AB_ROUNDS=3 AB_ARGS="--no-f64 --scale $sc" bash tools/ab_env.sh $O/wide$sc "GRAPHTAP_PB_WIDE=0" "GRAPHTAP_PB_WIDE=1" 2>&1 | cut -c1-150; done
echo "== the min programs, dispatched / persistent phase 1"
bash tools/ab_env_apps.sh $O/apps26 "--scale 26" "GRAPHTAP_PB_PERSIST=0" "GRAPHTAP_PB_PERSIST=1" 2>&1 | cut -c1-110
bash tools/ab_env_apps.sh $O/apps24 "--scale 24" "GRAPHTAP_PB_PERSIST=0" "GRAPHTAP_PB_PERSIST=1" 2>&1 | cut -c1-110
