#!/bin/bash
# tools/ab_round4_recheck2.sh: the library before the persistent phase 1 (variants/old.so) against the current one, and R-MAT-25 narrow / wide
O=gpurun_out/recheck2; mkdir -p $O
OLD=$PWD/graphtap_amd/lib/variants/old.so
echo "== PageRank R-MAT-26 with the f64 leg: old / now"
AB_ROUNDS=4 AB_ARGS=" " bash tools/ab_env.sh $O/pr26 "GRAPHTAP_LIB=$OLD" "X=1" 2>&1 | cut -c1-260
echo "== R-MAT-25 narrow / wide"
AB_ROUNDS=3 AB_ARGS="--no-f64 --scale 25" bash tools/ab_env.sh $O/wide25 "GRAPHTAP_PB_WIDE=0" "GRAPHTAP_PB_WIDE=1" 2>&1 | cut -c1-150
echo "== R-MAT-22 and 24, f32 messages: old / now"
for sc in 22 24; do AB_ROUNDS=3 AB_ARGS="--scale $sc" bash tools/ab_env.sh $O/pr$sc "GRAPHTAP_LIB=$OLD" "X=1" 2>&1 | cut -c1-260; done
echo "== the min programs: old / now"
bash tools/ab_env_apps.sh $O/apps26 "--scale 26" "GRAPHTAP_LIB=$OLD" "X=1" 2>&1 | cut -c1-110
bash tools/ab_env_apps.sh $O/apps24 "--scale 24" "GRAPHTAP_LIB=$OLD" "X=1" 2>&1 | cut -c1-110
