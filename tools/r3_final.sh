#!/bin/bash
# dev helper (GPU box): everything profiles/ holds for round 3 -> gpurun_out/prof_r03f_{base,wide,deep}, gpurun_out/r3_final/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_final; mkdir -p $O
cd $R
bash tools/profile_all.sh r03f_base --config 2 > $O/base.log 2>&1; tail -2 $O/base.log | cut -c1-300
bash tools/profile_all.sh r03f_wide --config 3 > $O/wide.log 2>&1; tail -2 $O/wide.log | cut -c1-300
bash tools/profile_all.sh r03f_deep --config 4 > $O/deep.log 2>&1; tail -2 $O/deep.log | cut -c1-300
timeout -k 10 300 python3 bench.py --config 4 --batch 1024 --no-cpu-baseline > $O/deep_B1024_bench.json 2> $O/deep1024.err; head -c 300 $O/deep_B1024_bench.json; echo
SMALL_B="1 2 8" bash tools/small_b.sh > $O/small_b.log 2>&1; cat $O/small_b.log | cut -c1-160
cp gpurun_out/smallb/kernel_stats1.csv $O/base_B1_kernel_stats.csv
python3 tools/decision_latency.py > $O/decision_latency.log 2>&1; tail -6 $O/decision_latency.log
