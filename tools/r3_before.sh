#!/bin/bash
# dev helper (GPU box): the state at the start of round 3 -- profiles of base (bench, kernel stats, PMC passes) and k_top's phases
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_before; mkdir -p $O
cd $R && bash tools/profile_all.sh r03a_base > $O/profile_all.log 2>&1; tail -4 $O/profile_all.log
for cfg in "--config 2" "--config 4" "--config 2 --batch 1"; do
  GNNB_LIB=$R/tools/ablate/toptime.so timeout -k 10 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only $cfg 2>&1 | grep "k_top phases" | tail -4
  echo "== $cfg"
done > $O/toptime.log 2>&1
cat $O/toptime.log
