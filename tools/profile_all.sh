#!/bin/bash
# dev helper (GPU box): everything profiles/ holds for one configuration -> gpurun_out/prof_<tag>/
#   tools/profile_all.sh <tag> [bench.py args, e.g. --config 3 | --net cifar_deep_kw --batch 1024]
#   bench JSON (with cpu_baseline), rocprofv3 --kernel-trace --stats summary of the same command, PMC passes summary
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; shift
O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd $R && timeout -k 10 500 python3 bench.py "$@" > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 --no-aggregate-only --no-two-in-flight --no-host-fed "$@" > $O/stats.log 2>&1 || { echo "stats failed"; tail -5 $O/stats.log; exit 1; }
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cd $R && bash tools/pmc.sh $TAG "$@" > $O/pmc.log 2>&1
python3 tools/pmc_table.py $TAG $O/pmc_summary.json > $O/pmc_table.txt 2>&1
rm -rf $O/stats $R/gpurun_out/pmc_$TAG/p*/*/*agent_info.csv
head -c 700 $O/bench.json; echo; head -12 $O/kernel_stats.csv
