#!/bin/bash
# dev helper (GPU box): the stand-alone edge-aggregation leg (GNNB_FUSE=0 GNNB_NO_EMBED_FUSE=1 GNNB_TAIL_MAX_B=0) of one configuration ->
# gpurun_out/prof_<tag>/: bench JSON of that path, rocprofv3 kernel stats, FETCH / WRITE counter passes per kernel template
#   tools/profile_aggonly.sh <tag> [bench.py args, e.g. --config 3]
R=$GRAFT_REPO_ROOT; TAG=${1:-r04_aggonly}; shift
O=$R/gpurun_out/prof_$TAG; mkdir -p $O
export GNNB_FUSE=0 GNNB_NO_EMBED_FUSE=1 GNNB_TAIL_MAX_B=0
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 --no-aggregate-only "$@" > $O/bench.json 2> $O/stats.log || { echo "stats failed"; tail -5 $O/stats.log; exit 1; }
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
cd $R && PMC_PASSES="3 4" bash tools/pmc.sh $TAG "$@" > $O/pmc.log 2>&1
python3 tools/pmc_table.py $TAG $O/pmc_summary.json > $O/pmc_table.txt 2>&1
rm -rf $O/stats $R/gpurun_out/pmc_$TAG/p*/*/*agent_info.csv
head -c 300 $O/bench.json; echo; cut -d, -f1,2,4 $O/kernel_stats.csv | head -12
