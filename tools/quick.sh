#!/bin/bash
# dev helper (GPU box): parity subset + kernel-trace stats of the three bench configurations -> gpurun_out/quick/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; mkdir -p $O
cd $R && timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q ${QUICK_K:+-k "$QUICK_K"} > $O/test.log 2>&1; tail -3 $O/test.log
for c in ${QUICK_CONFIGS:-2 3 4}; do
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$c -- python3 $R/bench.py --no-cpu-baseline --config $c > $O/bench$c.json 2> $O/stats$c.log || { echo "stats failed"; tail -5 $O/stats$c.log; exit 1; }
cp $(ls $O/stats$c/*/*kernel_stats.csv | head -1) $O/kernel_stats$c.csv; rm -rf $O/stats$c
head -c 300 $O/bench$c.json; echo; head -9 $O/kernel_stats$c.csv | cut -c1-150
done
