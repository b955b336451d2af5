#!/bin/bash
# dev helper (GPU box): per-kernel durations of one forward at small batch sizes -> gpurun_out/smallb/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/smallb; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in ${SMALL_B:-1 8}; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$B -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 --no-aggregate-only --net ${SMALL_NET:-cifar_base_kw} --batch $B --steps 200 --warmup 20 > $O/bench$B.json 2> $O/s$B.log || { echo "failed"; tail -5 $O/s$B.log; exit 1; }
cp $(ls $O/s$B/*/*kernel_stats.csv | head -1) $O/kernel_stats$B.csv; rm -rf $O/s$B
python3 -c "
import json,sys
d=json.loads(open('$O/bench$B.json').read().strip().splitlines()[-1]); print('B=$B ms_per_step', d['ms_per_step'])"
cut -d, -f1,2,4 $O/kernel_stats$B.csv | head -16
done
