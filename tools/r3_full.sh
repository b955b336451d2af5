#!/bin/bash
# dev helper (GPU box): the whole -m gpu suite, then step times of the bench configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_full; mkdir -p $O
cd $R && timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
for cfg in "--config 2" "--config 3" "--config 4" "--config 2 --batch 1" "--config 4 --batch 1024"; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$cfg', d['ms_per_step'], round(d['value']/1e6,2), 'launches', sum(v['launches'] for v in k.values())//20, {n: v['avg_us'] for n, v in k.items()})"
done 2>&1 | tee $O/bench.log
