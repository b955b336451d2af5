#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_tail; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_online.py -m gpu -x -q -k "tail or classify_pre or host_entry or masks_all or scores_and_decisions or online" -s > $O/test.log 2>&1; echo "rc=$?"; grep -v "^graph requires" $O/test.log | tail -12
for cfg in "GNNB_TAIL_MAX_B=0 GNNB_CLSPRE_MAX_B=0" "GNNB_TAIL_MAX_B=8 GNNB_CLSPRE_MAX_B=0" "GNNB_TAIL_MAX_B=8 GNNB_CLSPRE_MAX_B=8"; do for B in 1 2 8; do
  env $cfg timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only --batch $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$cfg B=$B', d['ms_per_step'], 'launches', sum(v['launches'] for v in k.values())//100, {n: v['avg_us'] for n, v in k.items() if n in ('k_score','k_gather','k_classify','k_pre')})"
done; done 2>&1 | tee $O/bench.log
