#!/bin/bash
# dev helper (GPU box): k_top after a change -- parity subset, phase stamps, step time at base B=256 / deep B=128 / base B=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_top; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "${TOPK:-top or scores_and_decisions or batched_equals or large_batch or bf16x3}" -s > $O/test.log 2>&1; tail -4 $O/test.log
[ -f $R/tools/ablate/toptime.so ] && for cfg in "--config 2" "--config 4" "--config 2 --batch 1"; do
  GNNB_LIB=$R/tools/ablate/toptime.so timeout -k 10 120 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only $cfg 2>&1 | grep "k_top phases" | tail -2
  echo "== $cfg"
done > $O/toptime.log 2>&1
cat $O/toptime.log
for cfg in "--config 2" "--config 4" "--config 2 --batch 1" "--config 4 --batch 1024"; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only $cfg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$cfg', d['ms_per_step'], round(d['value']/1e6,2), 'k_top', k.get('k_top',{}).get('avg_us'), 'instr', d['instrumented_ms_per_step'])"
done 2>&1 | tee $O/bench.log
