#!/bin/bash
# dev helper (GPU box): same-box A/B of the shipped library against an ablation build:  tools/r3_abl.sh <ablate/x.so> [bench args]
R=$GRAFT_REPO_ROOT; LIBX=$1; shift
for lib in "" "$R/$LIBX"; do
  GNNB_LIB=$lib; [ -z "$lib" ] && unset GNNB_LIB || export GNNB_LIB
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('${lib:-shipped}', d['ms_per_step'], {n: v['avg_us'] for n, v in k.items()})"
done
