#!/bin/bash
# dev helper (GPU box): same-box comparison of the shipped library with ablation builds:  tools/r3_abl2.sh "bench args" lib1 lib2 ...
R=$GRAFT_REPO_ROOT; ARGS=$1; shift
for lib in "" "$@"; do
  if [ -z "$lib" ]; then unset GNNB_LIB; else export GNNB_LIB=$R/tools/ablate/$lib.so; fi
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('${lib:-shipped}', d['ms_per_step'], {n: v['avg_us'] for n, v in k.items()})"
done
