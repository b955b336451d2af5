#!/bin/bash
# dev helper (GPU box): bench under env settings with a given library:  tools/r3_env.sh <lib.so|-> "bench args" "ENV1=.." "ENV2=.."
R=$GRAFT_REPO_ROOT; LIBX=$1; ARGS=$2; shift; shift
[ "$LIBX" != "-" ] && export GNNB_LIB=$R/$LIBX
for cfg in "X=0" "$@"; do
  env $cfg timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$cfg', d['ms_per_step'], {n: v['avg_us'] for n, v in k.items()}, [u['kernel'] for u in d['plan']])"
done
