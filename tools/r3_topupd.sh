#!/bin/bash
# dev helper (GPU box): the update of layer L-1 inside k_top (default) against its own launch (GNNB_TOP_FUSE_UPD=0), same box
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only $@ 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('  ', d['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if n in ('k_top','k_node_update')})"; }
for cfg in "--config 2" "--config 3" "--config 4" "--config 4 --batch 1024" "--net cifar_base_kw --batch 1" "--net cifar_deep_kw --batch 1" "--net cifar_base_kw --batch 8"; do
  echo "$cfg"
  for knob in 0 1; do export GNNB_TOP_FUSE_UPD=$knob; echo -n "  fuse=$knob"; STEPS=$([[ "$cfg" == *"batch 1"* || "$cfg" == *"batch 8"* ]] && echo 200 || echo 30) run $cfg; done
done
