cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only $@ 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('  ', d['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if n in ('k_top','k_node_update')})"; }
for cfg in "--config 4" "--net cifar_base_kw --batch 128" "--net cifar_wide_kw --batch 128" "--net cifar_base_kw --batch 96"; do
  echo "$cfg"
  for knob in 4 2; do export GNNB_TOP_SPLIT=$knob; echo -n "  split_max=$knob"; run $cfg; done
done
