#!/bin/bash
# dev helper (GPU box): rocprofv3 kernel stats of a short bench run under each given env setting:  tools/kstats.sh "A=1" "B=2" [pattern]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  O=$R/gpurun_out/kstats_tmp; rm -rf $O; mkdir -p $O
  for kv in $cfg; do export "$kv"; done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/run.log 2>&1 || { echo "$cfg failed"; tail -3 $O/run.log; exit 1; }
  f=$(ls $O/*/*kernel_stats.csv | head -1)
  echo "== $cfg: $(python3 -c "import json;print(json.loads(open('$O/run.log').read().split(chr(10))[[i for i,l in enumerate(open('$O/run.log').read().split(chr(10))) if l.startswith('{')][0]])['ms_per_step'])")"
  python3 -c "
import csv,sys
for r in csv.reader(open('$f')):
    if '${KPAT:-k_node_update}' in r[0]: print('   %-60s calls %5s avg %8.1f us' % (r[0][:60], r[1], float(r[3]) / 1e3))"
  rm -rf $O
done
