#!/bin/bash
# dev helper (GPU box): bench step time under a list of env settings:  tools/ab.sh "A=1" "B=2 C=3" ...
for cfg in "$@"; do
  r=$(env $cfg python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], round(d['value']/1e6,2))")
  echo "$cfg -> $r"
done
