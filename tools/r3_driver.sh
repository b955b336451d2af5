#!/bin/bash
# dev helper (GPU box): what the driver runs at round end -- the whole -m gpu suite, smoke(), the default bench line (timed)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_driver; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/test.log 2>&1; echo "pytest rc=$?"; tail -3 $O/test.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
T0=$(date +%s); timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? wall $(( $(date +%s) - T0 )) s"; head -c 600 $O/bench.json; echo
python3 - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().split('\n')[-1])
print({k: d[k] for k in ('value','ms_per_step','exact_fp32_ms_per_step','bf3_max_abs_delta')})
print(d['roofline']); print(d['roofline_message_passing']['frac'], d['roofline_message_passing']['frac_counter'])
for k,v in (d['binding_bounds'] or {}).items(): print(k, v['avg_launch_us'], v['sol_us'], v['binding'], v['frac_of_binding_bound'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['sample'][:120])
PY
