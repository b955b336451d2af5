#!/bin/bash
# dev helper (GPU box): online-learning step after a change -- tests, step latency, launch count
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_online; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_online.py tests/test_lp_producer.py -m gpu -x -q > $O/test.log 2>&1; echo "rc=$?"; tail -4 $O/test.log
for net in cifar_base_kw cifar_wide_kw cifar_deep_kw; do timeout -k 10 200 python3 tools/online_rate.py $net 2>&1 | grep "online step"; done | tee $O/rate.log
cd /tmp && export TMPDIR=/tmp
ONLINE_B=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/online_rate.py cifar_base_kw > $O/prof.log 2>&1
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/online_kernel_stats.csv; rm -rf $O/prof
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('$O/online_kernel_stats.csv'))]
steps=3+10+10
tot=sum(int(r['Calls']) for r in rows if 'k_t' in r['Name'] or 'gnnb_train' in r['Name'])
print('training-form kernel launches per step ~', tot/steps)
for r in rows[:14]: print('  %-70s calls %6s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
