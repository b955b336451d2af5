#!/bin/bash
# dev helper (GPU box): per-kernel time of the online-learning step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/online_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && export ONLINE_B=${ONLINE_B:-1}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/online_rate.py cifar_base_kw > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats
grep "online step" $O/log.txt
head -16 $O/kernel_stats.csv | cut -c1-150
export ONLINE_B=1; python3 $R/tools/host_online.py 2>&1 | tail -3
