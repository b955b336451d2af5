#!/bin/bash
# dev helper (GPU box): where gnnb_set_weights (the tail of an online step) spends its time, with and without helper threads
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_pack; mkdir -p $O
cd $R
lscpu | grep "Model name" > $O/pack.log
for th in 1 0; do
  echo "== GNNB_PACK_THREADS=$th" >> $O/pack.log
  GNNB_PACK_THREADS=$th GNNB_PACK_TIMING=1 timeout -k 10 200 python3 tools/host_online.py 2>&1 | tail -6 >> $O/pack.log
done
for net in cifar_base_kw; do for th in 1 0; do GNNB_PACK_THREADS=$th timeout -k 10 200 python3 tools/online_rate.py $net 2>&1 | grep "online step"; done; done >> $O/pack.log
cat $O/pack.log
