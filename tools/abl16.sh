#!/bin/bash
# dev helper (GPU box): average time of the plain / embedding 16-node gathers under the ablation builds of tools/mk_abl16.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" noload nomfma nostore none; do
  if [ -z "$v" ]; then unset GNNB_LIB; else export GNNB_LIB=$R/tools/ablate/g16_$v.so; fi
  O=$R/gpurun_out/abl16_${v:-base}; rm -rf $O; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/run.log 2>&1 || { echo "$v failed"; tail -3 $O/run.log; exit 1; }
  f=$(ls $O/*/*kernel_stats.csv | head -1)
  echo "== ${v:-base}: $(grep 'k_gather16<false, false>' $f | cut -d, -f2-4)"
  rm -rf $O
done
