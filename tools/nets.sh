#!/bin/bash
# dev helper (GPU box): step time and kernel table of the three networks
for net in cifar_base_kw cifar_wide_kw cifar_deep_kw; do
  b=256; [ $net = cifar_deep_kw ] && b=128
  python bench.py --net $net --batch $b --no-cpu-baseline > gpurun_out/bench_$net.json 2>/dev/null
  python tools/kern_table.py gpurun_out/bench_$net.json
done
