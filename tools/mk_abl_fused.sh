#!/bin/bash
# dev helper: ablation builds of k_gather_update -> tools/ablate/fused_<waves>_<mask>.so  (results are wrong; timing only)
#   FUSED_ABL bits: 1 no row loads, 2 no gather MFMAs, 4 no chain, 8 no compaction
R=/root/repo; mkdir -p $R/tools/ablate
for cfg in "$@"; do
  w=${cfg%%_*}; m=${cfg##*_}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DFUSED_WAVES=$w -DFUSED_ABL=$m -o $R/tools/ablate/fused_${w}_${m}.so $R/gnn_branching_amd/csrc/gnnb.hip 2>&1 | grep -i " error" &
done
wait
ls -la $R/tools/ablate/fused_*.so
