#!/bin/bash
# dev helper (GPU box): per-dispatch timeline of the last bench step -> gpurun_out/timeline_<tag>.txt
R=$GRAFT_REPO_ROOT; TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/tl_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$TAG -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/tl_$TAG/run.log 2>&1 || { echo "trace failed"; tail -5 $R/gpurun_out/tl_$TAG/run.log; exit 1; }
python3 $R/tools/timeline.py $R/gpurun_out/tl_$TAG > $R/gpurun_out/timeline_$TAG.txt
cat $R/gpurun_out/timeline_$TAG.txt
