#!/bin/bash
# dev helper (GPU box): the round's measurement recipes in ONE script.   tools/run.sh <recipe> [args]
#   dist [tag]            bench.py --gpus 1 --dist for BASELINE config 4 (deep, 128 per rank) and config 2 (base, 256): the RCCL path at world size 1
#   aggonly [tag]         the stand-alone edge-aggregation leg of base / wide / deep: kernel stats + FETCH / WRITE passes (tools/profile_aggonly.sh)
#   full [tag]            tools/profile_all.sh for base / wide / deep (bench JSON with cpu_baseline, kernel stats, all PMC passes)
#   bench [tag] [args]    one bench line (no side legs) -> gpurun_out/<tag>.json
#   ab <libA> <libB> [bench args]   same-box A/B of two libraries (GNNB_LIB), three alternating runs each
R=$GRAFT_REPO_ROOT; cd $R
recipe=$1; shift
case $recipe in
  dist)
    TAG=${1:-r04}; O=$R/gpurun_out/dist_$TAG; mkdir -p $O
    timeout -k 10 400 python3 bench.py --gpus 1 --dist --config 4 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only > $O/deep_B128_dist.json 2> $O/deep_B128_dist.log || { echo "dist deep failed"; tail -20 $O/deep_B128_dist.log; exit 1; }
    timeout -k 10 400 python3 bench.py --gpus 1 --dist --config 2 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only > $O/base_B256_dist.json 2> $O/base_B256_dist.log || { echo "dist base failed"; tail -20 $O/base_B256_dist.log; exit 1; }
    python3 - <<PY
import json
for n in ("deep_B128_dist", "base_B256_dist"):
    d = json.loads(open("$O/" + n + ".json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["dist"])
PY
    ;;
  aggonly)
    TAG=${1:-r04}
    bash tools/profile_aggonly.sh ${TAG}_base_aggonly --config 2 && bash tools/profile_aggonly.sh ${TAG}_wide_aggonly --config 3 && bash tools/profile_aggonly.sh ${TAG}_deep_aggonly --config 4
    ;;
  full)
    TAG=${1:-r04}
    bash tools/profile_all.sh ${TAG}_base --config 2 && bash tools/profile_all.sh ${TAG}_wide --config 3 && bash tools/profile_all.sh ${TAG}_deep --config 4
    ;;
  bench)
    TAG=${1:-bench}; shift
    timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-exact-fp32 --no-aggregate-only "$@" > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err || { tail -5 $R/gpurun_out/$TAG.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$R/gpurun_out/$TAG.json').read().strip().splitlines()[-1]); print('$TAG', d['ms_per_step'], {n: v['avg_us'] for n, v in d['kernels'].items()})"
    ;;
  ab)
    A=$1; B=$2; shift; shift
    for rep in 1 2 3; do for L in $A $B; do
      GNNB_LIB=$R/$L timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$L', d['ms_per_step'], {n: v['avg_us'] for n, v in k.items()})" || exit 1
    done; done
    ;;
  *) echo "unknown recipe $recipe"; exit 2;;
esac
