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
    TAG=${1:-r05}; O=$R/gpurun_out/dist_$TAG; mkdir -p $O
    timeout -k 10 400 python3 bench.py --gpus 1 --dist --config 4 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only > $O/deep_B128_dist.json 2> $O/deep_B128_dist.log || { echo "dist deep failed"; tail -20 $O/deep_B128_dist.log; exit 1; }
    timeout -k 10 400 python3 bench.py --gpus 1 --dist --config 2 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only > $O/base_B256_dist.json 2> $O/base_B256_dist.log || { echo "dist base failed"; tail -20 $O/base_B256_dist.log; exit 1; }
    python3 - <<PY
import json
for n in ("deep_B128_dist", "base_B256_dist"):
    d = json.loads(open("$O/" + n + ".json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d["dist"])
    print(n, "two handles in flight + all-gather:", d["dist"].get("dist_two_in_flight"))
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
  top_phases)
    # LDS bank conflicts of k_top by phase: libraries tools/ablate/r4top{1,2,6,7,8}.so (tools/mklib.sh WORK r4topN -DTOP_STOP=N: k_top leaves
    # before F1 / after F1 / after F2 / after F3 / after B1) and the shipped one; counters of the k_top dispatches under each
    O=$R/gpurun_out/top_phases; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
    for L in r4top1 r4top2 r4top6 r4top7 r4top8 full; do
      [ $L = full ] && unset GNNB_LIB || export GNNB_LIB=$R/tools/ablate/$L.so
      timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/$L -- python3 $R/tools/top_phases.py "$@" > $O/$L.log 2>&1 || { echo "$L failed"; tail -3 $O/$L.log; exit 1; }
    done
    python3 - <<PY
import csv, glob, collections
prev = None
print("%-8s %10s %14s %14s %12s %12s" % ("stop", "k_top us", "bank_conflict", "active_lds", "insts_lds", "insts_valu"))
for L, name in (("r4top1", "lists"), ("r4top2", "+F1"), ("r4top6", "+F2"), ("r4top7", "+F3"), ("r4top8", "+B1"), ("full", "+B2+upd")):
    vals = collections.defaultdict(list)
    for f in glob.glob("$O/%s/*/*_counter_collection.csv" % L):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(("void k_top", "k_top")): vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob("$O/%s/*/*_kernel_trace.csv" % L):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(("void k_top", "k_top")): dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    m = {k: sum(v) / len(v) for k, v in vals.items()}
    row = (sum(dur[2:]) / max(1, len(dur[2:])), m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_ACTIVE_INST_LDS", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_VALU", 0))
    d = tuple(a - b for a, b in zip(row, prev)) if prev else row
    print("%-8s %10.1f %14.0f %14.0f %12.0f %12.0f   (this phase: %.1f us, %.0f conflict cycles, %.0f LDS instructions)" % ((name,) + row + (d[0], d[1], d[3])))
    prev = row
PY
    ;;
  *) echo "unknown recipe $recipe"; exit 2;;
esac
