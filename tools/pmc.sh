#!/bin/bash
# dev helper (GPU box): collect SQ/TCC counters per dispatch in separate passes -> gpurun_out/pmc_<tag>/
#   [PMC_PASSES="3 4"] [GNNB_FUSE=0 ...] tools/pmc.sh <tag> [bench.py args]
# (counters only with --kernel-trace: gpurun refuses --pmc with other trace domains; environment variables set by the caller -- e.g. the
#  aggregate-only leg's GNNB_FUSE=0 GNNB_NO_EMBED_FUSE=1 -- are inherited by the profiled bench run; PMC_PASSES selects passes by number)
R=$GRAFT_REPO_ROOT; TAG=${1:-x}; shift
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum" ; do
  i=$((i+1))
  if [ -n "$PMC_PASSES" ] && ! echo " $PMC_PASSES " | grep -q " $i "; then continue; fi
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-aggregate-only --no-two-in-flight --no-host-fed "$@" > $R/gpurun_out/pmc_$TAG/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_$TAG/p$i.log; break; }
done
ls -R $R/gpurun_out/pmc_$TAG | head -30
