R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ablb1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for a in ${ABL_LIST}; do
GNNB_LIB=$R/tools/ablate/$a.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$a -- python3 $R/bench.py --no-cpu-baseline --net cifar_base_kw --batch 1 --steps 200 --warmup 20 > $O/b_$a.json 2> $O/s_$a.log || { echo "failed $a"; tail -3 $O/s_$a.log; }
echo "$a B=1: $(grep k_top $(ls $O/s_$a/*/*kernel_stats.csv | head -1) | cut -d, -f4)"; rm -rf $O/s_$a
done
