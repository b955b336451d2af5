R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for a in ${ABL_LIST}; do for c in 2 3; do
GNNB_LIB=$R/tools/ablate/$a.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$a$c -- python3 $R/bench.py --no-cpu-baseline --config $c > $O/b_$a$c.json 2> $O/s_$a$c.log || { echo "failed $a $c"; tail -3 $O/s_$a$c.log; }
echo "$a config $c: $(grep k_top $(ls $O/s_$a$c/*/*kernel_stats.csv | head -1) | cut -d, -f4)"; rm -rf $O/s_$a$c
done; done
