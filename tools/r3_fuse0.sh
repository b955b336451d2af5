R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuse0; mkdir -p $O
cd /tmp && export TMPDIR=/tmp GNNB_FUSE=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 $R/bench.py --no-cpu-baseline --no-exact-fp32 --no-aggregate-only > $O/bench.json 2> $O/err.log
cp $(ls $O/s/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/s
cut -d, -f1,2,4 $O/kernel_stats.csv | head -16
