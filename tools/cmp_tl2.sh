GNNB_LIB=$GRAFT_REPO_ROOT/tools/ablate/oldrange.so bash tools/timeline.sh oldr > /dev/null 2>&1
echo "oldrange: $(grep 'k_gather' gpurun_out/timeline_oldr.txt | awk '{print $6}' | tr '\n' ' ')"
bash tools/timeline.sh cur > /dev/null 2>&1
echo "current : $(grep 'k_gather' gpurun_out/timeline_cur.txt | awk '{print $6}' | tr '\n' ' ')"
