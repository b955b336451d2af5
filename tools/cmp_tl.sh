GNNB_LIB=$GRAFT_REPO_ROOT/tools/ablate/head.so bash tools/timeline.sh headso > /dev/null 2>&1
echo "head.so: $(grep 'k_gather' gpurun_out/timeline_headso.txt | awk '{print $6}' | tr '\n' ' ')"
bash tools/timeline.sh cur > /dev/null 2>&1
echo "current: $(grep 'k_gather' gpurun_out/timeline_cur.txt | awk '{print $6}' | tr '\n' ' ')"
GNNB_LIB=$GRAFT_REPO_ROOT/tools/ablate/head.so bash tools/timeline.sh headso > /dev/null 2>&1
echo "head.so: $(grep 'k_gather' gpurun_out/timeline_headso.txt | awk '{print $6}' | tr '\n' ' ')"
