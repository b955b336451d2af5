GNNB_LIB=$GRAFT_REPO_ROOT/tools/ablate/interleave.so bash tools/timeline.sh il > /dev/null 2>&1
echo "== interleave: $(grep 'k_gather' gpurun_out/timeline_il.txt | awk '{print $6}' | tr '\n' ' ')"
bash tools/timeline.sh full > /dev/null 2>&1; echo "== full: $(grep 'k_gather' gpurun_out/timeline_full.txt | awk '{print $6}' | tr '\n' ' ')"
bash tools/ab.sh "GNNB_LIB=tools/ablate/interleave.so" "X=1" "GNNB_LIB=tools/ablate/interleave.so" "X=1"
