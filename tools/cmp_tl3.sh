GNNB_LIB=$GRAFT_REPO_ROOT/tools/ablate/gi12.so bash tools/timeline.sh gi12 > /dev/null 2>&1
echo "gi12: $(grep 'k_gather_input' gpurun_out/timeline_gi12.txt | awk '{print $6}' | tr '\n' ' ')"
bash tools/timeline.sh cur > /dev/null 2>&1
echo "cur : $(grep 'k_gather_input' gpurun_out/timeline_cur.txt | awk '{print $6}' | tr '\n' ' ')"
GNNB_LIB=$GRAFT_REPO_ROOT/tools/ablate/gi12.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
