for q in 0 1 2 3 4 5; do
  GNNB_LS_ONLY=$q bash tools/timeline.sh ls$q > /dev/null 2>&1
  echo "job $q: $(grep k_livesum gpurun_out/timeline_ls$q.txt)"
done
