for n in none nostage notiles; do
  GNNB_NO_GATHER16=1 GNNB_LIB=$GRAFT_REPO_ROOT/tools/ablate/abl_$n.so bash tools/timeline.sh abl_$n > /dev/null 2>&1
  echo "== $n: $(grep 'k_gather<' gpurun_out/timeline_abl_$n.txt | awk '{print $6}' | tr '\n' ' ')"
done
