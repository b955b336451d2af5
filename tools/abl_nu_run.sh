for n in nomfma noload nostore; do
  GNNB_LIB=$GRAFT_REPO_ROOT/tools/ablate/nu_$n.so bash tools/timeline.sh nu_$n > /dev/null 2>&1
  echo "== $n: $(grep 'k_node_update' gpurun_out/timeline_nu_$n.txt | awk '{print $6}' | tr '\n' ' ')"
done
bash tools/timeline.sh full > /dev/null 2>&1; echo "== full: $(grep 'k_node_update' gpurun_out/timeline_full.txt | awk '{print $6}' | tr '\n' ' ')"
