R=$GRAFT_REPO_ROOT; TAG=r01; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd $R && bash tools/pmc.sh $TAG > $O/pmc.log 2>&1
python3 tools/pmc_table.py $TAG $O/pmc_summary.json > $O/pmc_table.txt 2>&1
rm -f $R/gpurun_out/pmc_$TAG/p*/*/*agent_info.csv
tail -3 $O/pmc.log; head -c 400 $O/pmc_summary.json
