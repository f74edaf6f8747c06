cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/full2.log 2>&1) 2> gpurun_out/full2.time; echo "rc=$?" >> gpurun_out/full2.log
tail -14 gpurun_out/full2.log
grep "AUC per round\|oracle with" gpurun_out/full2.log
bash tools/pmc.sh r03
python3 tools/pmc_traffic.py gpurun_out/r03_fetch.csv gpurun_out/r03_write.csv > gpurun_out/r03_traffic.json
python3 tools/sq_counters.py gpurun_out/r03_sq.csv > gpurun_out/r03_sq_counters.json
rm -f gpurun_out/r03_fetch.csv gpurun_out/r03_write.csv gpurun_out/r03_sq.csv
head -c 1500 gpurun_out/r03_traffic.json
