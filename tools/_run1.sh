cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/full1.log 2>&1) 2> gpurun_out/full1.time; echo "rc=$?" >> gpurun_out/full1.log
tail -25 gpurun_out/full1.log; cat gpurun_out/full1.time
