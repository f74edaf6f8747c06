set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout -k 10 1100 python -m pytest tests/test_auc_parity_gpu.py tests/test_fullsize_gpu.py tests/test_trainer_gpu.py tests/test_engine_rn_gpu.py -q -s -k "rccl or auc or 3d_oct or control or trajectory_fp32" --durations=15 > gpurun_out/t2.log 2>&1) 2> gpurun_out/t2.time; echo "rc=$?" >> gpurun_out/t2.log
grep -v "^\s*$" gpurun_out/t2.log | grep "AUC per round\|oracle with\|rn50 bf16\|oct3d\|passed\|failed\|FAILED" | tail -30
