cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_trainer_gpu.py tests/test_ot_head_gpu.py tests/test_edge_gpu.py tests/test_engine_rn_gpu.py -x -q -k "not rccl" > gpurun_out/t3.log 2>&1; echo "rc=$?" >> gpurun_out/t3.log
tail -15 gpurun_out/t3.log
for i in 1 2; do timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-trainer --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms/step', j['ms_per_step'], 'enqueue', j['config']['host_enqueue_ms_per_step'])"; done
