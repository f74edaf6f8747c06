cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q -k "not rn50" 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-trainer --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms/step', j['ms_per_step'])"
