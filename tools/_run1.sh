cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_trainer_gpu.py tests/test_engine_rn_gpu.py tests/test_edge_gpu.py tests/test_ot_head_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-trainer --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms/step', j['ms_per_step'])"; done
python bench.py --config c5 --steps 20 --warmup 3 --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c5 ms/step', j['ms_per_step'])"
