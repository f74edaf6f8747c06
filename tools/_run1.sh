cd $GRAFT_REPO_ROOT
FFM_ATTN_PARTS=4 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or attn" 2>&1 | tail -2
for i in 1 2; do
echo parts2; timeout -k 10 200 python tools/bench_attn.py 2>&1 | grep -v amdgpu
echo parts4; FFM_ATTN_PARTS=4 timeout -k 10 200 python tools/bench_attn.py 2>&1 | grep -v amdgpu
done
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-trainer --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('parts2 ms/step', j['ms_per_step'])"
FFM_ATTN_PARTS=4 timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-trainer --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('parts4 ms/step', j['ms_per_step'])"
done
