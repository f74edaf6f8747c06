cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or attn" 2>&1 | tail -2
for i in 1 2; do
echo old; FFM_LIB_PATH=$GRAFT_REPO_ROOT/tools/proto/libffm_old.so timeout -k 10 200 python tools/bench_attn.py 2>&1 | grep -v amdgpu
echo new; timeout -k 10 200 python tools/bench_attn.py 2>&1 | grep -v amdgpu
done
