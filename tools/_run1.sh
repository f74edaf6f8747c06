cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_ot_head_gpu.py tests/test_trainer_gpu.py tests/test_evaluator_gpu.py tests/test_edge_gpu.py tests/test_conv_gpu.py tests/test_engine_gpu.py tests/test_engine_rn_gpu.py -x -q 2>&1 | tail -3
