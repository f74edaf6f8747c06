cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FFM_SERIAL=1 bash tools/prof_tool.sh r03_rn50_serial bench_rn50.py 32 10
bash tools/prof_tool.sh r03_rn50 bench_rn50.py 32 10
FFM_SERIAL=1 bash tools/prof_tool.sh r03_oct3d_serial bench_oct3d.py --json
rm -f gpurun_out/*_kernel_trace.csv
python bench.py --config c5 --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-400
