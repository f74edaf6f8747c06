cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r03_v1_bench.json 2> gpurun_out/r03_v1_bench.err; echo "bench rc=$?"
bash tools/prof.sh r03_v1_bench
bash tools/prof.sh r03_v1_serial --serial
python tools/trace_timeline.py gpurun_out/r03_v1_bench_kernel_trace.csv > gpurun_out/r03_v1_timeline.txt 2>&1 || true
rm -f gpurun_out/*_kernel_trace.csv
cut -c1-600 gpurun_out/r03_v1_bench.json
head -30 gpurun_out/r03_v1_timeline.txt
