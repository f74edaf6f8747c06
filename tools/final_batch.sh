set -o pipefail
V=$1
cd $GRAFT_REPO_ROOT
# tools/final_batch.sh <name> [notests]: the round's evidence in one gpurun call (the test suite alone takes ~6 minutes)
if [ "$2" != "notests" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/${V}_final_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/${V}_final_gpu_tests.log
fi
timeout -k 10 400 python bench.py > gpurun_out/${V}_bench.json 2> gpurun_out/${V}_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/${V}_bench.json
timeout -k 10 300 bash tools/prof.sh ${V} --engine-step && python tools/trace_timeline.py gpurun_out/${V}_kernel_trace.csv 8 > gpurun_out/${V}_timeline.txt
timeout -k 10 300 bash tools/prof.sh ${V}_serial --serial --engine-step
timeout -k 10 600 bash tools/pmc.sh ${V} && python tools/pmc_traffic.py gpurun_out/${V}_fetch.csv gpurun_out/${V}_write.csv > gpurun_out/${V}_traffic.json && python tools/sq_counters.py gpurun_out/${V}_sq.csv > gpurun_out/${V}_sq_counters.json
PMC_PASSES="fetch write" timeout -k 10 400 bash tools/pmc.sh ${V}_c4 --config c4 && python tools/pmc_traffic.py gpurun_out/${V}_c4_fetch.csv gpurun_out/${V}_c4_write.csv > gpurun_out/${V}_traffic_c4.json
PMC_PASSES="fetch write" timeout -k 10 400 bash tools/pmc.sh ${V}_c5 --config c5 && python tools/pmc_traffic.py gpurun_out/${V}_c5_fetch.csv gpurun_out/${V}_c5_write.csv 40 > gpurun_out/${V}_traffic_c5.json
FFM_SERIAL=1 timeout -k 10 300 bash tools/prof_tool.sh ${V}_rn50_serial bench_rn50.py 32 10
FFM_SERIAL=1 timeout -k 10 300 bash tools/prof_tool.sh ${V}_oct3d_serial bench_oct3d.py --json
if [ -n "$ATTN_PROFILES" ]; then   # round 4's attention evidence (needs tools/attn_phases.sh's stamps twin rebuilt for the current ABI)
(export FFM_LIB_PATH=$PWD/tools/proto/libffm_a3stamps.so; timeout -k 10 200 python tools/attn_stamps.py fwd; timeout -k 10 200 python tools/attn_stamps.py dkv) > gpurun_out/${V}_attn3_stamps.txt 2>&1
(timeout -k 10 200 bash tools/attn_pmc.sh attn3; FFM_ATTN=v2 timeout -k 10 200 bash tools/attn_pmc.sh attn2) > gpurun_out/${V}_attn_sq.txt 2>&1
(for g in v2 v3; do FFM_ATTN=$g timeout -k 10 120 python tools/bench_attn.py; done; ATTN_SETS=1 timeout -k 10 120 python tools/bench_attn.py; timeout -k 10 120 python tools/bench_attn.py fp16) > gpurun_out/${V}_attn_bench.txt 2>&1
fi
rm -f gpurun_out/*_kernel_trace.csv gpurun_out/${V}*_fetch.csv gpurun_out/${V}*_write.csv gpurun_out/${V}_sq.csv
ls gpurun_out | grep ${V}
