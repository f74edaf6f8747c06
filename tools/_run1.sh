cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_trainer_gpu.py tests/test_edge_gpu.py -x -q -k "not rccl" 2>&1 | tail -3
bash tools/prof.sh tt --serial > /dev/null 2>&1
python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/tt_kernel_stats.csv')):
    if 'lora_down' in r['Name']: print('%-60s calls %4s avg %6.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
"
rm -f gpurun_out/tt_kernel_trace.csv
for i in 1 2; do timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-trainer --no-roofline 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms/step', j['ms_per_step'])"; done
