cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "text_tower" 2>&1 | tail -2
bash tools/prof.sh tt --serial > /dev/null 2>&1
python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/tt_kernel_stats.csv')):
    if 'text_' in r['Name']: print('%-60s calls %4s avg %6.1f us' % (r['Name'][23:80], r['Calls'], float(r['AverageNs'])/1e3))
"
rm -f gpurun_out/tt_kernel_trace.csv
