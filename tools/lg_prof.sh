#!/bin/bash
# Per-kernel price of FFM_EPI_LGRAD (the two large LoRA-gradient reductions inside the dX product of c_proj): rocprofv3 kernel
# averages of bench.py with FFM_LGRAD=0 / 1, overlapped and --serial, side by side (GPU box).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in 0 1; do
  FFM_LGRAD=$v bash tools/prof.sh lg${v} > /dev/null 2>&1
  FFM_LGRAD=$v bash tools/prof.sh lg${v}s --serial > /dev/null 2>&1
done
python3 - <<'PY'
import csv
def load(p):
    return {r['Name']:(int(r['Calls']),float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e3) for r in csv.DictReader(open(p))}
for mode in ('', 's'):
    a,b=load('gpurun_out/lg0%s_kernel_stats.csv'%mode),load('gpurun_out/lg1%s_kernel_stats.csv'%mode)
    print('==== ', 'serial' if mode else 'overlapped', ' (LGRAD 0 -> 1), us per step')
    keys=sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0,0))[2]+b.get(k,(0,0,0))[2]))
    for k in keys[:26]:
        x,y=a.get(k,(0,0,0)),b.get(k,(0,0,0))
        print("%5d x %7.2f | %5d x %7.2f   tot/step %8.1f -> %8.1f  %s"%(x[0],x[1],y[0],y[1],x[2]/13,y[2]/13,k[:80]))
    print('sum', sum(v[2] for v in a.values())/13, sum(v[2] for v in b.values())/13)
PY
