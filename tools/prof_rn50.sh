#!/bin/bash
# rocprofv3 kernel-trace summary of tools/bench_rn50.py on the GPU box:  tools/prof_rn50.sh <name>
set -e
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o run -- python3 $R/tools/bench_rn50.py 32 10 "$@" > $R/gpurun_out/${name}.log 2>&1
f=$(find /tmp/prof_$name -name 'run_kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${name}_kernel_stats.csv
cp "$(dirname $f)/run_kernel_trace.csv" $R/gpurun_out/${name}_kernel_trace.csv
tail -1 $R/gpurun_out/${name}.log | cut -c1-300
