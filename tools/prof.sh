#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py on the GPU box:  tools/prof.sh <name> [--serial] [extra bench.py flags]
# -> gpurun_out/<name>_kernel_stats.csv, <name>_kernel_trace.csv (copy the stats into profiles/ to keep them)
set -e
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o run -- python3 $R/bench.py --steps 10 --warmup 3 \
    --no-cpu-baseline --no-trainer --no-secondary --no-roofline "$@" > $R/gpurun_out/${name}.log 2>&1
f=$(find /tmp/prof_$name -name 'run_kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${name}_kernel_stats.csv
cp "$(dirname $f)/run_kernel_trace.csv" $R/gpurun_out/${name}_kernel_trace.csv
tail -1 $R/gpurun_out/${name}.log | cut -c1-300
