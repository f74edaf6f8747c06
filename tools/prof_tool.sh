#!/bin/bash
# rocprofv3 kernel-trace summary of one of the tools on the GPU box:  tools/prof_tool.sh <name> <tool.py> [args]
# -> gpurun_out/<name>_kernel_stats.csv, <name>_kernel_trace.csv   (FFM_SERIAL=1 in the environment: one stream)
set -e
name=$1; tool=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o run -- python3 $R/tools/$tool "$@" > $R/gpurun_out/${name}.log 2>&1
f=$(find /tmp/prof_$name -name 'run_kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${name}_kernel_stats.csv
cp "$(dirname $f)/run_kernel_trace.csv" $R/gpurun_out/${name}_kernel_trace.csv
grep -a "ms/step\|ms_per_step" $R/gpurun_out/${name}.log | tail -2 | cut -c1-300
