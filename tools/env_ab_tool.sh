#!/bin/bash
# tools/env_ab_tool.sh <pairs> "<tool and args>" "VAR=a" "VAR=b" ...: like env_ab3.sh for any tool under tools/ that prints a JSON
# line or an "ms/step" line last (bench_oct3d.py --json, bench_rn50.py ...), alternating the settings in ONE gpurun call
R=${GRAFT_REPO_ROOT:-/root/repo}
n=$1; tool=$2; shift 2
for rep in $(seq $n); do
    for v in "$@"; do
        echo "== $v: $(env $v python3 $R/tools/$tool 2>&1 | tail -1 | cut -c1-200)"
    done
done
