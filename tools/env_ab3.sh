#!/bin/bash
# A/B/C of environment settings on the bench workload's step time, alternating, in ONE gpurun call:
#   tools/env_ab3.sh [pairs] "VAR=a VAR2=b" "VAR=c" ...     -> tools/step_time.py under each setting in turn, [pairs] times
R=${GRAFT_REPO_ROOT:-/root/repo}
n=$1; shift
for rep in $(seq $n); do
    for v in "$@"; do
        echo "== $v: $(env $v python3 $R/tools/step_time.py 2>&1 | tail -1)"
    done
done
