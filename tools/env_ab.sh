#!/bin/bash
# A/B of one environment switch on the bench workload's step time, alternating, in ONE gpurun call (boxes differ by 2-4 %):
#   tools/env_ab.sh FFM_LGRAD 0 1 [pairs]          -> tools/step_time.py under VAR=a, VAR=b, VAR=a, ...
R=${GRAFT_REPO_ROOT:-/root/repo}
var=$1; a=$2; b=$3; n=${4:-3}
for rep in $(seq $n); do
    for v in $a $b; do
        echo "== $var=$v: $(env $var=$v python3 $R/tools/step_time.py 2>&1 | tail -1)"
    done
done
