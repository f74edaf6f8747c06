#!/bin/bash
# PMC passes of bench.py on the GPU box (separate passes, --kernel-trace only, as the guide prescribes):
#   tools/pmc.sh <name> [extra bench.py flags, e.g. --config c4]  ->  gpurun_out/<name>_fetch.csv, _write.csv, _sq.csv
#   PMC_PASSES="fetch write" limits the passes
set -e
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 2 --serial --no-cpu-baseline --no-trainer --no-secondary --no-roofline --engine-step $@"
for pass in ${PMC_PASSES:-fetch write sq}; do
    case $pass in
        fetch) C="FETCH_SIZE";;
        write) C="WRITE_SIZE";;
        sq) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE";;
    esac
    rm -rf /tmp/pmc_$pass
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$pass -o run -- $B > $R/gpurun_out/${name}_$pass.log 2>&1
    f=$(find /tmp/pmc_$pass -name 'run_counter_collection.csv' | head -1)
    cp "$f" $R/gpurun_out/${name}_$pass.csv
    echo "$pass done: $(wc -l < $R/gpurun_out/${name}_$pass.csv) rows"
done
