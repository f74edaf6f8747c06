#!/bin/bash
# Load phase vs compute phase of the attention kernels: rocprofv3 per-kernel averages of tools/bench_attn.py on the product
# library and on the two ablation twins (=1: return behind the load phase, =2: no tile DMA); the twins are built here (hipcc
# cross-compiles without a GPU) into tools/proto/libffm_abl{1,2}.so and travel with the snapshot:
#   tools/attn_phases.sh build          (build container; attention3.hip with -DFFM_ATTN3_ABL, attention.hip with -DFFM_ATTN_ABL)
#   tools/attn_phases.sh [attn3|attn2]  (GPU box)
# Round 3, attn2 (us, full / loads only / no DMA): dK/dV 21.5 / 4.7 / 23.3, dQ 20.1 / 5.6 / 16.3, forward 14.8 / 5.5 / 12.3.
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
    cd $R/fairfedmed_amd/csrc
    M="-include _gen_rename_main.h"             # the product objects' entry points are renamed (fairfedmed_amd/build.py)
    for a in 1 2; do
        hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-honor-nans -DFFM_ATTN3_ABL=$a $M -c attention3.hip -o /tmp/attention3_abl$a.o
        hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DFFM_ATTN_ABL=$a $M -c attention.hip -o /tmp/attention_abl$a.o
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/proto/libffm_abl$a.so \
            $(ls *.o | grep -v stamps | grep -v "^attention.o" | grep -v "^attention3.o") /tmp/attention_abl$a.o /tmp/attention3_abl$a.o
    done
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-honor-nans -DFFM_ATTN3_STAMPS $M -c attention3.hip -o /tmp/attention3_stamps.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/proto/libffm_a3stamps.so $(ls *.o | grep -v stamps | grep -v "^attention3.o") /tmp/attention3_stamps.o
    exit 0
fi
pat=${1:-attn3}
[ $pat = attn2 ] && export FFM_ATTN=v2
cd /tmp && export TMPDIR=/tmp
for v in full abl1 abl2; do
    if [ $v = full ]; then unset FFM_LIB_PATH; else export FFM_LIB_PATH=$R/tools/proto/libffm_$v.so; fi
    rm -rf /tmp/attn_$v
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attn_$v -o run -- python3 $R/tools/bench_attn.py ${@:2} > /dev/null 2>&1
    f=$(find /tmp/attn_$v -name 'run_kernel_stats.csv' | head -1)
    echo "== $v"
    python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if '$pat' in r['Name']: print('%-44s calls %5s avg %6.1f us' % (r['Name'][22:64], r['Calls'], float(r['AverageNs'])/1e3))
"
done
