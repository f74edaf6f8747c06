#!/bin/bash
# Load phase vs compute phase of the attn2 kernels: rocprofv3 per-kernel averages of tools/bench_attn.py on the product
# library and on the two ablation twins (FFM_ATTN_ABL=1: return behind the load phase, =2: no tile DMA); the twins are built
# by hand into tools/proto/libffm_abl{1,2}.so:
#   cd fairfedmed_amd/csrc; for a in 1 2; do hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DFFM_ATTN_ABL=$a -c attention.hip \
#     -o /tmp/attention_abl$a.o; hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/proto/libffm_abl$a.so \
#     $(ls *.o | grep -v stamps | grep -v "^attention.o") /tmp/attention_abl$a.o; done
# Round 3 (us, full / loads only / no DMA): dK/dV 21.5 / 4.7 / 23.3, dQ 20.1 / 5.6 / 16.3, forward 14.8 / 5.5 / 12.3 - the compute
# phase, not the tile load, is what these kernels spend their time in.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in full abl1 abl2; do
    if [ $v = full ]; then unset FFM_LIB_PATH; else export FFM_LIB_PATH=$R/tools/proto/libffm_$v.so; fi
    rm -rf /tmp/attn_$v
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attn_$v -o run -- python3 $R/tools/bench_attn.py > /dev/null 2>&1
    f=$(find /tmp/attn_$v -name 'run_kernel_stats.csv' | head -1)
    echo "== $v"
    python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'attn2' in r['Name']: print('%-48s calls %5s avg %6.1f us' % (r['Name'].split('(')[1][-40:] if False else r['Name'][28:70], r['Calls'], float(r['AverageNs'])/1e3))
"
done
