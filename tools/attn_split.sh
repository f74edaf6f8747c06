#!/bin/bash
# Would ONE launch holding the blocks of both attention backward kernels beat two launches?  Twin of the library whose
# ffm_attention_bwd puts the dK/dV kernel on a second stream beside the dQ kernel (-DFFM_ATTN3_SPLIT_PROBE: it reads a stale
# delta, results wrong, timing only):   tools/attn_split.sh build  (build container);  tools/attn_split.sh  (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
    cd $R/fairfedmed_amd/csrc
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-honor-nans -DFFM_ATTN3_SPLIT_PROBE -include _gen_rename_main.h -c attention3.hip -o /tmp/attention3_split.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/proto/libffm_a3split.so $(ls *.o | grep -v stamps | grep -v "^attention3.o") /tmp/attention3_split.o
    exit 0
fi
for rep in 1 2 3; do
    unset FFM_LIB_PATH
    echo "== two launches in a row : $(python3 $R/tools/step_time.py | tail -1)"
    export FFM_LIB_PATH=$R/tools/proto/libffm_a3split.so
    echo "== dK/dV beside dQ       : $(python3 $R/tools/step_time.py | tail -1)"
done
