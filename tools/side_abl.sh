#!/bin/bash
# Which resource of the side-stream kernels slows the vision chain down?  Twins of the library whose text-tower GEMM
# (gemm_skinny.hip, -DFFM_SKINNY_ABL) drops its weight loads (1), its activation loads (2) or both (3); the step is timed on
# each (results are garbage: timing only).   tools/side_abl.sh build  (build container);  tools/side_abl.sh  (GPU box)
# Then: the X3 product with 1 / 2 / 4 column tiles per block (FFM_SKINNY_NT), and the LoRA-gradient reductions of a block
# started behind its dX(c_fc) / its attention backward / the whole block (FFM_RED_AT).
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
    cd $R/fairfedmed_amd/csrc
    M="-include _gen_rename_main.h"
    for a in 1 2 3; do
        hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DFFM_SKINNY_ABL=$a -DFFM_SKINNY_NT_DEFAULT=1 $M -c gemm_skinny.hip -o /tmp/gemm_skinny_abl$a.o
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/proto/libffm_sk$a.so \
            $(ls *.o | grep -v stamps | grep -v "^gemm_skinny.o") /tmp/gemm_skinny_abl$a.o
    done
    exit 0
fi
if [ "$1" = abl ]; then
for rep in 1 2; do
for v in full sk1 sk2 sk3; do
    if [ $v = full ]; then unset FFM_LIB_PATH; else export FFM_LIB_PATH=$R/tools/proto/libffm_$v.so; fi
    echo "== $v: $(FFM_SKINNY_NT=1 python3 $R/tools/step_time.py)"
done
done
unset FFM_LIB_PATH
for rep in 1 2; do
for v in 0 1 2; do
    echo "== reductions start at $v: $(FFM_RED_AT=$v python3 $R/tools/step_time.py)"
done
done
fi
for rep in 1 2 3; do
for v in 1 2 4; do
    echo "== X3 tiles per block $v: $(FFM_SKINNY_NT=$v python3 $R/tools/step_time.py)"
done
done
