#!/bin/bash
# Where the step goes that is not a kernel: on ONE box in ONE call, the step without either side stream (tools/step_ablate.py
# "neither") against the sum of the main-queue kernels' own durations in a --serial rocprofv3 trace of bench.py - the
# difference is what the ~163 kernel boundaries of the chain cost.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python tools/step_ablate.py 2>&1 | grep -v amdgpu | tail -4
bash tools/prof.sh gapS --serial > /dev/null 2>&1
python3 - <<'PY'
import csv
side_pat=('skinny','layernorm_fwd_kernel<float','layernorm_bwd_kernel<float','attn_fwd_kernel<float','attn_bwd_d','lora_grad_mfma','reduce_multi','text_','lora_pack','eval_pairs')
init_pat=('copyBuffer','FillFunctor','transpose_cast','cast_from_f32','pack_b_kernel','rocblas','reduce_kernel','fillBuffer','BinaryFunctor','bfloat16_copy','CUDAFunctor','elementwise_kernel_manual_unroll<128, 4, at::native::gpu_kernel_impl_nocast')
rows=[]
for r in csv.DictReader(open('gpurun_out/gapS_kernel_trace.csv')):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sg=[i for i,r in enumerate(rows) if 'sgd_n_kernel' in r[2]]
# steps between consecutive sgd kernels (last 8)
for a,b in list(zip(sg[:-1],sg[1:]))[-6:]:
    seg=rows[a+1:b+1]
    main=[r for r in seg if not any(p in r[2] for p in side_pat+init_pat)]
    print("kernels %d main %d: main kernel time %.0f us, all kernel time %.0f us, span %.0f us"%(len(seg),len(main),sum(r[1]-r[0] for r in main)/1e3,sum(r[1]-r[0] for r in seg)/1e3,(rows[b][1]-rows[a][1])/1e3))
PY
