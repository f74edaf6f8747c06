#!/usr/bin/env python3
"""What the FairLoRA epilogues of the 128x128 kernel cost on RN50's short-K 1x1 convolutions: plain / + LoRA update from
given ts / + fused down projection (RANKOP) / + dS partials, at layer1 .. layer4 shapes (bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

dt = torch.bfloat16
r, G = 8, 2


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (M, N, K, rps) in [(100352, 256, 64, 3136), (100352, 64, 256, 3136), (25088, 512, 128, 784), (6272, 1024, 256, 196),
                       (6272, 256, 1024, 196), (1568, 512, 2048, 49)]:
    g = torch.Generator("cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    b = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    res = torch.randn(M, N, device="cuda", generator=g).to(dt)
    ts = torch.randn(M, r, device="cuda", generator=g)
    lw = torch.randn(r, N, device="cuda", generator=g)
    lwk = torch.randn(N, r, device="cuda", generator=g)
    P = torch.randn(K, r, device="cuda", generator=g) * 0.1
    rk = torch.zeros(16, K, device="cuda", dtype=dt)
    ops.PackPlan([(P, False, rk)], dt, "cuda").run()
    S = torch.randn(G, r, device="cuda", generator=g)
    attr = torch.randint(0, G, ((M + rps - 1) // rps,), device="cuda", dtype=torch.int32)
    t, tso = torch.empty(M, r, device="cuda"), torch.empty(M, r, device="cuda")
    tf = torch.randn(M, r, device="cuda", generator=g)
    dsp = torch.empty(ops.gemm_tiles_m(M), G, r, device="cuda")
    ro_f = ops.RankOp(rk, S, attr, rps, 0.25, 0.7, t_out=t, ts_out=tso)
    ro_b = ops.RankOp(rk, S, attr, rps, 0.25, 0.7, ts_out=tso, t_fwd=tf, ds_part=dsp)
    u = [bench(lambda: ops.gemm_nt(a, b, out)),
         bench(lambda: ops.gemm_nt(a, b, out, ts=ts, lw=lw)),
         bench(lambda: ops.gemm_nt(a, b, out, lw=lw, rankop=ro_f)),
         bench(lambda: ops.gemm_nt(a, b, out, lw=lwk, lw_is_kr=True, rankop=ro_b)),
         bench(lambda: ops.gemm_nt(a, b, out, lw=lwk, lw_is_kr=True, rankop=ro_b, res=res))]
    by = (M * K + N * K + M * N) * 2
    print(f"M {M:6d} N {N:4d} K {K:4d}: plain {u[0]:6.1f} us ({by / u[0] / 1e3:5.0f} GB/s) | +lora(ts) {u[1]:6.1f} | rankop fwd {u[2]:6.1f} | "
          f"rankop bwd+dS {u[3]:6.1f} | +res {u[4]:6.1f}", flush=True)
