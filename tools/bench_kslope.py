#!/usr/bin/env python3
"""Panel GEMM time against K at fixed M, N: the slope is the cost of one K64 step, the intercept what a launch pays
outside its K loop (prologue, epilogue, stores)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

dt, M = torch.bfloat16, 6304


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for N in (768, 3072):
    pts = []
    for K in (512, 768, 1536, 3072, 6144):
        a = torch.randn(M, K, device="cuda").to(dt)
        b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        bp = ops.pack_b(b)
        bias = torch.randn(N, device="cuda")
        us = bench(lambda: ops.gemm_nt(a, b, out, bias=bias, b_packed=bp))
        us1 = bench(lambda: ops.gemm_nt(a, b, out, bias=bias))
        pts.append((K, us, us1))
        print(f"N={N:5d} K={K:5d}: panel {us:7.1f} us ({2.0*M*N*K/us/1e6:6.0f} TF/s)   128x128 {us1:7.1f} us ({2.0*M*N*K/us1/1e6:6.0f} TF/s)")
    (k0, u0, v0), (k1, u1, v1) = pts[-2], pts[-1]
    sl, sl1 = (u1 - u0) / ((k1 - k0) / 64), (v1 - v0) / ((k1 - k0) / 64)
    print(f"   slope per K64 step: panel {sl*1e3:.0f} ns (intercept {u1 - sl*k1/64:.1f} us), 128x128 {sl1*1e3:.0f} ns (intercept {v1 - sl1*k1/64:.1f} us)")

# the same for the FairLoRA epilogues (c_fc forward: bias + rank-op + LoRA update + GELU, two outputs; dX(c_fc): rank-op)
R, G = 8, 3
for name, N, mode in (("c_fc fwd", 3072, "fwd"), ("dX(c_proj)", 3072, "dx"), ("c_proj fwd", 768, "pfwd"), ("dX(c_fc)", 768, "fdx")):
    pts = []
    for K in (768, 1536, 3072):
        g = torch.Generator("cuda").manual_seed(1)
        a = torch.randn(M, K, device="cuda", generator=g).to(dt)
        b = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        rk = torch.zeros(16, K, device="cuda", dtype=dt)
        ops.PackPlan([(torch.randn(K, R, device="cuda", generator=g) * 0.1, False, rk)], dt, "cuda").run()
        attr = torch.randint(0, G, (32,), device="cuda", dtype=torch.int32)
        S = torch.randn(G, R, device="cuda", generator=g)
        if mode in ("fwd", "pfwd"):
            ro = ops.RankOp(rk, S, attr, 197, 0.25, 0.7, t_out=torch.empty(M, R, device="cuda"), ts_out=torch.empty(M, R, device="cuda"))
            kw = dict(bias=torch.randn(N, device="cuda", generator=g), lw=torch.randn(R, N, device="cuda", generator=g), rankop=ro)
            if mode == "fwd":
                kw["gelu_out"] = torch.empty(M, N, device="cuda", dtype=dt)
            else:
                kw["res"] = torch.randn(M, N, device="cuda", generator=g).to(dt)
        else:
            ro = ops.RankOp(rk, S, attr, 197, 0.25, 0.7, ts_out=torch.empty(M, R, device="cuda"),
                            t_fwd=torch.randn(M, R, device="cuda", generator=g), ds_part=torch.empty(512, G, R, device="cuda"))
            kw = dict(lw=torch.randn(N, R, device="cuda", generator=g), lw_is_kr=True, rankop=ro)
            if mode == "dx":
                kw["dgelu_aux"] = torch.randn(M, N, device="cuda", generator=g).to(dt)
        bp = ops.pack_b(b)
        pts.append((K, bench(lambda: ops.gemm_nt(a, b, out, b_packed=bp, **kw))))
    (k0, u0), (k1, u1) = pts[-2], pts[-1]
    sl = (u1 - u0) / ((k1 - k0) / 64)
    print(f"{name:11s} N={N:5d}: " + "  ".join(f"K={k}: {u:6.1f} us" for k, u in pts) +
          f"   slope {sl*1e3:.0f} ns per K64 step, intercept {u1 - sl*k1/64:.1f} us")
