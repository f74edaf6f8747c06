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
