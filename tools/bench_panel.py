#!/usr/bin/env python3
"""Panel GEMM (csrc/gemm_panel.hip) vs the 128x128 kernel on the vision-tower shapes:
correctness against torch and us / TFLOP/s per shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops


def run(M, N, K, mode, packed, iters=30, check=True):
    dt = torch.bfloat16
    g = torch.Generator("cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    b = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    kw = {}
    if "b" in mode:
        kw["bias"] = torch.randn(N, device="cuda", generator=g)
    if "r" in mode:
        kw["res"] = torch.randn(M, N, device="cuda", generator=g).to(dt)
    if packed:
        kw["b_packed"] = ops.pack_b(b)
    err = None
    ops.gemm_nt(a, b, out, **kw)
    torch.cuda.synchronize()
    if check:
        ref = a.float() @ b.float().t()
        if "b" in mode:
            ref += kw["bias"]
        if "r" in mode:
            ref += kw["res"].float()
        err = float((out.float() - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        ops.gemm_nt(a, b, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm_nt(a, b, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us / 1e6, err


SHAPES = [("qkv fwd", 6304, 2304, 768, "b"), ("out fwd", 6304, 768, 768, "br"), ("fc plain", 6304, 3072, 768, "b"),
          ("proj plain", 6304, 768, 3072, "br"), ("do bwd", 6304, 768, 768, ""), ("dh1 bwd", 6304, 768, 2304, ""),
          ("odd M", 5000, 3072, 768, "b"), ("big", 8192, 6144, 4096, "")]

if __name__ == "__main__":
    for name, M, N, K, mode in SHAPES:
        tiles = ops.gemm_tiles_m(M, N, K, (1 if "b" in mode else 0) | (8 if "r" in mode else 0), 0, torch.bfloat16, True)
        u0, t0, e0 = run(M, N, K, mode, False)
        u1, t1, e1 = run(M, N, K, mode, True)
        print(f"{name:11s} M{M} N{N} K{K} [{mode:2s}] 128x128 {u0:7.1f} us {t0:7.1f} TF/s err {e0:.1e} | "
              f"panel({tiles} row tiles) {u1:7.1f} us {t1:7.1f} TF/s err {e1:.1e}", flush=True)
