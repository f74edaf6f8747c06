#!/usr/bin/env python3
"""Micro-benchmark of ffm_gemm_nt on the shapes of the ViT-B/16 FairLoRA step
(bs 32 -> 6304 token rows).  Random operands (zero operands read high: guide
rule 25).  Prints us / TFLOP/s per shape; --sweep adds a K sweep."""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops


def time_gemm(M, N, K, dt, mode, iters=30):
    a = torch.randn(M, K, device="cuda").to(dt)
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    kw = {}
    if "b" in mode:
        kw["bias"] = torch.randn(N, device="cuda")
    if "l" in mode:
        kw["ts"] = torch.randn(M, 8, device="cuda")
        kw["lw"] = torch.randn(8, N, device="cuda")
    if "r" in mode:
        kw["res"] = torch.randn(M, N, device="cuda").to(dt)
    if "g" in mode:
        kw["gelu_out"] = torch.empty(M, N, device="cuda", dtype=dt)
    if "d" in mode:
        kw["dgelu_aux"] = torch.randn(M, N, device="cuda").to(dt)
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
    return us, 2.0 * M * N * K / us / 1e6


SHAPES = [
    ("qkv fwd", 6304, 2304, 768, "b"), ("out fwd", 6304, 768, 768, "br"), ("fc fwd", 6304, 3072, 768, "blg"),
    ("proj fwd", 6304, 768, 3072, "blr"), ("final proj", 6304, 512, 768, ""), ("patch", 6272, 768, 768, ""),
    ("dact bwd", 6304, 3072, 768, "ld"), ("dh2 bwd", 6304, 768, 3072, "l"), ("do bwd", 6304, 768, 768, ""),
    ("dh1 bwd", 6304, 768, 2304, ""), ("dhpost bwd", 6304, 768, 512, ""),
]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--sweep", action="store_true")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    tot_us = tot_fl = 0.0
    for name, M, N, K, mode in SHAPES:
        us, tf = time_gemm(M, N, K, dt, mode)
        print(f"{name:12s} M{M} N{N} K{K} [{mode:4s}] {us:8.1f} us {tf:8.1f} TF/s")
        tot_us += us
        tot_fl += 2.0 * M * N * K
    print(f"sum {tot_us:.1f} us, aggregate {tot_fl / tot_us / 1e6:.1f} TF/s")
    if args.sweep:
        for K in (256, 768, 1536, 3072, 8192):
            us, tf = time_gemm(6304, 3072, K, dt, "")
            print(f"sweep N3072 K{K}: {us:8.1f} us {tf:8.1f} TF/s")
        for K in (768, 3072, 8192):
            us, tf = time_gemm(6304, 768, K, dt, "")
            print(f"sweep N768 K{K}: {us:8.1f} us {tf:8.1f} TF/s")
        us, tf = time_gemm(8192, 8192, 8192, dt, "", iters=5)
        print(f"8192^3: {us:8.1f} us {tf:8.1f} TF/s")
