#!/usr/bin/env python3
"""Micro-benchmark of the non-GEMM kernels at the ViT-B/16 bs-32 shapes (6304 rows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
es = 2 if dt == torch.bfloat16 else 4
M, W, R, G, B, L, H = 6304, 768, 8, 3, 32, 197, 12


def timeit(fn, iters=20):
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


def report(name, us, nbytes):
    print(f"{name:34s} {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s (algorithmic {nbytes / 1e6:.1f} MB)")


rn = lambda *s: torch.randn(*s, device="cuda")
x768, x3072 = rn(M, W).to(dt), rn(M, 4 * W).to(dt)
attr = torch.randint(0, G, (B,), device="cuda", dtype=torch.int32)
S = rn(G, R)
t, ts, tf = torch.empty(M, R, device="cuda"), torch.empty(M, R, device="cuda"), rn(M, R)
dsp = torch.empty(ops.lora_down_blocks(M, W, R, dt), G, R, device="cuda")
for K, x in ((W, x768), (4 * W, x3072)):
    A, Bm = rn(K, R), rn(R, K)
    report(f"lora_down K={K} [K,r]", timeit(lambda: ops.lora_down(x, A, False, S, attr, R, G, L, 0.25, 0.7, t, ts)), M * K * es)
    report(f"lora_down K={K} [r,K]+dS", timeit(lambda: ops.lora_down(x, Bm, True, S, attr, R, G, L, 0.25, 0.7, t, ts, tf, dsp)), M * K * es)
    ns = ops.lora_grad_splits(M)
    part = torch.empty(ns, K, R, device="cuda")
    out = torch.empty(K, R, device="cuda")
    report(f"lora_grad_partial K={K}", timeit(lambda: ops.lora_grad_partial(x, tf, R, part)), M * K * es)
    report(f"reduce_partials n={K * R} x{ns}", timeit(lambda: ops.reduce_partials(part, ns, K * R, out)), ns * K * R * 4)
g, b = rn(W), rn(W)
y = torch.empty_like(x768)
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
report("layernorm_fwd", timeit(lambda: ops.layernorm_fwd(x768, y, g, b, mean, rstd)), 2 * M * W * es)
report("layernorm_bwd(+res)", timeit(lambda: ops.layernorm_bwd(y, x768, g, mean, rstd, x768, y)), 4 * M * W * es)
qkv = rn(M, 3 * W).to(dt)
o, do, dqkv = torch.empty(M, W, device="cuda", dtype=dt), rn(M, W).to(dt), torch.empty(M, 3 * W, device="cuda", dtype=dt)
lse, delta = torch.empty(B, H, L, device="cuda"), torch.empty(B, H, L, device="cuda")
fl = 4.0 * B * H * L * L * 64
us = timeit(lambda: ops.attention_fwd(qkv, o, lse, B, L, H))
print(f"{'attention_fwd':34s} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
us = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, delta, dqkv, B, L, H))
print(f"{'attention_bwd (delta+dq+dkv)':34s} {us:8.1f} us  {2.5 * fl / us / 1e6:7.1f} TFLOP/s (5 products)")
