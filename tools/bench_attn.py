#!/usr/bin/env python3
"""Attention forward / backward at the bench shape (B 32, L 197, 12 heads, bf16): us per launch; run once per generation
(FFM_ATTN=v1 | v2 | v3 in the environment; default v3 = csrc/attention3.hip).  argv[1] = fp16 for the half-precision kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

B, L, H = 32, 197, 12
E = H * 64
dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "fp16" else torch.bfloat16
g = torch.Generator("cuda").manual_seed(1)
sets = []
NSETS = int(os.environ.get("ATTN_SETS", "6"))      # buffer sets cycled through (1: everything stays cache-resident)
for _ in range(NSETS):
    qkv = torch.randn(B * L, 3 * E, device="cuda", generator=g).to(dt)
    out = torch.empty(B * L, E, device="cuda", dtype=dt)
    lse = torch.empty(B, H, L, device="cuda")
    dout = torch.randn(B * L, E, device="cuda", generator=g).to(dt)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, L, device="cuda")
    sets.append((qkv, out, lse, dout, dqkv, delta))


def bench(fn, iters=60):
    for i in range(6):
        fn(sets[i % NSETS])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(sets[i % NSETS])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


f = bench(lambda s: ops.attention_fwd(s[0], s[1], s[2], B, L, H, False))
b = bench(lambda s: ops.attention_bwd(s[0], s[1], s[3], s[2], s[5], s[4], B, L, H, False))
gen = os.environ.get("FFM_ATTN", "v3 (default)")
print(f"FFM_ATTN={gen}: forward {f:.1f} us, backward {b:.1f} us per launch "
      f"(fwd {4 * B * H * L * L * 64 / f / 1e6:.0f} TF/s, bwd {10 * B * H * L * L * 64 / b / 1e6:.0f} TF/s)")
