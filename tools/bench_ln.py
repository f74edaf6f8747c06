#!/usr/bin/env python3
"""LayerNorm backward at the bench shape (6304 x 768 bf16, with residual): us per launch over 12 buffer sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

M, W = 6304, 768
dt = torch.bfloat16
sets = []
for i in range(12):
    g = torch.Generator("cuda").manual_seed(i)
    mk = lambda: torch.randn(M, W, device="cuda", generator=g).to(dt)
    sets.append((mk(), mk(), mk(), torch.empty(M, W, device="cuda", dtype=dt)))
gamma = torch.ones(W, device="cuda")
mean, rstd = torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")


def run():
    for dy, x, res, out in sets:
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, res, out)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print(f"layernorm_bwd {M}x{W} bf16: {e0.elapsed_time(e1) * 1e3 / 240:.2f} us per launch "
      f"({4 * M * W * 2 / (e0.elapsed_time(e1) * 1e-3 / 240) / 1e12:.2f} TB/s)")
