import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops
dt = torch.bfloat16
M, N, K = 6304, 3072, 768
a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
out = torch.empty(M, N, device="cuda", dtype=dt); bias = torch.randn(N, device="cuda")
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print("bias only", t(lambda: ops.gemm_nt(a, b, out, bias=bias)))
for r in (4, 8, 16, 24, 32):
    ts = torch.randn(M, r, device="cuda"); lw = torch.randn(r, N, device="cuda")
    print("r", r, "bias+lora", t(lambda: ops.gemm_nt(a, b, out, bias=bias, ts=ts, lw=lw)))
