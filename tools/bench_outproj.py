import os, sys
sys.path.insert(0, os.getcwd())
import torch
from fairfedmed_amd import ops, _lib as L
dt = torch.bfloat16
M, N, K = 6304, 768, 768
g = torch.Generator("cuda").manual_seed(1)
NS = 12
sets = []
for i in range(NS):
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    sets.append(dict(a=a, w=w, bp=ops.pack_b(w), res=torch.randn(M, N, device="cuda", generator=g).to(dt), out=torch.empty(M, N, device="cuda", dtype=dt),
                     bias=torch.randn(N, device="cuda", generator=g)))
tn = ops.gemm_tiles_n(M, N, K, L.EPI_BIAS | L.EPI_RESIDUAL | L.EPI_ROWSTATS, 0, dt, True)
rowp = torch.empty(tn * M * 2, device="cuda")
def timed(fn, iters=60):
    for i in range(6): fn(sets[i % NS])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): fn(sets[i % NS])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for r in range(2):
    print("plain (no epilogue)      %.1f us" % timed(lambda s: ops.gemm_nt(s["a"], s["w"], s["out"], b_packed=s["bp"])))
    print("bias                     %.1f us" % timed(lambda s: ops.gemm_nt(s["a"], s["w"], s["out"], bias=s["bias"], b_packed=s["bp"])))
    print("bias + residual          %.1f us" % timed(lambda s: ops.gemm_nt(s["a"], s["w"], s["out"], bias=s["bias"], res=s["res"], b_packed=s["bp"])))
    print("bias + residual + rowstat %.1f us" % timed(lambda s: ops.gemm_nt(s["a"], s["w"], s["out"], bias=s["bias"], res=s["res"], b_packed=s["bp"], rowstats=rowp)))
