#!/usr/bin/env python3
"""Why are in-step GEMM launches ~20 % slower than their micro-benchmark?  The c_fc forward launch (M 6304, N 3072,
K 768, FairLoRA + GELU epilogue, two 38.7 MB outputs) timed (a) on one set of buffers, (b) cycling through 12 sets
(what the 12 layers of a step do: inputs and outputs are never cache-resident), (c) as (b) with other kernels between."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

dt = torch.bfloat16
M, N, K, R, G = 6304, 3072, 768, 8, 3
g = torch.Generator("cuda").manual_seed(1)
NS = 12
sets = []
for i in range(NS):
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    b = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    P = torch.randn(K, R, device="cuda", generator=g) * 0.1
    rk = torch.zeros(16, K, device="cuda", dtype=dt)
    ops.PackPlan([(P, False, rk)], dt, "cuda").run()
    attr = torch.randint(0, G, (32,), device="cuda", dtype=torch.int32)
    ro = ops.RankOp(rk, torch.randn(G, R, device="cuda", generator=g), attr, 197, 0.25, 0.7,
                    t_out=torch.empty(M, R, device="cuda"), ts_out=torch.empty(M, R, device="cuda"))
    if "dx" in sys.argv:
        # dX(c_proj): reads the saved pre-activation (38.7 MB, cold in a real step) for the GELU derivative
        rows = max(ops.gemm_tiles_m(M, N, K, 0, 0, dt, False), 512)
        ro = ops.RankOp(rk, torch.randn(G, R, device="cuda", generator=g), attr, 197, 0.25, 0.7,
                        ts_out=torch.empty(M, R, device="cuda"), t_fwd=torch.randn(M, R, device="cuda", generator=g),
                        ds_part=torch.empty(rows, G, R, device="cuda"))
        kw = dict(dgelu_aux=torch.randn(M, N, device="cuda", generator=g).to(dt),
                  lw=torch.randn(N, R, device="cuda", generator=g), lw_is_kr=True, rankop=ro)
    else:
        kw = dict(bias=torch.randn(N, device="cuda", generator=g), gelu_out=torch.empty(M, N, device="cuda", dtype=dt),
                  lw=torch.randn(R, N, device="cuda", generator=g), rankop=ro)
    sets.append(dict(a=a, b=b, out=torch.empty(M, N, device="cuda", dtype=dt), bp=ops.pack_b(b), kw=kw))
filler_a = torch.randn(6304, 3072, device="cuda").to(dt)
filler_b = torch.empty_like(filler_a)


def run(s):
    ops.gemm_nt(s["a"], s["b"], s["out"], b_packed=s["bp"], **s["kw"])


def timed(pick, between=False, iters=48):
    evs = []
    for it in range(-6, iters):
        s = sets[pick(it)]
        if between:
            filler_b.copy_(filler_a)            # 77 MB of unrelated traffic, like the kernels between two GEMMs
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(s); e1.record()
        if it >= 0:
            evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e3


print("same buffers, back to back        : %.1f us" % timed(lambda i: 0))
print("12 buffer sets, back to back      : %.1f us" % timed(lambda i: i % NS))
print("12 buffer sets, traffic in between: %.1f us" % timed(lambda i: i % NS, True))
print("same buffers, traffic in between  : %.1f us" % timed(lambda i: 0, True))
