#!/usr/bin/env python3
"""Would two half-batch chains on two streams beat one full-batch chain?  Every vision kernel is ONE round of blocks whose
phases (prologue, MFMA loop, HBM-bound epilogue) run in lockstep over the chip; two chains of half the rows put two kernels
of ~125 blocks on the 256 CUs at once, and their phases can interleave.  The probe replays the launches of LAYERS vision
blocks (the eight GEMMs with the engine's epilogues, attention forward + backward, two LayerNorm backwards; per-layer
buffers) as recorded launch plans:
   one chain of 32 images on one stream | two chains of 16 images, interleaved, on two streams | the same two on ONE stream
No text tower, no reductions: kernel time only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fairfedmed_amd import ops

dt = torch.bfloat16
W, R, G, L, H = 768, 8, 3, 197, 12
LAYERS = int(os.environ.get("DUAL_LAYERS", "6"))
gen = torch.Generator("cuda").manual_seed(1)
rn = lambda *s, scale=1.0: (torch.randn(*s, device="cuda", generator=gen) * scale)


class Weights:
    def __init__(self):
        self.w, self.bp = {}, {}
        for name, N, K in (("in", 3 * W, W), ("out", W, W), ("fc", 4 * W, W), ("proj", W, 4 * W), ("proj_t", 4 * W, W),
                           ("fc_t", W, 4 * W), ("out_t", W, W), ("in_t", W, 3 * W)):
            self.w[name] = rn(N, K, scale=K ** -0.5).to(dt)
            self.bp[name] = ops.pack_b(self.w[name])
        self.bias = {n: rn(self.w[n].shape[0]) for n in ("in", "out", "fc", "proj")}
        self.S = rn(G, R)
        self.rk, self.lw = {}, {}
        for name, K, N, kr in (("fc", W, 4 * W, False), ("proj", 4 * W, W, False), ("proj_t", W, 4 * W, True), ("fc_t", 4 * W, W, True)):
            rk = torch.zeros(16, K, device="cuda", dtype=dt)
            ops.PackPlan([(rn(K, R, scale=0.1), False, rk)], dt, "cuda").run()
            self.rk[name] = rk
            self.lw[name] = rn(N, R) if kr else rn(R, N)
        self.ln = (torch.ones(W, device="cuda"), torch.zeros(W, device="cuda"))


def chain(wt: Weights, images: int):
    """-> a function that launches LAYERS blocks (forward, then backward) on the current stream."""
    M = images * L
    e = lambda *s: torch.empty(*s, device="cuda", dtype=dt)
    f = lambda *s: torch.empty(*s, device="cuda")
    attr = torch.randint(0, G, (images,), device="cuda", dtype=torch.int32)
    nds = max(ops.gemm_tiles_m(M, 4 * W, W, 2 | 4 | 32 | 64, R, dt, True), ops.gemm_tiles_m(M, W, 4 * W, 2 | 4 | 64, R, dt, True), 64)
    layers = []
    for _ in range(LAYERS):
        b = dict(x=rn(M, W).to(dt), qkv=e(M, 3 * W), o=e(M, W), xm=e(M, W), pre=e(M, 4 * W), act=e(M, 4 * W), y=e(M, W),
                 lse=f(images * H * L), t1=f(M, R), ts1=f(M, R), t2=f(M, R), ts2=f(M, R), g=rn(M, W).to(dt), dpre=e(M, 4 * W),
                 dh=e(M, W), g1=e(M, W), do=e(M, W), dqkv=e(M, 3 * W), delta=f(images * H * L), us1=f(M, R), us2=f(M, R),
                 ds=f(nds, G, R), mean=torch.zeros(M, device="cuda"), rstd=torch.ones(M, device="cuda"), gout=e(M, W))
        layers.append(b)

    def run():
        for b in layers:
            ops.gemm_nt(b["x"], wt.w["in"], b["qkv"], bias=wt.bias["in"], b_packed=wt.bp["in"])
            ops.attention_fwd(b["qkv"], b["o"], b["lse"], images, L, H, False)
            ops.gemm_nt(b["o"], wt.w["out"], b["xm"], bias=wt.bias["out"], res=b["x"], b_packed=wt.bp["out"])
            ro = ops.RankOp(wt.rk["fc"], wt.S, attr, L, 0.25, 0.7, t_out=b["t1"], ts_out=b["ts1"])
            ops.gemm_nt(b["xm"], wt.w["fc"], b["pre"], bias=wt.bias["fc"], lw=wt.lw["fc"], gelu_out=b["act"], rankop=ro,
                        b_packed=wt.bp["fc"])
            ro = ops.RankOp(wt.rk["proj"], wt.S, attr, L, 0.25, 0.7, t_out=b["t2"], ts_out=b["ts2"])
            ops.gemm_nt(b["act"], wt.w["proj"], b["y"], bias=wt.bias["proj"], lw=wt.lw["proj"], res=b["xm"], rankop=ro,
                        b_packed=wt.bp["proj"])
        for b in reversed(layers):
            ro = ops.RankOp(wt.rk["proj_t"], wt.S, attr, L, 0.25, 0.7, ts_out=b["us2"], t_fwd=b["t2"], ds_part=b["ds"])
            ops.gemm_nt(b["g"], wt.w["proj_t"], b["dpre"], lw=wt.lw["proj_t"], lw_is_kr=True, dgelu_aux=b["pre"], rankop=ro,
                        b_packed=wt.bp["proj_t"])
            ro = ops.RankOp(wt.rk["fc_t"], wt.S, attr, L, 0.25, 0.7, ts_out=b["us1"], t_fwd=b["t1"], ds_part=b["ds"])
            ops.gemm_nt(b["dpre"], wt.w["fc_t"], b["dh"], lw=wt.lw["fc_t"], lw_is_kr=True, rankop=ro, b_packed=wt.bp["fc_t"])
            ops.layernorm_bwd(b["dh"], b["xm"], wt.ln[0], b["mean"], b["rstd"], b["g"], b["g1"])
            ops.gemm_nt(b["g1"], wt.w["out_t"], b["do"], b_packed=wt.bp["out_t"])
            ops.attention_bwd(b["qkv"], b["o"], b["do"], b["lse"], b["delta"], b["dqkv"], images, L, H, False)
            ops.gemm_nt(b["dqkv"], wt.w["in_t"], b["dh"], b_packed=wt.bp["in_t"])
            ops.layernorm_bwd(b["dh"], b["x"], wt.ln[0], b["mean"], b["rstd"], b["g1"], b["gout"])
    return run


def record(fn, stream):
    plan = []
    with torch.cuda.stream(stream), ops.record(plan):
        fn()
    return plan


def timed(plans, n=8):
    """plans: launch lists replayed interleaved (round-robin, one launch each)."""
    order = []
    for i in range(max(len(p) for p in plans)):
        for p in plans:
            if i < len(p):
                order.append(p[i])
    for _ in range(2):
        for f in order:
            f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in order:
            f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, len(order)


wt = Weights()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
full = record(chain(wt, 32), s1)
a1, b2 = record(chain(wt, 16), s1), record(chain(wt, 16), s2)
a1s, b1s = record(chain(wt, 16), s1), record(chain(wt, 16), s1)
torch.cuda.synchronize()
for rep in range(2):
    t, n = timed([full])
    print("one chain of 32 images, one stream          : %.3f ms for %d layers (%d launches)" % (t, LAYERS, n), flush=True)
    t, n = timed([a1, b2])
    print("two chains of 16 images, two streams         : %.3f ms (%d launches)" % (t, n), flush=True)
    t, n = timed([a1s, b1s])
    print("two chains of 16 images, ONE stream          : %.3f ms (%d launches)" % (t, n), flush=True)
