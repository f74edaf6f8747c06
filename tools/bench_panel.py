#!/usr/bin/env python3
"""Panel GEMM (csrc/gemm_panel_impl.h) vs the 128x128 kernel on the eight GEMMs of a vision block (bs 32):
us / TFLOP/s per shape with the engine's epilogues."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

dt = torch.bfloat16
M, W, R, G, RPS = int(os.environ.get('FFM_BENCH_M', 6304)), 768, 8, 3, 197


def bench(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def case(name, N, K, mode):
    g = torch.Generator("cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    b = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    kw = {}
    if "b" in mode:
        kw["bias"] = torch.randn(N, device="cuda", generator=g)
    if "r" in mode:
        kw["res"] = torch.randn(M, N, device="cuda", generator=g).to(dt)
    if "g" in mode:
        kw["gelu_out"] = torch.empty(M, N, device="cuda", dtype=dt)
    if "d" in mode:
        kw["dgelu_aux"] = torch.randn(M, N, device="cuda", generator=g).to(dt)
    if "l" in mode:
        kr = "k" in mode
        P = torch.randn(K, R, device="cuda", generator=g) * 0.1
        rk = torch.zeros(16, K, device="cuda", dtype=dt)
        ops.PackPlan([(P, False, rk)], dt, "cuda").run()
        attr = torch.randint(0, G, ((M + RPS - 1) // RPS,), device="cuda", dtype=torch.int32)
        t, ts = torch.empty(M, R, device="cuda"), torch.empty(M, R, device="cuda")
        bwd = kr
        rows = max(ops.gemm_tiles_m(M, N, K, 0, 0, dt, False), 512)
        ro = ops.RankOp(rk, torch.randn(G, R, device="cuda", generator=g), attr, RPS, 0.25, 0.7, t_out=None if bwd else t,
                        ts_out=ts, t_fwd=torch.randn(M, R, device="cuda", generator=g) if bwd else None,
                        ds_part=torch.empty(rows, G, R, device="cuda") if bwd else None)
        kw.update(lw=torch.randn(N, R, device="cuda", generator=g) if kr else torch.randn(R, N, device="cuda", generator=g),
                  lw_is_kr=kr, rankop=ro)
    bp = ops.pack_b(b)
    u0 = bench(lambda: ops.gemm_nt(a, b, out, **kw))
    u1 = bench(lambda: ops.gemm_nt(a, b, out, b_packed=bp, **kw))
    fl = 2.0 * M * N * K
    print(f"{name:10s} N{N:5d} K{K:5d} [{mode:5s}] 128x128 {u0:6.1f} us {fl / u0 / 1e6:7.1f} TF/s | panel {u1:6.1f} us "
          f"{fl / u1 / 1e6:7.1f} TF/s", flush=True)
    return u0, u1


if __name__ == "__main__":
    tot = [0.0, 0.0]
    for name, N, K, mode in [("qkv fwd", 3 * W, W, "b"), ("out fwd", W, W, "br"), ("fc fwd", 4 * W, W, "blg"),
                             ("proj fwd", W, 4 * W, "blr"), ("proj dX", 4 * W, W, "lkd"), ("fc dX", W, 4 * W, "lk"),
                             ("out dX", W, W, ""), ("qkv dX", W, 3 * W, "")]:
        u = case(name, N, K, mode)
        tot[0] += u[0]
        tot[1] += u[1]
    print(f"sum per block: 128x128 {tot[0]:.1f} us, panel {tot[1]:.1f} us")
