#!/usr/bin/env python3
"""Where does a FairLoRA panel-GEMM launch spend its time?  Runs the diagnostic twin of the library
(python -m fairfedmed_amd.build --stamps; FFM_LIB_PATH) whose panel kernel writes {shader clock, 100 MHz clock}
pairs at its phase boundaries, one set per block:

  0 entry | 1 prologue done (ring stages 0-2 landed, epilogue operands in LDS) | 2 main loop done |
  3 rank-r update done | 4 epilogue stores issued | 5 stores acknowledged
  (inside the prologue: 6 ring stages + weight fragments issued | 7 epilogue operands loaded and in LDS)

Prints, per shape: launch duration by HIP events, the spread of block entry times, and the median / max per-phase
time over the blocks (us, from the 100 MHz clock; cycles from the shader clock).
    FFM_LIB_PATH=fairfedmed_amd/csrc/libffm_hip_stamps.so python tools/panel_stamps.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FFM_LIB_PATH", os.path.join(ROOT, "fairfedmed_amd", "csrc", "libffm_hip_stamps.so"))
import numpy as np
import torch
from fairfedmed_amd import ops

dt = torch.bfloat16
M, W, R, G, RPS = 6304, 768, 8, 3, 197
NST = 12


def plain(a, b, out, bp, bias, res, st):
    """ffm_gemm_nt with a plain epilogue and the stamp buffer in the unused `ts` field (ops.gemm_nt would read a
    `ts` argument as the unfused LoRA epilogue)."""
    import ctypes as C
    from fairfedmed_amd import _lib as L
    flags = (L.EPI_BIAS if bias is not None else 0) | (L.EPI_RESIDUAL if res is not None else 0)
    Mm, Kk = a.shape
    args = L.GemmArgs(L.ptr(a), L.ptr(b), L.ptr(out), Mm, b.shape[0], Kk, a.stride(0), b.stride(0), out.stride(0), flags, 0,
                      L.ptr(bias), L.ptr(st), None, L.ptr(res), None, None, None, None, None, None, None, None, None,
                      0, 0, 0.0, 0.0, L.ptr(bp))
    L.check(L.load().ffm_gemm_nt(C.byref(args), L.BF16, L.stream_ptr()), "ffm_gemm_nt")


def case(name, N, K, mode):
    g = torch.Generator("cuda").manual_seed(1)
    sets = []
    for _ in range(4):                                            # cycle buffer sets like the layers of a step do
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
            lwsrc = torch.randn(N, R, device="cuda", generator=g) if kr else torch.randn(R, N, device="cuda", generator=g)
            wide = torch.zeros(N, 32, device="cuda", dtype=dt)
            ops.PackPlan([(lwsrc, not kr, torch.zeros(16, N, device="cuda", dtype=dt), wide)], dt, "cuda").run()
            attr = torch.randint(0, G, (32,), device="cuda", dtype=torch.int32)
            t, ts = torch.empty(M, R, device="cuda"), torch.empty(M, R, device="cuda")
            rows = max(ops.gemm_tiles_m(M, N, K, 0, 0, dt, False), 512)
            ro = ops.RankOp(rk, torch.randn(G, R, device="cuda", generator=g), attr, RPS, 0.25, 0.7,
                            t_out=None if kr else t, ts_out=ts,
                            t_fwd=torch.randn(M, R, device="cuda", generator=g) if kr else None,
                            ds_part=torch.empty(rows, G, R, device="cuda") if kr else None,
                            lw_wide=None if "nowide" in sys.argv else wide)
            kw.update(lw=lwsrc, lw_is_kr=kr, rankop=ro)
        if "n" in mode:                                             # LayerNorm folded in (FFM_EPI_LNIN), 6 partials per row
            part = torch.randn(6, M, 2, device="cuda", generator=g).abs() + 1.0
            part[:, :, 1] = part[:, :, 0] ** 2 * 3 + 5
            kw["ln_in"] = ops.LnIn(part.contiguous(), 6, torch.randn(N, device="cuda", generator=g), torch.empty(M, device="cuda"),
                                   torch.empty(M, device="cuda"), rk=torch.randn(32, device="cuda", generator=g) if "l" in mode else None)
        sets.append((a, b, out, ops.pack_b(b), kw))
    stamps = torch.zeros(1024 * NST * 2, device="cuda", dtype=torch.int64)
    nostamp = None
    evs = []
    for it in range(24):
        a, b, out, bp, kw = sets[it % 4]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st = stamps if it == 23 else nostamp
        if "l" in mode:
            ops.gemm_nt(a, b, out, b_packed=bp, ts=st, **kw)       # `ts` is ignored under FFM_EPI_RANKOP
        else:
            plain(a, b, out, bp, kw.get("bias"), kw.get("res"), st)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    us = np.median([a.elapsed_time(b) * 1e3 for a, b in evs[4:23]])
    s = stamps.cpu().numpy().reshape(1024, NST, 2)
    nb = int((s[:, 0, 1] != 0).sum())
    s = s[:nb].astype(np.float64)
    cyc, rt = s[:, :, 0], s[:, :, 1] / 100.0                       # us
    t0 = rt[:, 0].min()
    print(f"{name:9s} N{N:5d} K{K:5d} [{mode:5s}]  {us:6.1f} us/launch (events), {nb} blocks; block entry spread "
          f"{rt[:, 0].max() - t0:.2f} us; first entry -> last block done {rt[:, 5].max() - t0:.2f} us")
    names = ["prologue", "main loop", "rank-r update", "output epilogue", "store drain"]
    for i, n in enumerate(names):
        d, dc = rt[:, i + 1] - rt[:, i], cyc[:, i + 1] - cyc[:, i]
        print(f"    {n:16s} median {np.median(d):6.2f} us  max {d.max():6.2f} us   {np.median(dc):9.0f} cycles "
              f"({np.median(dc) / max(np.median(d), 1e-9) / 1e3:.2f} GHz)")
    for n, i, j in (("  pro: issue ring+B", 0, 6), ("  pro: epilogue operands", 6, 7), ("  pro: wait for all", 7, 1)):
        d = rt[:, j] - rt[:, i]
        print(f"    {n:24s} median {np.median(d):6.2f} us  max {d.max():6.2f} us")
    if "l" in mode:
        for n, i, j in (("  rank: ts in registers", 2, 8), ("  rank: barrier", 8, 9), ("  rank: dS sum, lw frags", 9, 10),
                        ("  rank: MFMAs", 10, 3)):
            d = rt[:, j] - rt[:, i]
            print(f"    {n:24s} median {np.median(d):6.2f} us  max {d.max():6.2f} us")
    end = rt[:, 5] - t0
    print(f"    block end time   median {np.median(end):6.2f} us  min {end.min():6.2f}  max {end.max():6.2f}")


if __name__ == "__main__":
    for name, N, K, mode in [("fc fwd+ln", 4 * W, W, "blgn"), ("fc fwd", 4 * W, W, "blg"), ("proj fwd", W, 4 * W, "blr"), ("proj dX", 4 * W, W, "lkd"),
                             ("fc dX", W, 4 * W, "lk"), ("out fwd", W, W, "br"), ("qkv dX", W, 3 * W, "")]:
        case(name, N, K, mode)
