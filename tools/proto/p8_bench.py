#!/usr/bin/env python3
"""Prototype check + timing: 8-wave panel main loop (tools/proto/p8.hip) against ffm_gemm_nt on the N = 768 shapes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fairfedmed_amd import ops
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libp8.so"))
lib.p8_run.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]
M, N = 6304, 768
dt = torch.bfloat16
for K in (3072, 2304, 768):
    g = torch.Generator("cuda").manual_seed(K)
    a = torch.randn(M, K, device="cuda", generator=g).to(dt)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
    bp = ops.pack_b(w)
    ref = torch.empty(M, N, device="cuda", dtype=dt)
    ops.gemm_nt(a, w, ref, b_packed=bp)
    for mfw in (5, 6):
        out = torch.zeros(M, N, device="cuda", dtype=dt)
        st = torch.zeros(1024 * 4, device="cuda", dtype=torch.int64)
        assert lib.p8_run(a.data_ptr(), bp.data_ptr(), out.data_ptr(), M, N, K, mfw, st.data_ptr(), None) == 0
        torch.cuda.synchronize()
        err = float((out.float() - ref.float()).abs().max() / ref.float().abs().max())
        def bench(fn, it=40):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / it
        t8 = bench(lambda: lib.p8_run(a.data_ptr(), bp.data_ptr(), out.data_ptr(), M, N, K, mfw, None, None))
        tp = bench(lambda: ops.gemm_nt(a, w, ref, b_packed=bp))
        s = st.cpu().numpy().reshape(-1, 4).astype(np.float64)
        nb = int((s[:, 0] != 0).sum()); s = s[:nb] / 100.0
        print(f"K {K} MFW {mfw}: err {err:.1e}; 8-wave {t8:.1f} us (loop median {np.median(s[:, 2] - s[:, 1]):.1f}, prologue "
              f"{np.median(s[:, 1] - s[:, 0]):.1f}, {nb} blocks) | panel {tp:.1f} us")
