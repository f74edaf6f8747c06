"""Reference point: torch.matmul (hipBLASLt / rocBLAS) on the vision tower's GEMM shapes, next to ffm_gemm_nt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

M = 6304
shapes = [("qkv", 2304, 768), ("out", 768, 768), ("fc", 3072, 768), ("proj", 768, 3072), ("dqkv", 768, 2304)]


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    bias = torch.randn(N, device="cuda", dtype=torch.float32)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    wp = ops.pack_b(w)
    t_lib = timeit(lambda: torch.matmul(a, w.t(), out=out))
    t_lin = timeit(lambda: torch.nn.functional.linear(a, w, bias.to(torch.bfloat16)))
    t_v1 = timeit(lambda: ops.gemm_nt(a, w, out, bias=bias))
    t_pk = timeit(lambda: ops.gemm_nt(a, w, out, bias=bias, b_packed=wp))
    gf = 2.0 * M * N * K / 1e6
    print(f"{name:5s} N={N:5d} K={K:5d}: matmul {t_lib:6.1f} us ({gf / t_lib:5.0f} TF/s)  linear+bias {t_lin:6.1f} us  "
          f"ffm 128x128 {t_v1:6.1f} us ({gf / t_v1:5.0f})  ffm packed {t_pk:6.1f} us ({gf / t_pk:5.0f})")
