import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops
dt = torch.bfloat16
def t(M, N, K, iters=50, check=False):
    a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt); bp = ops.pack_b(b)
    for _ in range(5): ops.gemm_nt(a, b, out, b_packed=bp)
    torch.cuda.synchronize()
    if check:
        ref = a.float() @ b.float().t()
        return float((out.float() - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_nt(a, b, out, b_packed=bp)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
print("err", t(6304, 2304, 768, check=True), t(6304, 768, 3072, check=True), t(1000, 768, 768, check=True), t(6304, 3072, 768, check=True))
print("tiles", ops.gemm_tiles_m(6304, 2304, 768, 0, 0, dt, True), ops.gemm_tiles_m(6304, 768, 768, 0, 0, dt, True), ops.gemm_tiles_m(6304, 3072, 768, 0, 0, dt, True))
for dbg in (0, 1, 2, 3, 7):
    os.environ["FFM_PANEL_DBG"] = str(dbg)
    print("dbg", dbg, " N3072(16,4) K768 %.1f K3072 %.1f | qkv(16,4) %.1f | proj(10,2) K3072 %.1f  K768 %.1f  K2304 %.1f" % (
        t(6304, 3072, 768), t(6304, 3072, 3072), t(6304, 2304, 768), t(6304, 768, 3072), t(6304, 768, 768), t(6304, 768, 2304)), flush=True)
