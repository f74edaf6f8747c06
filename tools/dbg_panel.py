import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops
M, N, K = 208 * 2, 384, 768
dt = torch.bfloat16
g = torch.Generator("cuda").manual_seed(1)
a = torch.randn(M, K, device="cuda", generator=g).to(dt)
b = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
out = torch.zeros(M, N, device="cuda", dtype=dt)
# force the panel path regardless of the cost model
os.environ["FFM_PANEL_FORCE"] = "0"
ops.gemm_nt(a, b, out, b_packed=ops.pack_b(b))
torch.cuda.synchronize()
ref = a.float() @ b.float().t()
err = (out.float() - ref).abs()
print("tiles_m", ops.gemm_tiles_m(M, N, K, 0, 0, dt, True), "max err", float(err.max()))
e = err[:208].reshape(13, 16, 24, 16).amax(dim=(1, 3))
torch.set_printoptions(linewidth=250, precision=2)
print((e > 0.05).int())
