import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops
torch.manual_seed(0)
dt = torch.bfloat16
M, N, K, r, G, rps = 6304, 3072, 768, 8, 3, 197
a = torch.randn(M, K, device="cuda").to(dt); b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
P = torch.randn(K, r, device="cuda") * 0.1; S = torch.randn(G, r, device="cuda")
bias = torch.randn(N, device="cuda")
attr = torch.randint(0, G, ((M + rps - 1) // rps,), device="cuda", dtype=torch.int32)
rk = torch.zeros(16, K, device="cuda", dtype=dt); ops.PackPlan([(P, False, rk)], dt, "cuda").run()
for kr in (False, True):
    for use_bias in (True, False):
        for gelu in (True, False):
            lw = torch.randn(N, r, device="cuda") if kr else torch.randn(r, N, device="cuda")
            out = torch.empty(M, N, device="cuda", dtype=dt); act = torch.empty_like(out)
            t = torch.empty(M, r, device="cuda"); ts = torch.empty(M, r, device="cuda")
            ro = ops.RankOp(rk, S, attr, rps, 0.25, 0.7, t_out=t, ts_out=ts)
            kw = {}
            if use_bias: kw["bias"] = bias
            if gelu: kw["gelu_out"] = act
            res = torch.randn(M, N, device="cuda").to(dt)
            if not gelu and use_bias: kw["res"] = res
            if not gelu and not use_bias and kr: kw["dgelu_aux"] = res
            try:
                ops.gemm_nt(a, b, out, lw=lw, lw_is_kr=kr, rankop=ro, b_packed=ops.pack_b(b), **kw)
            except Exception as e:
                print(kr, use_bias, gelu, "ERR", str(e)[:60]); continue
            torch.cuda.synchronize()
            ref = a.float() @ b.float().t() + ts @ (lw.t() if kr else lw)
            if use_bias: ref += bias
            if "res" in kw: ref += res.float()
            if "dgelu_aux" in kw:
                x = res.float(); sg = torch.sigmoid(1.702 * x); ref = ref * (sg * (1 + 1.702 * x * (1 - sg)))
            err = (out.float() - ref).abs()
            e = err[:208].reshape(13, 16, 192, 16).amax(dim=(1, 3))
            bad = (e > 0.1 * ref.abs().max())
            print("kr", kr, "bias", use_bias, "gelu", gelu, "maxerr %.3f" % float(err.max() / ref.abs().max()),
                  "bad frags (rows of 13):", bad.sum(1).tolist(), " bad cols first tile:", bad[:, :24].sum(0).tolist())
