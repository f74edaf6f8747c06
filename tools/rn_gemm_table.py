#!/usr/bin/env python3
"""RN50 r=8 G=2 bs 32, one stream, eager launches: every ffm_gemm_nt / ffm_conv3x3_nhwc launch of a step with its shape,
epilogue, time, TFLOP/s and the GB/s of its operand + result bytes (which launches are far from both roofs)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth, ops
from fairfedmed_amd.engine_rn import create_engine

bs = 32
mcfg = C.rn50(rank=8, num_groups=2)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
batch = synth.make_batch(mcfg, bs, seed=1234)
args = (batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
eng = create_engine(mcfg, sd, dtype=torch.bfloat16, max_images=bs)
eng.use_replay = False
eng.set_overlap(False)
for _ in range(2):
    eng.forward_backward(*args)
rec = []
og, oc = ops.gemm_nt, ops.conv3x3


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def gemm(a, b, out, **kw):
    e0 = ev(); r = og(a, b, out, **kw); e1 = ev()
    M, K = a.shape; N = b.shape[0]
    tag = "".join(k[0] for k in ("bias", "res", "rankop", "ts", "colstats") if kw.get(k) is not None) + ("K" if kw.get("lw_is_kr") else "")
    by = (M * K + N * K + M * N * (2 if kw.get("res") is not None else 1)) * a.element_size()
    rec.append(("gemm", M, N, K, tag, e0, e1, 2.0 * M * N * K, by))
    return r


def conv(x, w, out, B, H, W, zeros, scratch=None, colstats=None):
    e0 = ev(); r = oc(x, w, out, B, H, W, zeros, scratch, colstats); e1 = ev()
    M, Cc = x.shape; N, Kp = w.shape
    rec.append(("conv3", M, N, 9 * Cc, "s" if scratch is not None else "", e0, e1, 2.0 * M * N * 9 * Cc, (M * Cc + N * Kp + M * N) * x.element_size()))
    return r


ops.gemm_nt, ops.conv3x3 = gemm, conv
STEPS = 5
for _ in range(STEPS):
    eng.forward_backward(*args)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, M, N, K, tag, e0, e1, fl, by in rec:
    if M < 1024:
        continue                                   # text tower
    k = (kind, M, N, K, tag)
    a = agg.setdefault(k, [0, 0.0, fl, by])
    a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = 0.0
print(f"{'kind':6s} {'M':>7s} {'N':>5s} {'K':>5s} {'epi':6s} {'n/step':>6s} {'us':>7s} {'TF/s':>7s} {'GB/s':>7s} {'us/step':>8s}")
for (kind, M, N, K, tag), (n, us, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    u = us / n
    tot += us / STEPS
    print(f"{kind:6s} {M:7d} {N:5d} {K:5d} {tag:6s} {n / STEPS:6.1f} {u:7.1f} {fl / u / 1e6:7.1f} {by / u / 1e3:7.0f} {us / STEPS:8.1f}")
print("total us/step", tot)
