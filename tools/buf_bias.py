#!/usr/bin/env python3
"""bf16 engine vs fp32 engine, buffer by buffer (vit_tiny): where does a gradient's norm ratio leave 1?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

mcfg, bs = C.vit_tiny(rank=4), 8
sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
batch = synth.make_batch(mcfg, bs, seed=1234, signal=0.45)
eng = {}
for dt in (torch.float32, torch.bfloat16):
    e = FairLoRAEngine(mcfg, sd, dtype=dt, max_images=bs)
    e.forward_backward(batch["img"].cuda(), batch["attrs"].t()[0].contiguous().cuda(), batch["label"].cuda())
    torch.cuda.synchronize()
    eng[dt] = e
a, b = eng[torch.float32].vis, eng[torch.bfloat16].vis
rows = bs * mcfg.vision.tokens


def cmp(name, x, y):
    x, y = x.double().flatten(), y.double().flatten()
    print(f"{name:14s} |f32| {float(x.norm()):.5e}  ratio {float(y.norm() / x.norm()):.4f}  cos {float(torch.dot(x, y) / (x.norm() * y.norm())):.6f}"
          f"  rel err {float((x - y).norm() / x.norm()):.2e}")


for li in range(mcfg.vision.layers):
    for nm in ("x", "xm", "qkv", "o", "h2", "pre", "act", "t1", "ts1", "t2", "ts2", "g_l", "dpre_l", "us2", "us1"):
        cmp(f"L{li}.{nm}", getattr(a, nm)[li][:rows], getattr(b, nm)[li][:rows])
cmp("feat", eng[torch.float32].feat[:rows], eng[torch.bfloat16].feat[:rows])
cmp("dfeat", eng[torch.float32].dfeat[:rows], eng[torch.bfloat16].dfeat[:rows])

# is the us2 deviation inherited from g (u = g B^T recomputed in float64 from each engine's own g) or made by the kernel?
for li in range(mcfg.vision.layers):
    Bm = sd[f"image_encoder.transformer.resblocks.{li}.mlp.c_proj.lora_B.weight"].double()          # [r, w]
    u32 = a.g_l[li][:rows].double().cpu() @ Bm.t()
    u16 = b.g_l[li][:rows].double().cpu() @ Bm.t()
    cmp(f"L{li}.u (host)", u32, u16)
    # per row us2 = scaling * u * s_b: compare the kernels' us2 / u elementwise between the engines
    k32 = a.us2[li][:rows].double().cpu() / u32
    k16 = b.us2[li][:rows].double().cpu() / u16
    print(f"      us2/u  f32 engine: median {float(k32.median()):.5f}; bf16 engine: median {float(k16.median()):.5f};"
          f" |u| f32 {float(u32.norm()):.4e} vs |g||B| {float(a.g_l[li][:rows].double().norm() * Bm.norm()):.4e}")
