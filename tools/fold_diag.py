#!/usr/bin/env python3
"""LayerNorm folding on/off (engine.no_ln_fold): buffer-by-buffer comparison of one ViT-B/16 step at bs 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

mcfg, bs = C.vit_b16(rank=8), 32
sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
batch = synth.make_batch(mcfg, bs, seed=1234, signal=0.3)
eng = {}
for fold in (False, True):
    e = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=bs)
    e.no_ln_fold = not fold
    e.use_replay = False
    out = e.forward_backward(batch["img"].cuda(), batch["attrs"].t()[0].contiguous().cuda(), batch["label"].cuda())
    torch.cuda.synchronize()
    print("fold", fold, "loss", float(out["loss"]))
    eng[fold] = e
a, b = eng[False].vis, eng[True].vis
rows = bs * mcfg.vision.tokens


def cmp(name, x, y):
    x, y = x.double().flatten(), y.double().flatten()
    print(f"{name:12s} |ref| {float(x.norm()):.4e} rel err {float((x - y).norm() / x.norm()):.3e}")


for li in (0, 1, 11):
    for nm in ("x", "qkv", "xm", "pre", "act", "t1", "ts1", "t2"):
        cmp(f"L{li}.{nm}", getattr(a, nm)[li][:rows], getattr(b, nm)[li][:rows])
    cmp(f"L{li}.mean2", a.st2[li][0][:rows], b.st2[li][0][:rows])
    cmp(f"L{li}.rstd2", a.st2[li][1][:rows], b.st2[li][1][:rows])
keys = synth.trainable_keys(mcfg)
for k in keys:
    if "resblocks.0." in k or "resblocks.11." in k:
        cmp(k[-40:], eng[False].params.view(k, "grad"), eng[True].params.view(k, "grad"))
