#!/usr/bin/env python3
"""Is the bf16 engine's gradient BIASED (norm ratio != 1) or just noisy against the fp32 engine?  Same weights, same
batch: per trainable tensor |g_bf16| / |g_f32| and the cosine, plus the same for the logits' class difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

for name, mcfg, bs in (("vit_tiny", C.vit_tiny(rank=4), 8), ("vit_b16", C.vit_b16(rank=8), 8)):
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, bs, seed=1234, signal=0.45)
    keys = synth.trainable_keys(mcfg)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        eng = FairLoRAEngine(mcfg, sd, dtype=dt, max_images=bs)
        out = eng.forward_backward(batch["img"].cuda(), batch["attrs"].t()[0].contiguous().cuda(), batch["label"].cuda())
        torch.cuda.synchronize()
        res[dt] = ({k: eng.params.view(k, "grad").double().cpu().clone() for k in keys}, out["logits"].double().cpu().clone(),
                   float(out["loss"]))
        del eng
    g32, l32, loss32 = res[torch.float32]
    g16, l16, loss16 = res[torch.bfloat16]
    print(f"== {name}: loss f32 {loss32:.6f} bf16 {loss16:.6f}")
    d32, d16 = l32[:, 1] - l32[:, 0], l16[:, 1] - l16[:, 0]
    print("   logit difference l1-l0: f32", [round(float(v), 4) for v in d32[:8]], "\n" + " " * 27 + "bf16", [round(float(v), 4) for v in d16[:8]])
    ratios = []
    for k in keys:
        a, b = g16[k].flatten(), g32[k].flatten()
        if float(b.norm()) == 0:
            continue
        ratio, cs = float(a.norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))
        ratios.append(ratio)
        if name == "vit_tiny" or "resblocks.0." in k or "resblocks.11." in k or "ctx" in k:
            print(f"   {k[-60:]:60s} norm ratio {ratio:.4f}  cos {cs:.5f}  proj {float(torch.dot(a, b) / torch.dot(b, b)):.4f}")
    r = torch.tensor(ratios)
    print(f"   norm ratio over {len(ratios)} tensors: mean {float(r.mean()):.4f} min {float(r.min()):.4f} max {float(r.max()):.4f}")
