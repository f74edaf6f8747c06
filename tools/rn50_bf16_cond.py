#!/usr/bin/env python3
"""How well conditioned is the full RN50 bf16 step as a parity fixture?  Per-tensor gradient cosine of the bf16 engine
against the fp32 oracle at batch 4 / 32, with the synthetic weights as they are and with the last BatchNorm of every
Bottleneck scaled down (gamma3 x 0.25: a trained ResNet's residual branches are small next to the identity path; the
random-weight trunk amplifies a perturbation ~1.5x per block)."""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine_rn import create_engine
from oracle import fairlora_oracle as O


def cos(a, b):
    a, b = a.double().cpu().flatten(), torch.as_tensor(b).double().cpu().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))


mcfg = C.rn50(rank=8, num_groups=2)
keys = synth.trainable_keys(mcfg)
for bs in (32,):
    for g3 in (0.1, 0.05):
        sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
        for k in sd:
            if k.endswith("bn3.weight"):
                sd[k] = sd[k] * g3
        batch = synth.make_batch(mcfg, bs, seed=1234)
        t0 = time.time()
        loss, logits, grads = O.loss_and_grads(copy.deepcopy(sd), batch, mcfg, keys)
        to = time.time() - t0
        for dtype in (torch.float32, torch.bfloat16):
            eng = create_engine(mcfg, sd, dtype=dtype, max_images=bs)
            out = eng.forward_backward(batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
            torch.cuda.synchronize()
            cs = sorted((cos(eng.params.view(k, "grad"), grads[k]), k) for k in keys if float(grads[k].abs().max()) > 0)
            lr = float((out["logits"].cpu() - logits).abs().max() / logits.abs().max())
            print(f"bs {bs:2d} gamma3 x{g3:4.2f} {str(dtype)[6:]:8s} oracle {to:5.1f}s loss {float(out['loss']):.5f} vs {float(loss):.5f} "
                  f"logits rel {lr:.2e}  cos min {cs[0][0]:.4f} ({cs[0][1][-40:]}) p5 {cs[len(cs) // 20][0]:.4f} "
                  f"median {cs[len(cs) // 2][0]:.5f}  n<0.95: {sum(1 for c, _ in cs if c < 0.95)}/{len(cs)}", flush=True)
            del eng
