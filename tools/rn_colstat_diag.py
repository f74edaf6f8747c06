#!/usr/bin/env python3
"""Full-size RN50, bs 4: the step with BatchNorm statistics from the GEMM epilogue vs from BatchNorm's own pass --
per-layer batch mean / rstd and block outputs side by side (is a difference in the logits a defect or bf16 drift?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine_rn import create_engine


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


mcfg = C.rn50(rank=8, num_groups=2)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = synth.make_batch(mcfg, bs, seed=1234)
args = (batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
for dt in (torch.float32, torch.bfloat16):
    engs = []
    for mode in (False, True):
        eng = create_engine(mcfg, sd, dtype=dt, max_images=bs)
        eng.no_colstats = mode
        out = eng.forward_backward(*args)
        torch.cuda.synchronize()
        engs.append((eng, out["logits"].clone()))
    (e0, l0), (e1, l1) = engs
    print(dt, "logits epilogue vs own pass:", rel(l0, l1))
    for i, (b0, b1) in enumerate(zip(e0.bns, e1.bns)):
        print(f"  {b0.prefix:45s} C={b0.C:5d} mean {rel(b0.mean, b1.mean):.2e} rstd {rel(b0.rstd, b1.rstd):.2e}")
    for i, (k0, k1) in enumerate(zip(e0.blocks, e1.blocks)):
        print(f"  block {i:2d} out {rel(k0.out, k1.out):.2e}")
