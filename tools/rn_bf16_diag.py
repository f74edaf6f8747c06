#!/usr/bin/env python3
"""bf16 RN tiny geometries: per-tensor gradient cosine against the fp32 oracle with the BatchNorm statistics taken from
the GEMM epilogue (default) and from BatchNorm's own pass (eng.no_colstats) -- how much of the spread is summation order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine_rn import create_engine
from oracle import fairlora_oracle as O


def cos(a, b):
    a, b = a.double().cpu().flatten(), torch.as_tensor(b).double().cpu().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-300))


for tag, geom in (("rn_tiny", C.rn_tiny), ("rn_tiny2", C.rn_tiny2)):
    mcfg = geom(rank=4, num_groups=2)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, 6, seed=1234)
    keys = synth.trainable_keys(mcfg)
    _, _, grads = O.loss_and_grads(dict(sd), batch, mcfg, keys)
    res = {}
    for mode in (False, True):
        eng = create_engine(mcfg, sd, dtype=torch.bfloat16, max_images=6)
        eng.no_colstats = mode
        eng.forward_backward(batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
        torch.cuda.synchronize()
        res[mode] = {k: eng.params.view(k, "grad").clone() for k in keys}
    for k in keys:
        if float(grads[k].abs().max()) == 0:
            continue
        c0, c1 = cos(res[False][k], grads[k]), cos(res[True][k], grads[k])
        if min(c0, c1) < 0.9:
            print(f"{tag} {k:60s} n={grads[k].numel():6d} epilogue {c0:.3f} own-pass {c1:.3f} between {cos(res[False][k], res[True][k]):.3f}")
    for suffix in ("lora_S.weight", "lora_A.weight", "lora_B.weight"):
        ks = [k for k in keys if k.endswith(suffix) and float(grads[k].abs().max()) > 0]
        cat = lambda d: torch.cat([torch.as_tensor(d[k]).double().cpu().flatten() for k in ks])
        print(tag, suffix, "concatenated:", f"epilogue {cos(cat(res[False]), cat(grads)):.4f} own-pass {cos(cat(res[True]), cat(grads)):.4f}")
