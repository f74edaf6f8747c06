#!/usr/bin/env python3
"""RN50 step time by batch size and with / without side streams (what bounds the step: kernels or launches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine_rn import create_engine

mcfg = C.rn50(rank=8, num_groups=2)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")


def run(bs, overlap=True, steps=20, no_colstats=False):
    batch = synth.make_batch(mcfg, bs, seed=1234)
    args = (batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda())
    eng = create_engine(mcfg, sd, dtype=torch.bfloat16, max_images=bs)
    eng.no_colstats = no_colstats
    eng.set_overlap(overlap)
    for _ in range(3):
        eng.forward_backward(*args); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        eng.forward_backward(*args); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / steps * 1e3
    print(f"bs {bs:3d} overlap {overlap} no_colstats {no_colstats}: {ms:.2f} ms/step", flush=True)
    del eng
    torch.cuda.empty_cache()


for bs in (2, 8, 16, 32, 64):
    run(bs)
run(32, overlap=False)
run(32, no_colstats=True)
run(8, overlap=False)
