#!/usr/bin/env python3
"""Does stream priority keep the side streams (text tower, LoRA-gradient reductions) out of the vision chain's way?
The bench step on the default stream vs inside a high-priority stream (side streams stay at the default priority)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

mcfg, bs = C.vit_b16(rank=8), 32
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
batch = synth.make_batch(mcfg, bs, seed=1234)
img, attr, label = batch["img"].cuda(), batch["attrs"].t()[0].contiguous().cuda(), batch["label"].cuda()


def run(stream, steps=30):
    eng = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=bs)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(4):
            eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
for name, st in (("default stream", None), ("high-priority stream", torch.cuda.Stream(priority=-1)), ("default stream", None),
                 ("high-priority stream", torch.cuda.Stream(priority=-1))):
    print(f"{name:22s}: {run(st):.3f} ms/step")
