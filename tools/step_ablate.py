#!/usr/bin/env python3
"""What the side-stream work costs the vision chain: step time with the text tower / the LoRA-gradient reductions
removed (results are then wrong; timing experiment only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fairfedmed_amd import config as C, synth, ops
from fairfedmed_amd.engine import FairLoRAEngine

mcfg = C.vit_b16(rank=8)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
b = synth.make_batch(mcfg, 32, seed=1234)
img, attr, label = b["img"].cuda(), b["attrs"].t()[0].contiguous().cuda(), b["label"].cuda()


def run(eng, n=30):
    for _ in range(4):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eng = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=32)
print("full step            : %.3f ms" % run(eng))

eng2 = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=32)
eng2.forward_backward(img, attr, label)                      # one real step fills tbar_buf etc.
eng2.step_plans.clear()
eng2._text_forward = lambda *a, **k: None
eng2._text_backward = lambda *a, **k: None
print("no text tower        : %.3f ms" % run(eng2))

eng3 = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=32)
real = ops.lora_grad_partial
ops.lora_grad_partial = lambda *a, **k: None
print("no LoRA-grad partials: %.3f ms" % run(eng3))
eng3.step_plans.clear()
eng3._text_forward = lambda *a, **k: None
eng3._text_backward = lambda *a, **k: None
eng3.tbar_buf.copy_(eng.tbar_buf)
print("neither              : %.3f ms" % run(eng3))
ops.lora_grad_partial = real
