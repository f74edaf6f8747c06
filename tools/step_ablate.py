#!/usr/bin/env python3
"""What the side-stream work costs the vision chain: step time with the text tower / the LoRA-gradient reduction launches
removed (results are then wrong; timing experiment only).  ONE engine per process (a second engine's side
stream can land on the main stream's hardware queue); run it under FFM_LGRAD=0 FFM_SKINNY_NT=1 for the round-3 arrangement."""
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
real = (eng._text_forward, eng._text_backward, ops.lora_grad_partial, ops.lora_grad_partial_ln)
none = lambda *a, **k: None


def variant(name, text, red):
    eng.step_plans.clear()
    eng._text_forward, eng._text_backward = (real[0], real[1]) if text else (none, none)
    ops.lora_grad_partial, ops.lora_grad_partial_ln = (real[2], real[3]) if red else (none, none)
    print("%-28s: %.3f ms" % (name, run(eng)), flush=True)


for rep in range(2):
    variant("full step", True, True)
    variant("no text tower", False, True)
    variant("no LoRA-grad launches", True, False)
    variant("neither", False, False)
ops.lora_grad_partial, ops.lora_grad_partial_ln = real[2], real[3]
