#!/usr/bin/env python3
"""Step time of the bench workload (bs 32, bf16) on whatever library FFM_LIB_PATH names: one number, for A/B scripts.
STEP_TIME_LATE=1 moves the LoRA-gradient reductions of a block behind that block's attention backward (engine.late_reductions)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

mcfg = C.vit_b16(rank=8)
eng = FairLoRAEngine(mcfg, synth.make_state_dict(mcfg, seed=1, lora_init="reference"), dtype=torch.bfloat16, max_images=32)
b = synth.make_batch(mcfg, 32, seed=1234)
img, attr, label = b["img"].cuda(), b["attrs"].t()[0].contiguous().cuda(), b["label"].cuda()
out = []
for rep in range(int(os.environ.get("STEP_TIME_REPS", "2"))):
    for _ in range(4):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 40 * 1e3)
print(" ".join("%.3f" % v for v in out), "ms/step")
