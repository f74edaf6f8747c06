#!/usr/bin/env python3
"""Step time with the product library vs the diagnostic build whose LoRA-gradient kernel streams the same bytes WITHOUT
LDS (FFM_LGM_NOLDS; wrong results): run once per library (FFM_LIB_PATH)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine
mcfg = C.vit_b16(rank=8)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
b = synth.make_batch(mcfg, 32, seed=1234)
img, attr, label = b["img"].cuda(), b["attrs"].t()[0].contiguous().cuda(), b["label"].cuda()
eng = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=32)
for rep in range(3):
    for _ in range(4):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize()
    print(os.path.basename(os.environ.get("FFM_LIB_PATH", "libffm_hip.so")), "%.3f ms/step" % ((time.perf_counter() - t0) / 30 * 1e3))
