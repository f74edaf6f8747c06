#!/usr/bin/env python3
"""Step time of the bench workload with the side streams on / folded into the main stream / the two side streams merged."""
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


def run(n=20):
    for _ in range(3):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("overlap on : %.3f ms/step" % run())
eng.set_overlap(False)
print("overlap off: %.3f ms/step" % run())
eng.set_overlap(True)
print("overlap on : %.3f ms/step" % run())
# the LoRA-gradient reductions on the TEXT tower's stream (two side kernels never run at once; both still beside the chain)
for rep in range(2):
    eng.grad_stream = eng.side
    eng.step_plans.clear()
    print("grad reductions on the text stream: %.3f ms/step" % run())
    eng.set_overlap(True)
    print("three streams                     : %.3f ms/step" % run())
