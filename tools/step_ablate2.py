#!/usr/bin/env python3
"""Which side-stream work costs the vision chain what: step time without the text tower's forward / backward / both,
without the LoRA-gradient reductions, with neither (results are then wrong; timing only)."""
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


def make(no_tf=False, no_tb=False):
    eng = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=32)
    eng.forward_backward(img, attr, label)                      # one real step fills the text feature buffers
    torch.cuda.synchronize()
    eng.step_plans.clear()
    if no_tf:
        eng._text_forward = lambda *a, **k: None
    if no_tb:
        eng._text_backward = lambda *a, **k: None
    return eng


print("full step                 : %.3f ms" % run(make()))
print("no text backward          : %.3f ms" % run(make(no_tb=True)))
print("no text forward + backward: %.3f ms" % run(make(True, True)))
real = ops.lora_grad_partial, ops.lora_grad_partial_ln
ops.lora_grad_partial = lambda *a, **k: None
ops.lora_grad_partial_ln = lambda *a, **k: None
print("no LoRA-grad partials     : %.3f ms" % run(make()))
print("neither                   : %.3f ms" % run(make(True, True)))
ops.lora_grad_partial, ops.lora_grad_partial_ln = real

# the LoRA-gradient kernels replaced by an LDS-free read of the same bytes (column sums by a plain torch reduction):
# is it their HBM traffic or their LDS footprint (5 x 32 KB per CU) that slows the vision chain down?
bufs = {}
def fake(x, v, r, part, *a, **k):
    key = x.shape[1]
    if key not in bufs:
        bufs[key] = torch.empty(key, device=x.device, dtype=torch.float32)
    torch.sum(x, dim=0, dtype=torch.float32, out=bufs[key])
ops.lora_grad_partial = fake
ops.lora_grad_partial_ln = lambda x, v, mean, rstd, gamma, beta, r, part: fake(x, v, r, part)
eng = make()
eng.use_replay = False
print("LoRA-grad partials as LDS-free streaming reads (eager launch path): %.3f ms" % run(eng))
ops.lora_grad_partial, ops.lora_grad_partial_ln = real
eng = make()
eng.use_replay = False
print("full step, eager launch path: %.3f ms" % run(eng))
