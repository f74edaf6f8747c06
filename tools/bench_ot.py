#!/usr/bin/env python3
"""Step time of the bench workload (ViT-B/16 r=8, bs 32, bf16) with the Sinkhorn / COT logits heads instead of OT=None."""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # several engines in one process: keep their streams on separate queues
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

b = synth.make_batch(C.vit_b16(rank=8), 32, seed=1234)
img, attr, label = b["img"].cuda(), b["attrs"].t()[0].contiguous().cuda(), b["label"].cuda()
for ot in (sys.argv[1:] or ["None"]):                        # one head per process: python3 tools/bench_ot.py Sinkhorn
    mcfg = dataclasses.replace(C.vit_b16(rank=8), ot=ot, ot_top_percent=0.8)
    eng = FairLoRAEngine(mcfg, synth.make_state_dict(mcfg, seed=1, lora_init="reference"), dtype=torch.bfloat16, max_images=32)
    for _ in range(30):                                   # a fresh engine in the same process needs a long warm-up
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    extra = f", stopped after {int(eng.ot_istop) + 1} iterations" if ot != "None" else ""
    print(f"OT={ot:8s}: {ms:.2f} ms/step, {32 / ms * 1e3:.0f} img/s, loss {float(eng.loss):.4f}{extra}")
    del eng
    torch.cuda.empty_cache()
