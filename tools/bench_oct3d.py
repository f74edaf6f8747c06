#!/usr/bin/env python3
"""BASELINE.json configs[3] at full size: 3D OCT, B = 4 volumes of 200 B-scans, D = 8 -> S = 25 slice groups
-> 100 ViT-B/16 images per step, FairLoRA r = 16, G = 3.  Prints ms/step and checks the step against the
exact-f32 engine on the same batch (loss, conv-weight gradient)."""
import dataclasses
import json
import os
import sys
import time
JSON = "--json" in sys.argv

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

mcfg = dataclasses.replace(C.vit_b16(rank=16), dim_per_3d_slice=8)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
B, S = 4, 25
batch = synth.make_batch(mcfg, B, seed=3, slices=S, signal=0.2)
img, attr, label = batch["img"].cuda(), batch["attrs"].t()[0].contiguous().cuda(), batch["label"].cuda()
eng = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=B * S)
if os.environ.get("FFM_SERIAL"):                       # one stream, one kernel at a time: clean per-kernel durations
    eng.set_overlap(False)
if not JSON:
    print("input", tuple(img.shape), "->", B * S, "ViT images")
    ref = FairLoRAEngine(mcfg, sd, dtype=torch.float32, max_images=B * S)
    o = eng.forward_backward(img, attr, label)
    r = ref.forward_backward(img, attr, label)
    torch.cuda.synchronize()
    gw, rw = eng.params.view("proj_per_3d_slice.weight", "grad"), ref.params.view("proj_per_3d_slice.weight", "grad")
    cosw = float(torch.dot(gw.flatten().double(), rw.flatten().double()) / (gw.double().norm() * rw.double().norm()))
    print("loss bf16 %.6f  f32 %.6f  finite %d  conv dW cosine %.5f" % (float(o["loss"]), float(r["loss"]), int(o["finite"]), cosw))
    del ref
for _ in range(3):
    eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n):
    eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
if JSON:
    print(json.dumps({"workload": "3D OCT: 4 volumes of 200x224x224, D=8 -> 100 ViT-B/16 images, FairLoRA r=16 G=3, "
                                  "fwd+bwd (through the trainable slice conv)+SGD", "ms_per_step": dt * 1e3,
                      "volumes_per_sec": B / dt, "vit_images_per_sec": B * S / dt, "steps": n, "dtype": "bfloat16",
                      "trainable_elems": eng.params.numel, "final_loss": float(eng.loss)}))
    sys.exit(0)
print("3D OCT step: %.2f ms  (%.1f volumes/s, %.0f ViT images/s)" % (dt * 1e3, B / dt, B * S / dt))
