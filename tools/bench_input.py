#!/usr/bin/env python3
"""PCIe-inclusive step rate: pinned host batch -> H2D -> forward/backward/SGD, with the
reference's float32 transport (602 KB / image) and with the uint8 transport (50 KB / image, expanded on the GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

mcfg = C.vit_b16(rank=8)
eng = FairLoRAEngine(mcfg, synth.make_state_dict(mcfg, seed=1, lora_init="reference"), dtype=torch.bfloat16, max_images=32)
g = torch.Generator().manual_seed(0)
NB = 8
u8 = [torch.randint(0, 256, (32, 1, 224, 224), generator=g, dtype=torch.uint8).pin_memory() for _ in range(NB)]
f32 = [b.float().repeat_interleave(3, dim=1).contiguous().pin_memory() for b in u8]
attr = torch.randint(0, 3, (32,)).cuda()
label = torch.randint(0, 2, (32,)).cuda()
copy_stream = torch.cuda.Stream()


def run(batches, resident, n=40):
    dev = [b.cuda() for b in batches] if resident else None
    nxt = None
    for it in range(-4, n):
        if it == 0:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if resident:
            img = dev[it % NB]
        else:
            # double-buffered asynchronous copy on its own stream, as a DataLoader with pin_memory would feed it
            if nxt is None:
                with torch.cuda.stream(copy_stream):
                    nxt = batches[it % NB].cuda(non_blocking=True)
            torch.cuda.current_stream().wait_stream(copy_stream)
            img = nxt
            with torch.cuda.stream(copy_stream):
                nxt = batches[(it + 1) % NB].cuda(non_blocking=True)
        eng.forward_backward(img, attr, label)
        eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize()
    return 32 * n / (time.perf_counter() - t0)


print("float32 batch resident in HBM   : %7.0f img/s" % run(f32, True))
print("float32 over PCIe (19.3 MB/step): %7.0f img/s" % run(f32, False))
print("uint8   over PCIe ( 1.6 MB/step): %7.0f img/s" % run(u8, False))
print("uint8   batch resident in HBM   : %7.0f img/s" % run(u8, True))
