#!/usr/bin/env python3
"""Would two half-batch chains side by side beat one full-batch chain?  (single-round kernels cannot
overlap their own prologue / epilogue / HBM tails; two independent chains on two streams could.)  Two engines of 16 images
each on streams of their own, stepped alternately, against one engine of 32: time per 32 images.  Timing probe only - the
two engines do not share gradients."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine

mcfg = C.vit_b16(rank=8)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")


def batch(n, seed):
    b = synth.make_batch(mcfg, n, seed=seed)
    return b["img"].cuda(), b["attrs"].t()[0].contiguous().cuda(), b["label"].cuda()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


mode = sys.argv[1] if len(sys.argv) > 1 else "dual"
if mode == "single":
    eng = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=32)
    x = batch(32, 1)
    def one():
        eng.forward_backward(*x); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
    print("one engine, 32 images            : %.3f ms per 32 images" % timeit(one))
else:
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ea = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=16)
    eb = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=16)
    xa, xb = batch(16, 1), batch(16, 2)
    torch.cuda.synchronize()
    def two():
        with torch.cuda.stream(sa):
            ea.forward_backward(*xa); ea.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
        with torch.cuda.stream(sb):
            eb.forward_backward(*xb); eb.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
    print("two engines of 16, two streams   : %.3f ms per 32 images" % timeit(two))
    def seq():
        ea.forward_backward(*xa); ea.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
        eb.forward_backward(*xb); eb.sgd_step(1e-3, 0.9, 5e-4, repeats=2)
    ea.step_plans.clear(); eb.step_plans.clear()
    print("two engines of 16, one after the other: %.3f ms per 32 images" % timeit(seq))
