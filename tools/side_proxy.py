#!/usr/bin/env python3
"""What the PLACEMENT of the text tower's work costs the vision chain.  The text tower's launches are replaced by spin
kernels (tools/proto/spin.hip: no memory traffic, a given grid for a given time), so the results of the step are wrong and
only its time means something:
  * the real step, and the step without a text tower (tools/step_ablate.py's two ends);
  * `nf` + `nb` launches of G blocks x T threads for `us` microseconds each (the shape a narrow streaming text GEMM would have);
  * ONE launch of G blocks per direction (the shape a persistent text-tower kernel would have).
A panel block needs a whole CU, and the 240-248-block panels leave 8-16 of the 256 idle: the question is whether side work
that fits on those is free."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from fairfedmed_amd import config as C, synth, ops
from fairfedmed_amd.engine import FairLoRAEngine

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "proto", "libspin.so"))
lib.spin_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
lib.spin_fat_launch.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2 + [ctypes.c_int]
mcfg = C.vit_b16(rank=8)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
b = synth.make_batch(mcfg, 32, seed=1234)
img, attr, label = b["img"].cuda(), b["attrs"].t()[0].contiguous().cuda(), b["label"].cuda()
sink = torch.zeros(16, dtype=torch.int32, device="cuda")


def run(eng, n=30):
    for _ in range(4):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_backward(img, attr, label); eng.sgd_step(1e-3, 0.9, 5e-4)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def spins(count, blocks, threads, lds, us, fat=0):
    """fat: VGPRs per wave the stand-in allocates (0 / False: a handful; True: 128)"""
    fat = 128 if fat is True else int(fat)
    fn = (lambda *a: lib.spin_fat_launch(*a, fat)) if fat else lib.spin_launch

    def go(*a, **k):
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(count):
            def one(st=st):
                rc = fn(blocks, threads, lds, us, sink.data_ptr(), st)
                assert rc == 0, rc
            one()
            ops.record_callable(one)
    return go


# ONE engine per process (a second engine's side stream can land on the main stream's hardware queue)
eng = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=32)
real_f, real_b = eng._text_forward, eng._text_backward
print("full step                          : %.3f ms" % run(eng), flush=True)


def variant(name, fwd, bwd):
    eng.step_plans.clear()
    eng._text_forward, eng._text_backward = fwd, bwd
    print("%-35s: %.3f ms" % (name, run(eng)), flush=True)


none = lambda *a, **k: None
NF, NB = 87, 110
for rep in range(1):
    variant("no text tower", none, none)
    for G, T, us in ((8, 512, 8), (8, 1024, 8), (16, 512, 8), (32, 512, 8), (128, 512, 8), (128, 512, 4)):
        variant("%d+%d spins of %d x %d, %d us" % (NF, NB, G, T, us), spins(NF, G, T, 0, us), spins(NB, G, T, 0, us))
    # ... with a text GEMM's footprint: 128 VGPRs per wave and 48 KiB of LDS per block (such a block shares a CU with nothing large)
    for G, T, us in ((16, 512, 8), (64, 512, 8), (128, 512, 8)):
        variant("fat spins of %d x %d, %d us" % (G, T, us), spins(NF, G, T, 49152, us, True), spins(NB, G, T, 49152, us, True))
    for vg in (32, 64, 96, 128, 168, 240):
        for T in (512, 256):
            variant("%d VGPRs, no LDS, 64 x %d, 8 us" % (vg, T), spins(NF, 64, T, 0, 8, vg), spins(NB, 64, T, 0, 8, vg))
    variant("128 VGPRs, no LDS, 64 x 512, 8 us", spins(NF, 64, 512, 0, 8, True), spins(NB, 64, 512, 0, 8, True))
    variant("few VGPRs, 48 KiB LDS, 64 x 512, 8 us", spins(NF, 64, 512, 49152, 8, False), spins(NB, 64, 512, 49152, 8, False))
    variant("few VGPRs, 16 KiB LDS, 64 x 512, 8 us", spins(NF, 64, 512, 16384, 8, False), spins(NB, 64, 512, 16384, 8, False))
    variant("128 VGPRs, no LDS, 64 x 256, 8 us", spins(NF, 64, 256, 0, 8, True), spins(NB, 64, 256, 0, 8, True))
    variant("one 8 x 512 spin, 700 + 1000 us", spins(1, 8, 512, 0, 700), spins(1, 8, 512, 0, 1000))
    variant("one 8 x 1024 spin, 700 + 1000 us", spins(1, 8, 1024, 65536, 700), spins(1, 8, 1024, 65536, 1000))
    variant("one 16 x 512 spin, 500 + 800 us", spins(1, 16, 512, 0, 500), spins(1, 16, 512, 0, 800))
    variant("full step", real_f, real_b)
