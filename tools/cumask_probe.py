#!/usr/bin/env python3
"""Does confining the side streams to a few CUs (hipExtStreamCreateWithCUMask) take their cost off the main stream?
The panel GEMMs are one block per CU on 240-248 of the 256 CUs; a side kernel that lands on more than the 8-16 spare
CUs holds a panel block back.  usage: cumask_probe.py <text_cus> <grad_cus> [layout]   (0 = unmasked;
layout "low": the lowest n bits, "stride": every (256/n)-th bit)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth
from fairfedmed_amd.engine import FairLoRAEngine


def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 not loaded")


def masked_stream(n, layout):
    bits = [0] * 8
    idx = range(n) if layout == "low" else range(0, 256, 256 // n)
    for i in idx:
        bits[i // 32] |= 1 << (i % 32)
    arr = (ctypes.c_uint32 * 8)(*bits)
    s = ctypes.c_void_p()
    rc = hip_runtime().hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


ntext, ngrad = int(sys.argv[1]), int(sys.argv[2])
layout = sys.argv[3] if len(sys.argv) > 3 else "low"
torch.cuda.init(); torch.zeros(1, device="cuda")
mcfg = C.vit_b16(rank=8)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
eng = FairLoRAEngine(mcfg, sd, dtype=torch.bfloat16, max_images=32)
if ntext:
    eng.side = eng._side0 = masked_stream(ntext, layout)
if ngrad:
    eng.grad_stream = eng._grad0 = masked_stream(ngrad, layout)
prio = os.environ.get("PRIO", "")                 # stream priorities instead: "main" / "side" / "text" / "grad" run high
if prio in ("side", "text"):
    eng.side = eng._side0 = torch.cuda.Stream(priority=-1)
if prio in ("side", "grad"):
    eng.grad_stream = eng._grad0 = torch.cuda.Stream(priority=-1)
main = torch.cuda.Stream(priority=-1) if prio == "main" else torch.cuda.current_stream()
b = synth.make_batch(mcfg, 32, seed=1)
x = (b["img"].cuda(), b["attrs"].t()[0].contiguous().cuda(), b["label"].cuda())


def one():
    with torch.cuda.stream(main):
        eng.forward_backward(*x); eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)


for _ in range(8):
    one()
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        one()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
print("text CUs %3d  grad CUs %3d  layout %-6s prio %-5s: %.3f ms/step" % (ntext, ngrad, layout, prio, best), flush=True)
