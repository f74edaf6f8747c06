#!/usr/bin/env python3
"""In-kernel phase stamps of the attn3 kernels (csrc/attention3.hip built with -DFFM_ATTN3_STAMPS into
tools/proto/libffm_a3stamps.so: `tools/attn_phases.sh build` makes it):
    FFM_LIB_PATH=tools/proto/libffm_a3stamps.so python tools/attn_stamps.py [fwd|dkv]
Prints, per phase boundary, the median over the waves of (stamp - the block's first stamp) in shader cycles and the
spread; the stamps fence the scheduler, so only the SHARES mean anything."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fairfedmed_amd import ops, _lib

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, L, H = 32, 197, 12
E = H * 64
g = torch.Generator("cuda").manual_seed(1)
qkv = torch.randn(B * L, 3 * E, device="cuda", generator=g).bfloat16()
out = torch.empty(B * L, E, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, L, device="cuda")
dout = torch.randn(B * L, E, device="cuda", generator=g).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, H, L, device="cuda")
for _ in range(20):                              # warm clocks
    ops.attention_fwd(qkv, out, lse, B, L, H, False)
    ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, L, H, False)
torch.cuda.synchronize()
if which == "fwd":
    ops.attention_fwd(qkv, out, lse, B, L, H, False)
    names = ["entry", "prologue issued", "chunk 0 landed", "all tiles done", "stored"]
else:
    ops.attention_bwd(qkv, out, dout, lse, delta, dqkv, B, L, H, False)      # the dq kernel carries no stamps: dkv's remain
    names = ["entry", "prologue issued + frags back", "chunk 0 landed", "all tiles done", "stored"]
torch.cuda.synchronize()
lib = _lib.load()
n = 768 * 4 * 16
buf = (C.c_ulonglong * n)()
assert lib.ffm_attn3_read_stamps(buf, n) == 0
st = np.array(buf, dtype=np.uint64).reshape(768, 4, 16).astype(np.int64)
last = len(names) - 1
wall = (st[:, :, 15] - st[:, :, 14]).astype(np.float64) * 10.0          # s_memrealtime: 100 MHz -> ns
cyc = (st[:, :, last] - st[:, :, 0]).astype(np.float64)
ok = (st[:, :, last] > 0) & (wall > 0)
print(f"{which}: in-kernel clock {np.median(cyc[ok] / wall[ok]):.2f} GHz; wave lifetime median {np.median(wall[ok]) / 1e3:.2f} us, max {wall[ok].max() / 1e3:.2f} us; "
      f"kernel span (realtime, first entry to last exit) {(st[:, :, 15][ok].max() - st[:, :, 14][ok].min()) * 10 / 1e3:.2f} us")
for part in ("half 0 (waves 0-3)", "half 1 (waves 0-2)"):
    half = 0 if "half 0" in part else 1
    sel = st[(np.arange(768) >> 3) % 2 == half][:, : (4 if half == 0 else 3)]
    print(part)
    for i in range(1, len(names)):
        d = (sel[:, :, i] - sel[:, :, i - 1]).ravel()
        print(f"  {i:2d} {names[i]:28s} median {int(np.median(d)):6d}  p10 {int(np.percentile(d, 10)):6d} p90 {int(np.percentile(d, 90)):6d}")
    tot = (sel[:, :, last] - sel[:, :, 0]).ravel()
    print(f"     {'total':28s} median {int(np.median(tot)):6d}  p10 {int(np.percentile(tot, 10)):6d} p90 {int(np.percentile(tot, 90)):6d}")
