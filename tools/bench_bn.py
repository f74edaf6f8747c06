#!/usr/bin/env python3
"""BatchNorm forward / backward (column sums -> finalize -> apply) per call on the RN50 maps at batch 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import ops

dt = torch.bfloat16
for rows, C in [(25088, 256), (6272, 256), (6272, 512), (6272, 1024), (1568, 512), (1568, 2048)]:
    g = torch.Generator("cuda").manual_seed(1)
    x = torch.randn(rows, C, device="cuda", generator=g).to(dt)
    dy = torch.randn(rows, C, device="cuda", generator=g).to(dt)
    res = torch.randn(rows, C, device="cuda", generator=g).to(dt)
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    mean, rstd = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    part = torch.empty(ops.bn_blocks(rows) * 2 * C, device="cuda")
    k12, dg, db = torch.empty(2 * C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    y, dx, gout = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)

    def t(fn, n=50):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    f = t(lambda: ops.bn_fwd(x, gamma, beta, rm, rv, mean, rstd, part, y, True, True, res))
    b = t(lambda: ops.bn_bwd(dy, y, x, gamma, mean, rstd, part, k12, dg, db, dx, g_out=gout))
    print("rows %6d C %5d : fwd %6.1f us  bwd %6.1f us" % (rows, C, f, b), flush=True)
