#!/usr/bin/env python3
"""How noisy is the bf16 AUC-after-equal-rounds of the RN fixture (tests/test_auc_parity_gpu.py: rn_tiny2)?  The same bf16
engine under summation orders that differ only in rounding (FFM_BN_BWD_FUSED 0 / 1 in this process; run the script again
with FFM_BN_FOLD_ROWS=0 for two more) against the fp32 engine, over the learning rate and bn3 scale of the fixture.
    [SIGNALS=0.45,0.2] python tools/rn_auc_noise.py [lr ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rn_auc_sweep import run, f          # noqa: E402  (its module-level sweep is skipped below)

lrs = [float(a) for a in sys.argv[1:]] or [2e-3, 1e-3, 5e-4]
signals = [float(a) for a in os.environ.get("SIGNALS", "0.45").split(",")]
for bn3 in (0.25,):
    for lr in lrs:
        for signal in signals:
            a32 = run("fp32", signal, lr, bn3=bn3, train_b=12)
            gaps = []
            for fused in ("0", "1"):
                os.environ["FFM_BN_BWD_FUSED"] = fused
                a16 = run("bf16", signal, lr, bn3=bn3, train_b=12)
                gaps.append([abs(a - b) for a, b in zip(a32, a16)])
            print(f"bn3 x{bn3} lr {lr} signal {signal}: fp32 {f(a32)}  bf16 gaps per round, FFM_BN_BWD_FUSED=0: {f(gaps[0])}  =1: {f(gaps[1])}"
                  f"  (FFM_BN_FOLD_ROWS={os.environ.get('FFM_BN_FOLD_ROWS', 'default')})", flush=True)
