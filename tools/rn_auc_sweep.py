#!/usr/bin/env python3
"""RN tower, AUC after equal rounds: bf16 engine against the fp32 engine (which equals the oracle to < 1e-4 in AUC,
tests/test_auc_parity_gpu.py) over task difficulty (signal) and learning rate, and - as the control for 'how much does
this fixture move under ANY 2^-9 perturbation' - the fp32 engine on images perturbed by one bf16 rounding."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth, federated as F
from fairfedmed_amd.registry import build_trainer
from fairfedmed_amd.trainer import SyntheticFedData
import fairfedmed_amd.trainer  # noqa: F401
from tests.test_trainer_gpu import make_cfg

USERS = 2
geom = sys.argv[1] if (len(sys.argv) > 1 and __name__ == "__main__") else "rn_tiny2"
mcfg = getattr(C, geom)(rank=4, num_groups=2)


def run(prec, signal, lr, rounds=3, perturb=False, bn3=1.0, train_b=6, bs=8):
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * bn3
    data = SyntheticFedData(mcfg, USERS, train_batches=train_b, test_batches=32, batch_size=bs, signal=signal, test_batch_size=64,
                            attribute="gender")
    if perturb:
        for d in (data.fed_train_loader_x_dict, data.fed_test_loader_x_dict):
            for l in d.values():
                for b in l.dataset.batches:
                    b["img"] = b["img"].bfloat16().float()
    cfg = make_cfg(prec=prec, bs=bs, rank=4)
    cfg.TEST.BATCH_SIZE = 64
    cfg.OPTIM.LR = lr
    cfg.DATASET.USERS, cfg.TEST.NO_TEST, cfg.TRAIN.METRICS_EVERY = USERS, True, 0
    cfg.DATASET.ATTRIBUTES, cfg.DATASET.ATTRIBUTE_TYPE = ["gender"], "gender"
    cfg.MODEL.GEOMETRY = mcfg
    cfg.DATA, cfg.MODEL.STATE_DICT = data, sd
    h = F.run_fedotplora(build_trainer(cfg), F.FedArgs(num_users=USERS, frac=1.0, round=rounds, shared_half_s=True, seed=0),
                         log=lambda *_: None)
    return [a / 100 for a in h["auc"]]


f = lambda v: "[" + ", ".join("%.5f" % x for x in v) + "]"
for bn3 in ((0.1, 0.05, 0.25) if __name__ == "__main__" else ()):
    for signal, lr, tb in ((0.45, 2e-3, 12), (0.45, 5e-3, 12), (0.25, 5e-3, 12), (0.45, 1e-2, 6)):
        a32 = run("fp32", signal, lr, bn3=bn3, train_b=tb)
        a16 = run("bf16", signal, lr, bn3=bn3, train_b=tb)
        ap = run("fp32", signal, lr, perturb=True, bn3=bn3, train_b=tb)
        print(f"bn3 x{bn3} signal {signal} lr {lr} train_b {tb}: fp32 {f(a32)}  bf16 {f(a16)}  fp32+bf16-rounded pixels {f(ap)}  "
              f"gap bf16 {max(abs(a - b) for a, b in zip(a32, a16)):.5f}  gap perturbed {max(abs(a - b) for a, b in zip(a32, ap)):.5f}",
              flush=True)
