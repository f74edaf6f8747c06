#!/usr/bin/env python3
"""Where does the bf16 AUC gap of tests/test_auc_parity_gpu.py come from?  Trains the 2-client / 3-round FedOTPLoRA loop
in fp32 and in bf16, then evaluates BOTH final weight sets with an fp32 and with a bf16 engine (2048 test samples per
client): training precision and inference precision separately."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fairfedmed_amd import config as C, federated as F, synth
from fairfedmed_amd.registry import build_trainer
from fairfedmed_amd.trainer import SyntheticFedData
import fairfedmed_amd.trainer  # noqa: F401
from tests.test_trainer_gpu import make_cfg

USERS, ROUNDS, BS, TB, TBS = 2, 3, 8, 32, 64
mcfg = C.vit_tiny(rank=4)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
data = SyntheticFedData(mcfg, USERS, train_batches=6, test_batches=TB, batch_size=BS, signal=0.45, test_batch_size=TBS)
args = F.FedArgs(num_users=USERS, frac=1.0, round=ROUNDS, shared_half_s=True, seed=0)


def trainer(prec):
    cfg = make_cfg(prec=prec, bs=BS)
    cfg.TEST.BATCH_SIZE = TBS
    cfg.OPTIM.LR = 2e-2
    cfg.DATASET.USERS, cfg.TEST.NO_TEST, cfg.TRAIN.METRICS_EVERY = USERS, True, 0
    cfg.DATA, cfg.MODEL.STATE_DICT = data, sd
    return build_trainer(cfg)


res = {}
for prec in ("fp32", "bf16"):
    tr = trainer(prec)
    h = F.run_fedotplora(tr, args, log=lambda *_: None)
    res[prec] = h
    print(prec, "trained: AUC per round", [round(a / 100, 5) for a in h["auc"]])
for wprec in ("fp32", "bf16"):
    for eprec in ("fp32", "bf16"):
        tr = trainer(eprec)
        aucs = []
        for idx in range(USERS):
            tr.model.load_state_dict(res[wprec]["local_weights_per"][idx], strict=False)
            aucs.append(tr.test(idx=idx)[3] / 100)
        print(f"weights trained in {wprec}, evaluated in {eprec}: AUC {np.mean(aucs):.5f}  per client {[round(a, 5) for a in aucs]}")
