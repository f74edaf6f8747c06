#!/usr/bin/env python3
"""Pick a task difficulty for the ViT-B/16 AUC-after-equal-rounds test: HIP fp32 / bf16 AUC per round by signal and lr."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fairfedmed_amd import config as C, synth, federated as F
from fairfedmed_amd.registry import build_trainer
from fairfedmed_amd.trainer import SyntheticFedData
import fairfedmed_amd.trainer  # noqa: F401
from tests.test_trainer_gpu import make_cfg
mcfg = C.vit_b16(rank=8)
sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
for signal, lr in ((0.45, 2e-2), (0.1, 2e-2), (0.05, 2e-2), (0.03, 2e-2), (0.05, 5e-3)):
    res = {}
    for prec in ("fp32", "bf16"):
        data = SyntheticFedData(mcfg, 2, train_batches=4, test_batches=8, batch_size=8, signal=signal, test_batch_size=64, attribute="race")
        cfg = make_cfg(prec=prec, bs=8, rank=8)
        cfg.TEST.BATCH_SIZE = 64
        cfg.OPTIM.LR = lr
        cfg.DATASET.USERS, cfg.TEST.NO_TEST, cfg.TRAIN.METRICS_EVERY = 2, True, 0
        cfg.INPUT.SIZE = (224, 224)
        cfg.MODEL.BACKBONE.NAME = "ViT-B/16"
        cfg.DATA, cfg.MODEL.STATE_DICT = data, sd
        h = F.run_fedotplora(build_trainer(cfg), F.FedArgs(num_users=2, frac=1.0, round=2, shared_half_s=True, seed=0), log=lambda *_: None)
        res[prec] = [a / 100 for a in h["auc"]]
    print(f"signal {signal} lr {lr}: fp32 {res['fp32']} bf16 {res['bf16']}", flush=True)
