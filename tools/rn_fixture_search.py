#!/usr/bin/env python3
"""CPU search for a STABLE RN AUC-after-equal-rounds fixture (tests/test_auc_parity_gpu.py, round 6).

Runs the FedOTPLoRA loop on the oracle only (no GPU): fp32, fp32 with bf16-stored activations (oracle.STORE: the
independent statement of a 16-bit-storage trunk, the proxy for what ANY bf16 engine can reach) and fp32 on half-rounded
frozen weights (what the fp16 mode's model is).  A fixture is usable when the fp32 AUC rises monotonically into 0.75-0.92
and both controls stay well inside 0.002 of it.
    python tools/rn_fixture_search.py "signal,lr,rounds,train_b,bs,bn3[,test_b[,overlap]]" ...
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fairfedmed_amd import config as C, synth, federated as F
from fairfedmed_amd.trainer import SyntheticFedData
from tests.test_auc_parity_gpu import OracleTrainer
from oracle import fairlora_oracle as O

USERS = 2
torch.set_num_threads(8)
mcfg = getattr(C, os.environ.get("GEOM", "rn_tiny2"))(rank=4, num_groups=2)
f = lambda v: "[" + ", ".join("%.5f" % x for x in v) + "]"


def run(sd, data, lr, rounds, store=None):
    O.STORE = store
    try:
        tr = OracleTrainer(mcfg, data, sd, lr=lr, step_size=int(os.environ.get("STEPSIZE", 200)), gamma=0.1)
        tr.cfg.DATASET.ATTRIBUTE_TYPE = "gender"
        h = F.run_fedotplora(tr, F.FedArgs(num_users=USERS, frac=1.0, round=rounds, shared_half_s=True, seed=0), log=lambda *_: None)
    finally:
        O.STORE = None
    return [a / 100 for a in h["auc"]]


for spec in sys.argv[1:]:
    p = spec.split(",")
    signal, lr, rounds, train_b, bs, bn3 = float(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4]), float(p[5])
    test_b = int(p[6]) if len(p) > 6 else 32
    overlap = float(p[7]) if len(p) > 7 else 0.0
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * bn3
    data = SyntheticFedData(mcfg, USERS, train_batches=train_b, test_batches=test_b, batch_size=bs, signal=signal, test_batch_size=64,
                            attribute="gender", overlap=overlap)
    t0 = time.time()
    a32 = run(sd, data, lr, rounds)
    t1 = time.time()
    out = f"{spec}: fp32 {f(a32)} ({t1 - t0:.0f}s)"
    if os.environ.get("CONTROLS", "1") == "1":
        ab = run(sd, data, lr, rounds, store=O.store_bf16) if os.environ.get("CONTROLS_STORE", "1") == "1" else a32
        train = set(synth.trainable_keys(mcfg))
        # what the engines round: the image tower's frozen matrices (convolutions, attention pool, positional embedding);
        # biases, BatchNorm and the whole text tower stay float32 (engine_rn._load_vision_frozen, engine.py: x3 text tower)
        vis = lambda k, v: k.startswith("image_encoder.") and k not in train and v.is_floating_point() and v.dim() >= 2
        sd_h = {k: (v.half().float() if vis(k, v) else v) for k, v in sd.items()}
        ah = run(sd_h, data, lr, rounds)
        sd_b = {k: (v.bfloat16().float() if vis(k, v) else v) for k, v in sd.items()}
        abw = run(sd_b, data, lr, rounds)
        abwb = run(sd_b, data, lr, rounds, store=O.store_bf16) if os.environ.get("CONTROLS_STORE", "1") == "1" else abw
        out += (f"  bf16-store {f(ab)} gap {max(abs(a - b) for a, b in zip(a32, ab)):.5f}"
                f"  half-w {f(ah)} gap {max(abs(a - b) for a, b in zip(a32, ah)):.5f}"
                f"  bf16-w {f(abw)} gap {max(abs(a - b) for a, b in zip(a32, abw)):.5f}"
                f"  bf16-w+store gap vs fp32 {max(abs(a - b) for a, b in zip(a32, abwb)):.5f} vs bf16-w {max(abs(a - b) for a, b in zip(abw, abwb)):.5f}")
    print(out, flush=True)
