#!/usr/bin/env python3
"""RN tower, AUC after equal rounds on the round-6 fixture (tests/test_auc_parity_gpu.py CASES["rn_tiny2"]): every storage
mode under every summation order the engine has a switch for, against the fp32 engine (which sits < 1e-4 from the oracle,
same test).  One child process per variant: the C++ switches are read once per process.

    python tools/rn_auc_ab.py [signal,lr,rounds,train_b,bs,bn3[,test_b,overlap] ...]      (GPU box)

Answers the round-5 advisor's question - is the 16-bit distance summation ORDER or a BIAS in one of the fused paths
(FFM_EPI_BNBWD, split-K column sums, folded BatchNorm, four-stage ring)? - with the full A/B table, and prices the noise
floor of the fixture the test holds to +-0.002."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [
    ("default", {}),
    ("FFM_BN_BWD_FUSED=0", {"FFM_BN_BWD_FUSED": "0"}),
    ("FFM_BN_FOLD_ROWS=0", {"FFM_BN_FOLD_ROWS": "0"}),
    ("FFM_GEMM_DEEP=0", {"FFM_GEMM_DEEP": "0"}),
    ("FFM_CONV_DEEP=0", {"FFM_CONV_DEEP": "0"}),
    ("all four off", {"FFM_BN_BWD_FUSED": "0", "FFM_BN_FOLD_ROWS": "0", "FFM_GEMM_DEEP": "0", "FFM_CONV_DEEP": "0"}),
]


def child(spec, prec):
    import torch  # noqa: F401
    from fairfedmed_amd import config as C, synth, federated as F
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData
    import fairfedmed_amd.trainer  # noqa: F401
    from tests.test_trainer_gpu import make_cfg
    p = spec.split(",")
    signal, lr, rounds, train_b, bs, bn3 = float(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4]), float(p[5])
    overlap = float(p[7]) if len(p) > 7 else 0.0
    mcfg = C.rn_tiny2(rank=4, num_groups=2)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * bn3
    data = SyntheticFedData(mcfg, 2, train_batches=train_b, test_batches=32, batch_size=bs, signal=signal, test_batch_size=64,
                            attribute="gender", overlap=overlap)
    cfg = make_cfg(prec=prec, bs=bs, rank=4)
    cfg.TEST.BATCH_SIZE = 64
    cfg.OPTIM.LR, cfg.OPTIM.STEPSIZE = lr, 200
    cfg.DATASET.USERS, cfg.TEST.NO_TEST, cfg.TRAIN.METRICS_EVERY = 2, True, 0
    cfg.DATASET.ATTRIBUTES, cfg.DATASET.ATTRIBUTE_TYPE = ["gender"], "gender"
    cfg.MODEL.GEOMETRY = mcfg
    cfg.DATA, cfg.MODEL.STATE_DICT = data, sd
    h = F.run_fedotplora(build_trainer(cfg), F.FedArgs(num_users=2, frac=1.0, round=rounds, shared_half_s=True, seed=0), log=lambda *_: None)
    print("AUC " + json.dumps([a / 100 for a in h["auc"]]), flush=True)


def run_child(spec, prec, env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", spec, prec], env=e, capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("AUC "):
            return json.loads(line[4:])
    raise RuntimeError(f"{spec} {prec} {env}: rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    f = lambda v: "[" + ", ".join("%.5f" % x for x in v) + "]"
    for spec in sys.argv[1:] or ["0.25,1e-3,3,12,32,0.25,32,0.3"]:
        a32 = run_child(spec, "fp32", {})
        print(f"{spec}: fp32 engine {f(a32)}", flush=True)
        for prec in ("bf16", "fp16"):
            worst = 0.0
            for name, env in VARIANTS:
                a = run_child(spec, prec, env)
                gaps = [abs(x - y) for x, y in zip(a, a32)]
                worst = max(worst, max(gaps))
                print(f"  {prec} {name:20s} {f(a)}  gaps {f(gaps)}  max {max(gaps):.5f}", flush=True)
            print(f"  {prec}: worst over the {len(VARIANTS)} summation orders {worst:.5f}", flush=True)
