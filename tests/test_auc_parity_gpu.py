"""north_star's end-to-end criterion: after EQUAL ROUNDS of the FedOTPLoRA loop (client sampling, local SGD with the
shared optimizer / StepLR, FedAvg + EMA, per-round evaluation of every client) the HIP trainer's AUC equals that of
the reference algorithm within +-0.002.  The reference side here is the oracle (pinned against the imported reference
on logits, gradients, trajectories, aggregation and AUC) driven through the SAME round loop by a CPU trainer."""
import copy
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from fairfedmed_amd import config as C
from fairfedmed_amd import synth

pytestmark = pytest.mark.gpu
USERS, ROUNDS, BS = 2, 3, 8


class OracleTrainer:
    """The trainer surface federated.run_fedotplora drives, computed by the oracle on the CPU: one SGD state and one
    StepLR counter shared by all clients (federated_main.py:183)."""

    def __init__(self, mcfg, data, sd, lr=1e-3, step_size=2, gamma=0.1):
        from oracle import fairlora_oracle as O
        self.O, self.mcfg, self.sd = O, mcfg, {k: v.clone() for k, v in sd.items()}
        self.keys = synth.trainable_keys(mcfg)
        self.opt = O.SgdState(lr=lr, momentum=0.9, weight_decay=5e-4)
        self.lr0, self.step_size, self.gamma, self.last_epoch = lr, step_size, gamma, 0
        self.fed_train_loader_x_dict, self.fed_test_loader_x_dict = data.fed_train_loader_x_dict, data.fed_test_loader_x_dict
        self.cfg = NS(DATASET=NS(USERS=USERS, ATTRIBUTE_TYPE="race"))
        outer = self

        class _Model:
            def state_dict(self):
                # trainable tensors + (RN) the BatchNorm running statistics / counters: FedAvg averages those too
                return {k: outer.sd[k] for k in outer.keys + synth.buffer_keys(outer.mcfg)}

            def load_state_dict(self, w, strict=True):
                for k, v in w.items():
                    v = v.detach().clone().cpu()
                    # nn.Module.load_state_dict copies into the existing tensor: an averaged (float) counter truncates
                    outer.sd[k] = v.to(torch.int64) if k.endswith("num_batches_tracked") else v.float()
        self.model = _Model()

    def fed_before_train(self): pass
    def fed_after_train(self): pass

    def train(self, idx, global_epoch, is_fed, is_last_client):
        for batch in self.fed_train_loader_x_dict[idx]:
            self.O.train_step(self.sd, self.opt, batch, self.mcfg, self.keys)
        self.last_epoch += 2                   # update_lr: the shared StepLR is stepped once per registered name (2)
        self.opt.lr = self.lr0 * self.gamma ** (self.last_epoch // self.step_size)

    def test(self, idx, current_epoch):
        from fairfedmed_amd import metrics as M
        probs, labels = [], []
        with torch.no_grad():
            for b in self.fed_test_loader_x_dict[idx]:
                logits = self.O.clip_logits(self.sd, b["img"], b["attrs"].t()[0], self.mcfg, training=False)
                probs.append(torch.softmax(logits, -1))
                labels.append(b["label"])
        prob, y = torch.cat(probs).numpy(), torch.cat(labels).numpy()
        pred = prob.argmax(-1)
        acc = 100.0 * float((pred == y).mean())
        return [acc, 100.0 - acc, 100.0 * M.macro_f1(pred, y, 2), 100.0 * M.auc_macro_ovr(prob, y)]


# north_star: AUC within +-0.002 of the reference after equal rounds, in every precision (bf16 is the mode the headline
# throughput is quoted in, fp16 the reference's own 16-bit mode).  2048 test samples per client (32 batches of 64): ~10^6
# score pairs per client, so one swapped pair moves the AUC by 1e-6 and what is left is the systematic effect of the storage type.
#
# Three towers: the tiny ViT (fast), a reduced ModifiedResNet with identity-skip Bottlenecks and train-mode BatchNorm
# (rn_tiny2, gender = 2 groups: the RN50 row's end-to-end criterion), and the full ViT-B/16 r=8 at a reduced schedule
# (2 clients x 2 rounds x 4 batches of 8, 512 test samples per client: the oracle's CPU time bounds it).
#
# The RN fixture (round 6; tools/rn_fixture_search.py has the search, DESIGN.md section 2 the table).  Round 5's fixture
# (lr 2e-3 decaying 10x per client, batch 8, separable classes) sat in a regime where fp32 training LOWERED the AUC from 0.99
# to 0.83 and 16-bit runs differed from it by 0.001-0.004 by the luck of their summation order.  What a +-0.002 criterion needs
# is a task whose AUC is set by the DATA once the model has learnt it, as a trained model's is - not one on the steep part of a
# learning curve, where the oracle ITSELF moves by 0.03 when nothing but its frozen weights are rounded to bf16 (signal 0.13
# without overlap: 0.791 / 0.861 / 0.913 against 0.796 / 0.824 / 0.895), nor CLIP's bn3.weight = 0 initialisation, under which
# every residual branch is 1e-3 of its identity path for the first rounds and vanishes in ANY 16-bit sum (the oracle with
# bf16-stored activations: 0.69 against 0.92).  So: the reference's learning rate 1e-3 (configs/trainers/GLP_OT/rn50_oph.yaml:17)
# held constant over the rounds (OPTIM.STEPSIZE 200; the yaml's default decays it 100x per client), the config's batch size 32,
# bn3.weight x 0.25 (small residual branches, as in a trained ResNet), and classes that OVERLAP (synth.make_batch(overlap=0.3):
# per-sample N(0, 0.3^2) offset on the label-dependent shift of 0.25), so the fp32 oracle's AUC rises monotonically
# 0.788 -> 0.826 -> 0.831 toward the plateau the data sets.  On it the oracle's own precision controls sit at 0.0006
# (bf16-stored activations) and 0.0001 (the image tower's frozen weights rounded to half - the reference's fp16 model) from the
# fp32 run (tools/rn_fixture_search.py).
CASES = {
    #             geometry                               attribute  rounds train_b  bs  test_b test_bs  lr    signal bn3  stepsize overlap
    "vit_tiny": (lambda: C.vit_tiny(rank=4),               "race",   3,     6,      8,  32,    64,     2e-2, 0.45, 1.0,  2,       0.0),
    "rn_tiny2": (lambda: C.rn_tiny2(rank=4, num_groups=2), "gender", 3,     12,     32, 32,    64,     1e-3, 0.25, 0.25, 200,     0.3),
    # (signal 0.1: the AUC sits at 0.78 after two rounds - at 0.45 the 224 x 224 task saturates at 0.9999, tools/vitb_auc_calib.py)
    "vit_b16":  (lambda: C.vit_b16(rank=8),                "race",   2,     4,      8,  8,     64,     2e-2, 0.10, 1.0,  2,       0.0),
}
# Every row is held to north_star's PLAIN bound against the fp32 reference run - fp32 0.0005, bf16 and fp16 0.002 - with no
# expected failures and no differently-prepared oracle.  Measured on this fixture (round 6): RN fp32 2e-5, bf16 0.0010, fp16
# 0.0001; the oracle itself moves by 0.0021 when nothing but the image tower's frozen weights are rounded to bfloat16 and by
# 0.0001 under half rounding - the bf16 engine's weight and activation roundings do not add up in one direction.  The same
# engine under every summation order it has a switch for: tools/rn_auc_ab.py, profiles/r06_rn_auc_ab.txt.
_ORACLE_RUNS = {}


def _oracle_run(name, mcfg, data, sd, args, cfg):
    """The oracle's run does not depend on the HIP precision: computed once per tower."""
    from fairfedmed_amd import federated as F
    if name not in _ORACLE_RUNS:
        tr = OracleTrainer(mcfg, data, sd, lr=cfg.OPTIM.LR, step_size=cfg.OPTIM.STEPSIZE, gamma=cfg.OPTIM.GAMMA)
        tr.cfg.DATASET.ATTRIBUTE_TYPE = cfg.DATASET.ATTRIBUTE_TYPE
        _ORACLE_RUNS[name] = F.run_fedotplora(tr, args, log=lambda *_: None)
    return _ORACLE_RUNS[name]


# "fp16": the reference's own 16-bit format (FFM_F16), held to north_star's plain 0.002 against the fp32-weights oracle on
# every tower
@pytest.mark.parametrize("prec,tol", [("fp32", 0.0005), ("bf16", 0.002), ("fp16", 0.002)])
@pytest.mark.parametrize("tower", list(CASES))
def test_auc_after_equal_rounds(tower, prec, tol):
    from fairfedmed_amd import federated as F
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData
    import fairfedmed_amd.trainer  # noqa: F401
    from tests.test_trainer_gpu import make_cfg
    geom, attribute, rounds, train_b, bs, test_b, test_bs, lr, signal, bn3, stepsize, overlap = CASES[tower]
    mcfg = geom()
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * bn3
    data = SyntheticFedData(mcfg, USERS, train_batches=train_b, test_batches=test_b, batch_size=bs, signal=signal,
                            test_batch_size=test_bs, attribute=attribute, overlap=overlap)
    assert all(len(l.dataset) >= 512 for l in data.fed_test_loader_x_dict.values())
    args = F.FedArgs(num_users=USERS, frac=1.0, round=rounds, shared_half_s=True, seed=0)
    cfg = make_cfg(prec=prec, bs=bs, rank=mcfg.lora.rank)
    cfg.TEST.BATCH_SIZE = test_bs
    cfg.OPTIM.LR, cfg.OPTIM.STEPSIZE = lr, stepsize
    cfg.DATASET.USERS, cfg.TEST.NO_TEST, cfg.TRAIN.METRICS_EVERY = USERS, True, 0
    cfg.DATASET.ATTRIBUTES, cfg.DATASET.ATTRIBUTE_TYPE = [attribute], attribute
    cfg.INPUT.SIZE = (mcfg.vision.image_size,) * 2
    if tower == "vit_b16":
        cfg.MODEL.BACKBONE.NAME = "ViT-B/16"
    else:
        cfg.MODEL.GEOMETRY = mcfg
    cfg.DATA, cfg.MODEL.STATE_DICT = data, sd
    hip = F.run_fedotplora(build_trainer(cfg), args, log=lambda *_: None)
    ref = _oracle_run(tower, mcfg, data, sd, args, cfg)
    # test() reports the AUC in percent, as the reference's evaluator does (evaluation/evaluator_oph.py:88-96)
    hip_auc, ref_auc = [a / 100.0 for a in hip["auc"]], [a / 100.0 for a in ref["auc"]]
    print(tower, prec, "AUC per round  HIP", [round(a, 5) for a in hip_auc], " oracle", [round(a, 5) for a in ref_auc])
    assert max(ref_auc) - min(ref_auc) > 0.002 or abs(ref_auc[-1] - 0.5) > 0.02, "the run must move the AUC"
    if tower.startswith("rn"):
        # the fixture's own conditions: the fp32 reference rises monotonically into 0.75 - 0.92 (not saturated, not collapsing)
        assert all(b_ > a_ for a_, b_ in zip(ref_auc, ref_auc[1:])) and 0.75 <= ref_auc[-1] <= 0.92, ref_auc
    for r in range(rounds):
        assert abs(hip_auc[r] - ref_auc[r]) <= tol, (r, hip_auc, ref_auc, tol)
        # accuracy (percent, mean over the clients): fp32 may differ by one test sample of one client, 16-bit modes by 5 points
        one_sample = 100.0 / (test_b * test_bs)
        assert abs(hip["acc"][r] - ref["acc"][r]) <= (1e-9 if prec == "fp32" and tower == "vit_tiny" else one_sample if prec == "fp32" else 5.0)
    if prec == "fp32" and tower == "vit_tiny":
        for k, v in ref["global_weights"].items():
            a, b = hip["global_weights"][k].double().cpu(), v.double()
            assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-7, k
