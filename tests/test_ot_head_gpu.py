"""Sinkhorn / COT logits heads on the GPU: csrc/head_ot.hip through the engine against the
oracle and the goldens produced by the imported reference (tests/golden/ot.npz)."""
import dataclasses
import json
import os

import numpy as np
import pytest
import torch

from fairfedmed_amd import config as C
from fairfedmed_amd import synth

pytestmark = pytest.mark.gpu


def cos(got, ref):
    got = torch.as_tensor(got).double().cpu().flatten()
    ref = torch.as_tensor(ref).double().cpu().flatten()
    return float(torch.dot(got, ref) / (got.norm() * ref.norm()).clamp_min(1e-300))


def rel(got, ref):
    got = torch.as_tensor(got).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def to_dev(batch):
    return batch["img"].cuda(), batch["attrs"].t()[0].cuda(), batch["label"].cuda()


@pytest.mark.parametrize("ot,top", [("Sinkhorn", 1.0), ("COT", 0.8)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_ot_head_step_vs_reference_golden_and_oracle(golden_dir, ot, top, dtype):
    from oracle import fairlora_oracle as O
    from fairfedmed_amd.engine import FairLoRAEngine
    gold = np.load(os.path.join(golden_dir, "ot.npz"))
    meta = json.load(open(os.path.join(golden_dir, "ot.json")))
    tag = f"ot_{ot.lower()}"
    mcfg = dataclasses.replace(C.vit_tiny(rank=4), ot=ot, ot_top_percent=top)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, 8, seed=1234)
    keys = synth.trainable_keys(mcfg)
    eng = FairLoRAEngine(mcfg, sd, dtype=dtype, max_images=8)
    out = eng.forward_backward(*to_dev(batch))
    f32 = dtype == torch.float32
    assert int(out["finite"]) == 1
    assert rel(out["logits"], gold[f"{tag}.logits"]) < (2e-5 if f32 else 3e-2)
    l0 = meta[f"{tag}.loss0"]
    assert abs(float(out["loss"]) - l0) <= (1e-5 if f32 else 1e-2) * abs(l0)
    loss, logits, grads = O.loss_and_grads(sd, batch, mcfg, keys)
    for k in keys:
        g, ref = eng.params.view(k, "grad"), grads[k]
        if float(ref.abs().max()) == 0.0:
            assert float(g.abs().max()) < 1e-12, k
        elif f32:
            assert rel(g, ref) < 2e-3 and rel(g, gold[f"{tag}.grad.{k}"]) < 2e-3, (k, rel(g, ref))
        else:
            assert cos(g, ref) > 0.985, (k, cos(g, ref))
    assert rel(eng.forward(batch["img"].cuda(), batch["attrs"].t()[0].cuda()), out["logits"]) < 1e-6
    if f32:                                                            # three SGD steps on the reference's trajectory
        for i, ref in enumerate(meta[f"{tag}.traj"]):
            o = out if i == 0 else eng.forward_backward(*to_dev(batch))
            eng.sgd_step(1e-3, 0.9, 5e-4, repeats=2)                   # shared optimizer, two names (quirk 9)
            assert abs(float(o["loss"]) - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (i, float(o["loss"]), ref)


@pytest.mark.parametrize("ot", ["Sinkhorn", "COT"])
def test_ot_head_on_the_resnet_tower_and_no_early_stop(ot):
    """The heads on the RN tower's attention-pool tokens (4 tokens, 256-d), and a threshold that is never met (all
    max_iter iterations run), against the oracle."""
    from oracle import fairlora_oracle as O
    from fairfedmed_amd.engine_rn import create_engine
    import copy
    mcfg = dataclasses.replace(C.rn_tiny(rank=4, num_groups=2), ot=ot, ot_thresh=0.0, ot_max_iter=7, ot_top_percent=0.7)
    sd = synth.make_state_dict(mcfg, seed=2, lora_init="random")
    batch = synth.make_batch(mcfg, 5, seed=9)
    keys = synth.trainable_keys(mcfg)
    eng = create_engine(mcfg, sd, dtype=torch.float32, max_images=5)
    out = eng.forward_backward(*to_dev(batch))
    loss, logits, grads = O.loss_and_grads(copy.deepcopy(sd), batch, mcfg, keys)
    assert rel(out["logits"], logits) < 5e-5 and abs(float(out["loss"]) - float(loss)) <= 2e-5 * abs(float(loss))
    assert int(eng.ot_istop) == 6
    for k in keys:
        if float(grads[k].abs().max()) > 0:
            assert cos(eng.params.view(k, "grad"), grads[k]) > 1 - 1e-4, k


def test_trainer_selects_the_ot_head():
    from tests.test_trainer_gpu import make_cfg
    from fairfedmed_amd.trainer import GLP_OT_SVLoRA, SyntheticFedData
    mcfg = C.vit_tiny(rank=4)
    cfg = make_cfg(prec="fp32")
    cfg.TRAINER.GLP_OT.OT, cfg.TRAINER.GLP_OT.EPS, cfg.TRAINER.GLP_OT.THRESH = "COT", 0.1, 1e-3
    cfg.TRAINER.GLP_OT.MAX_ITER, cfg.TRAINER.GLP_OT.TOP_PERCENT = 100, 0.8
    tr = GLP_OT_SVLoRA(cfg, data=SyntheticFedData(mcfg, 1, 2, 1, 8))
    assert tr.engine.ot == "COT" and tr.engine.cfg.ot_top_percent == 0.8
    tr.num_batches, tr.batch_idx = 10, 0
    s = tr.forward_backward(synth.make_batch(mcfg, 8, seed=3))
    assert np.isfinite(s["loss"])
    cfg.TRAINER.GLP_OT.OT = "Wasserstein"
    with pytest.raises(NotImplementedError):
        GLP_OT_SVLoRA(cfg, data=SyntheticFedData(mcfg, 1, 1, 1, 8))


@pytest.mark.parametrize("ot", ["Sinkhorn", "COT"])
def test_ot_head_with_the_3d_oct_front_end(ot):
    """3D OCT: every slice group is an image of the transport problem and the logits are averaged over the slices
    afterwards (trainers/GLP_OT_SVLoRA.py:752-754); the stopping test runs over all B * S * n_cls problems."""
    from oracle import fairlora_oracle as O
    from fairfedmed_amd.engine import FairLoRAEngine
    mcfg = dataclasses.replace(C.vit_tiny_3d(rank=4, dim_per_3d_slice=4), ot=ot, ot_top_percent=0.9)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    batch = synth.make_batch(mcfg, 4, seed=21)                       # 4 volumes x 2 slice groups = 8 ViT images
    keys = synth.trainable_keys(mcfg)
    eng = FairLoRAEngine(mcfg, sd, dtype=torch.float32, max_images=8)
    out = eng.forward_backward(*to_dev(batch))
    loss, logits, grads = O.loss_and_grads(sd, batch, mcfg, keys)
    assert rel(out["logits"], logits) < 3e-5 and abs(float(out["loss"]) - float(loss)) <= 2e-5 * abs(float(loss))
    for k in keys:
        if float(grads[k].abs().max()) > 0:
            assert rel(eng.params.view(k, "grad"), grads[k]) < 3e-3, (k, rel(eng.params.view(k, "grad"), grads[k]))
