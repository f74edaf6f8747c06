"""Trainer-level boundary on the GPU: state_dict contract, the FairLoRALinear
layer against the reference's golden layer vectors, and the GLP_OT_SVLoRA
trainer API that federated_main.py drives."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from fairfedmed_amd import config as C
from fairfedmed_amd import synth
from tests.golden.make_golden import LAYER_CASES, layer_inputs, sub

pytestmark = pytest.mark.gpu


def rel(got, ref):
    got = torch.as_tensor(got).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def test_state_dict_keys_shapes_and_sharing():
    from fairfedmed_amd.model import CustomCLIP
    mcfg = C.vit_tiny(rank=4)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    model = CustomCLIP(mcfg, sd, dtype=torch.float32, max_images=4)
    got = model.state_dict()
    man = synth.manifest(mcfg)
    assert list(got.keys()) == list(man.keys())                 # same keys, same order as the reference
    for k, shp in man.items():
        assert tuple(got[k].shape) == tuple(shp) and got[k].dtype == torch.float32, k
        assert torch.equal(got[k].cpu(), sd[k]), k
    train = {n for n, p in model.named_parameters() if p.requires_grad}
    assert train == set(synth.trainable_keys(mcfg))
    # trainable parameters alias the engine's flat buffer; load_state_dict(strict=False) updates it in place
    k = "image_encoder.transformer.resblocks.0.mlp.c_fc.lora_S.weight"
    new = {k: torch.full_like(sd[k], 0.25), "prompt_learner.ctx": torch.zeros_like(sd["prompt_learner.ctx"])}
    model.load_state_dict(new, strict=False)
    assert float(model.engine.params.view(k).mean()) == 0.25
    assert float(model.engine.params.view("prompt_learner.ctx").abs().max()) == 0.0


@pytest.mark.parametrize("case", LAYER_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_fairlora_linear_vs_reference_golden(golden_dir, case, dtype):
    """FairLoRALinear.forward/backward (HIP) vs the imported reference layer, nn.Linear and 1x1-conv (RN50) forms."""
    from fairfedmed_amd.model import FairLoRALinear
    unit = np.load(os.path.join(golden_dir, "unit.npz"))
    name, L, Bn, fin, fout, r, G, S, hw = case
    x, g, W, bias, A, Sm, Bm, attr = layer_inputs(*case)
    if hw:
        lin = torch.nn.Conv2d(fin, fout, 1, bias=False)
        lin.weight.data = W.reshape(fout, fin, 1, 1).clone()
    else:
        lin = torch.nn.Linear(fin, fout)
        lin.weight.data, lin.bias.data = W.clone(), bias.clone()
    layer = FairLoRALinear(lin.cuda(), rank=r, alpha=2.0, num_attrs=G)
    layer.lora_A.weight.data.copy_(A)
    layer.lora_S.weight.data.copy_(Sm)
    layer.lora_B.weight.data.copy_(Bm)
    if hw:
        xin = x.permute(1, 2, 0).reshape(Bn, fin, hw[0], hw[1]).cuda().to(dtype).requires_grad_(True)
        y = layer(xin, attr.cuda())
        y.backward(g.reshape(hw[0], hw[1], Bn, fout).permute(2, 3, 0, 1).cuda().to(dtype))
        y_tok = y.detach().reshape(Bn, fout, -1).permute(2, 0, 1)
        dx_tok = xin.grad.reshape(Bn, fin, -1).permute(2, 0, 1)
    else:
        xin = x.cuda().to(dtype).requires_grad_(True)
        y = layer(xin, attr.cuda())
        y.backward(g.cuda().to(dtype))
        y_tok, dx_tok = y.detach(), xin.grad
    f32 = dtype == torch.float32
    t1, t2 = (3e-5, 1e-4) if f32 else (1.5e-2, 4e-2)
    pick = lambda t: t.float().cpu().numpy() if t.numel() <= 65536 else sub(t.float().cpu())
    assert rel(pick(y_tok), unit[f"layer.{name}.y"]) < t1
    assert rel(pick(dx_tok), unit[f"layer.{name}.dx"]) < t1
    assert rel(layer.lora_A.weight.grad, unit[f"layer.{name}.dA"]) < t2
    assert rel(layer.lora_S.weight.grad, unit[f"layer.{name}.dS"]) < t2
    assert rel(layer.lora_B.weight.grad, unit[f"layer.{name}.dB"]) < t2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_fairlora_linear_global_s_vs_reference_golden(golden_dir, dtype):
    """FairLoRALinear(global_s=True): lora_S_global [r] is added to every sample's singular values
    (trainers/GLP_OT_SVLoRA.py:359-363, 418-422, 467-468); forward, all five gradients and weight() vs the reference."""
    from fairfedmed_amd.model import FairLoRALinear
    unit = np.load(os.path.join(golden_dir, "unit.npz"))
    case = LAYER_CASES[0]
    name, L, Bn, fin, fout, r, G, S, hw = case
    x, g, W, bias, A, Sm, Bm, attr = layer_inputs(*case)
    lin = torch.nn.Linear(fin, fout)
    lin.weight.data, lin.bias.data = W.clone(), bias.clone()
    layer = FairLoRALinear(lin.cuda(), rank=r, alpha=2.0, global_s=True, num_attrs=G)
    assert tuple(layer.lora_S_global.weight.shape) == (r,) and layer.lora_S_global.weight.is_cuda
    assert rel(layer.lora_S_global.weight.detach(), unit[f"layer.{name}.gs.sg_init"]) < 1e-7
    layer.lora_A.weight.data.copy_(A)
    layer.lora_S.weight.data.copy_(Sm)
    layer.lora_B.weight.data.copy_(Bm)
    layer.lora_S_global.weight.data.copy_(torch.from_numpy(unit[f"layer.{name}.gs.Sg"]))
    xin = x.cuda().to(dtype).requires_grad_(True)
    y = layer(xin, attr.cuda())
    y.backward(g.cuda().to(dtype))
    t1, t2 = (3e-5, 1e-4) if dtype == torch.float32 else (1.5e-2, 4e-2)
    assert rel(y.detach().float().cpu(), unit[f"layer.{name}.gs.y"]) < t1
    assert rel(xin.grad.float().cpu(), unit[f"layer.{name}.gs.dx"]) < t1
    for nm in ("A", "S", "B", "S_global"):
        assert rel(getattr(layer, "lora_" + nm).weight.grad, unit[f"layer.{name}.gs.d{nm}"]) < t2, nm
    assert rel(layer.weight(xin.detach().float(), attr.cuda()).detach(), unit[f"layer.{name}.gs.weight_attr"]) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_lora_linear_vs_reference_golden(golden_dir, dtype):
    """LoRALinear (plain LoRA of the RN50 attention pool) on the HIP kernels vs the imported reference layer."""
    from fairfedmed_amd.model import LoRALinear
    from tests.golden.make_golden import rng_tensor
    unit = np.load(os.path.join(golden_dir, "unit.npz"))
    L, Bn, fin, fout, r = 50, 4, 128, 192, 8
    lin = torch.nn.Linear(fin, fout)
    lin.weight.data = rng_tensor("lora_plain.W", (fout, fin)) * fin ** -0.5
    lin.bias.data = rng_tensor("lora_plain.b", (fout,)) * 0.1
    layer = LoRALinear(lin.cuda(), rank=r, alpha=2.0)
    layer.lora_A.weight.data.copy_(rng_tensor("lora_plain.A", (fin, r)) * 0.1)
    layer.lora_B.weight.data.copy_(rng_tensor("lora_plain.B", (r, fout)))
    xin = rng_tensor("lora_plain.x", (L, Bn, fin)).cuda().to(dtype).requires_grad_(True)
    y = layer(xin)
    y.backward(rng_tensor("lora_plain.g", (L, Bn, fout)).cuda().to(dtype))
    t1, t2 = (3e-5, 1e-4) if dtype == torch.float32 else (1.5e-2, 4e-2)
    assert rel(y.detach().float().cpu(), unit["lora_plain.y"]) < t1
    assert rel(xin.grad.float().cpu(), unit["lora_plain.dx"]) < t1
    assert rel(layer.lora_A.weight.grad, unit["lora_plain.dA"]) < t2
    assert rel(layer.lora_B.weight.grad, unit["lora_plain.dB"]) < t2
    assert rel(layer.weight().detach(), unit["lora_plain.weight"]) < 1e-6


def make_cfg(prec="fp32", rank=4, bs=8):
    return NS(
        SEED=1, OUTPUT_DIR="", VERBOSE=False,
        INPUT=NS(PIXEL_MEAN=list(C.CLIP_PIXEL_MEAN), PIXEL_STD=list(C.CLIP_PIXEL_STD), SIZE=(64, 64)),
        DATASET=NS(NAME="FairFedMed", ATTRIBUTES=["race"], ATTRIBUTE_TYPE="race"),
        MODEL=NS(BACKBONE=NS(NAME="tiny"), GEOMETRY=C.vit_tiny(), STATE_DICT=None),
        TRAINER=NS(NAME="GLP_OT_SVLoRA", LAMBDA_FAIRNESS=0.0,
                   GLP_OT=NS(N=2, N_CTX=4, PREC=prec, OT="None"),
                   GLP_OT_LORA=NS(RANK=rank, ALPHA=2.0, TYPE="FairLoRA", GLOBAL_S=False, DISABLE_ATTR=False,
                                  UNFREEZE_IMAGE_ENCODER=True)),
        OPTIM=NS(NAME="sgd", LR=1e-3, MOMENTUM=0.9, WEIGHT_DECAY=5e-4, LR_SCHEDULER="single_step", STEPSIZE=2,
                 GAMMA=0.1, MAX_EPOCH=1),
        DATALOADER=NS(TRAIN_X=NS(BATCH_SIZE=bs)), TEST=NS(BATCH_SIZE=bs, NO_TEST=True),
        TRAIN=NS(METRICS_EVERY=1, CHECKPOINT_FREQ=0),
    )


def test_trainer_3d_oct_trajectory(golden_dir):
    """DATASET.MODALITY_TYPE 'oct_bscans': the trainable per-slice conv trains with the LoRA factors
    (trainers/GLP_OT_SVLoRA.py:584-595, 862-863); trajectory of the reference's forward_backward."""
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData, _ListDataset, _Loader
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    gold = np.load(os.path.join(golden_dir, "tiny3d.npz"))
    mcfg = C.vit_tiny_3d(rank=4, dim_per_3d_slice=4)
    cfg = make_cfg(bs=6)
    cfg.DATASET.MODALITY_TYPE, cfg.DATASET.DIM_PER_3D_SLICE = "oct_bscans", 4
    data = SyntheticFedData(mcfg, 1, 1, 1, 6)
    batch = synth.make_batch(mcfg, 6, seed=1234)
    data.fed_train_loader_x_dict[0] = _Loader(_ListDataset([batch], ["race"], {"race": 3}))
    cfg.DATA = data
    cfg.MODEL.STATE_DICT = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    tr = build_trainer(cfg)
    assert tr.engine.is3d and tr.engine.max_images == 12
    tr.num_batches = 10 ** 9
    for i, ref in enumerate(meta["tiny3d_r4.traj"]):
        tr.batch_idx = i
        s = tr.forward_backward(batch)
        assert abs(s["loss"] - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (s, ref)
        assert abs(s["acc"] - ref["acc"]) < 1e-3 and abs(s["auc"] - ref["auc"]) < 1e-9
    sd = tr.model.state_dict()
    for k in ("proj_per_3d_slice.weight", "proj_per_3d_slice.bias"):
        ref = torch.from_numpy(gold[f"tiny3d_r4.post.{k}"])
        assert float((sd[k].cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-7, k


def test_trainer_reproduces_reference_trajectory(golden_dir):
    """forward_backward through the registry-built trainer == the reference's forward_backward trajectory."""
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData, _ListDataset, _Loader
    import fairfedmed_amd.trainer  # noqa: F401
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    mcfg = C.vit_tiny(rank=4)
    cfg = make_cfg()
    data = SyntheticFedData(mcfg, 1, 1, 1, 8)
    batch = synth.make_batch(mcfg, 8, seed=1234)                      # the golden batch
    data.fed_train_loader_x_dict[0] = _Loader(_ListDataset([batch], ["race"], {"race": 3}))
    cfg.DATA = data
    cfg.MODEL.STATE_DICT = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    tr = build_trainer(cfg)
    assert type(tr).__name__ == "GLP_OT_SVLoRA"
    tr.num_batches = 10 ** 9
    for i, ref in enumerate(meta["tiny_r4.traj"]):
        tr.batch_idx = i
        s = tr.forward_backward(batch)
        assert abs(s["loss"] - ref["loss"]) <= 1e-4 * abs(ref["loss"])
        assert abs(s["acc"] - ref["acc"]) < 1e-3 and abs(s["auc"] - ref["auc"]) < 1e-9
    assert tr.fed_train_loader_x_dict[0].dataset.count_by_attribute("race") == \
        np.bincount(batch["attrs"][:, 0].numpy(), minlength=3).tolist()


@pytest.mark.parametrize("tag,ltype,gs", [("tiny_globals", "FairLoRA", True), ("tiny_svlora", "SVLoRA", False),
                                          ("tiny_svlora_globals", "SVLoRA", True), ("tiny_lora", "LoRA", False)])
def test_trainer_other_adapter_types_and_global_s(golden_dir, tag, ltype, gs):
    """TRAINER.GLP_OT_LORA.TYPE in {LoRA, SVLoRA} and GLOBAL_S through the registry-built trainer
    (trainers/GLP_OT_SVLoRA.py:516-540, 838-840): state_dict keys / shapes / order of the reference, its
    forward_backward trajectory and final weights."""
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData, _ListDataset, _Loader
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    gold = np.load(os.path.join(golden_dir, "tiny.npz"))
    mcfg = C.vit_tiny_lora(ltype, gs)
    cfg = make_cfg()
    cfg.TRAINER.GLP_OT_LORA.TYPE, cfg.TRAINER.GLP_OT_LORA.GLOBAL_S = ltype, gs
    data = SyntheticFedData(mcfg, 1, 1, 1, 8)
    batch = synth.make_batch(mcfg, 8, seed=1234)
    data.fed_train_loader_x_dict[0] = _Loader(_ListDataset([batch], ["race"], {"race": 3}))
    cfg.DATA = data
    cfg.MODEL.STATE_DICT = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    tr = build_trainer(cfg)
    sd = tr.model.state_dict()
    assert list(sd.keys()) == list(synth.manifest(mcfg).keys())     # make_golden asserts manifest == reference's keys
    assert sum(p.numel() for p in tr.model.parameters() if p.requires_grad) == meta[f"{tag}.trainable_elems"]
    k0 = "image_encoder.transformer.resblocks.0.mlp.c_fc."
    assert (k0 + "lora_S_global.weight" in sd) == gs and (k0 + "lora_S.weight" in sd) == (ltype != "LoRA")
    if ltype == "SVLoRA":
        assert tuple(sd[k0 + "lora_S.weight"].shape) == (4,)
    tr.num_batches = 10 ** 9
    for i, ref in enumerate(meta[f"{tag}.traj"]):
        tr.batch_idx = i
        s = tr.forward_backward(batch)
        assert abs(s["loss"] - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (i, s, ref)
        assert abs(s["acc"] - ref["acc"]) < 1e-3 and abs(s["auc"] - ref["auc"]) < 1e-9
    sd = tr.model.state_dict()
    for k in synth.trainable_keys(mcfg):
        assert rel(sd[k], gold[f"{tag}.post.{k}"]) < 1e-4, k


def test_trainer_round_api_and_lr_schedule(tmp_path):
    """train()/test()/state_dict round trip as federated_main.py uses them; the shared StepLR is stepped once per
    registered model name, i.e. twice per local epoch (Dassl/dassl/engine/trainer.py:253-258)."""
    from fairfedmed_amd.trainer import GLP_OT_SVLoRA, SyntheticFedData
    mcfg = C.vit_tiny(rank=4)
    cfg = make_cfg(prec="bf16")
    cfg.OUTPUT_DIR = str(tmp_path)
    cfg.TEST.NO_TEST = False
    data = SyntheticFedData(mcfg, num_clients=2, train_batches=3, test_batches=2, batch_size=8, signal=0.4)
    tr = GLP_OT_SVLoRA(cfg, data=data)
    tr.fed_before_train()
    global_weights = {k: v.clone() for k, v in tr.model.state_dict().items()}
    assert len(tr.fed_train_loader_x_dict[0].dataset) == 24
    lrs = []
    for rnd in range(3):
        for idx in (0, 1):
            tr.model.load_state_dict(global_weights, strict=False)
            tr.train(idx=idx, global_epoch=rnd, is_fed=True)
            lrs.append(tr.get_current_lr())
    # StepLR(step_size 2, gamma .1) advanced by 2 per client-epoch: one decade per client-epoch
    assert abs(lrs[0] - 1e-4) < 1e-12 and abs(lrs[1] - 1e-5) < 1e-13 and abs(lrs[-1] - 1e-9) < 1e-17
    assert tr.sched.last_epoch == 12
    res = tr.test(idx=0, current_epoch=0)
    assert len(res) == 4 and 0 <= res[3] <= 100 and abs(res[0] + res[1] - 100) < 1e-9
    assert res[3] > 1.0 or res[3] == 0.0                             # percent, like the reference's evaluator
    assert os.path.exists(os.path.join(str(tmp_path), "epoch2_client1.pth"))
    saved = torch.load(os.path.join(str(tmp_path), "epoch2_client1.pth"))
    assert "prompt_learner.ctx" in saved and "prompt_learner.token_prefix" in saved
    assert not any("original_linear" in k for k in saved)
    tr.fed_after_train()


def _fed_setup(users=2, rounds=2):
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData
    import fairfedmed_amd.trainer  # noqa: F401
    mcfg = C.vit_tiny(rank=4)
    cfg = make_cfg(bs=8)
    cfg.DATASET.USERS = users
    cfg.TEST.NO_TEST = True
    cfg.TRAIN.METRICS_EVERY = 0
    cfg.DATA = SyntheticFedData(mcfg, users, 2, 1, 8, signal=0.3)
    cfg.MODEL.STATE_DICT = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
    return build_trainer(cfg)


def test_federated_round_loop_single_process_and_ranks_agree():
    """fairfedmed_amd.federated: the FedOTPLoRA round loop (federated_main.py:604-726) on the HIP trainer; the
    one-process driver and the torch.distributed driver (world 1 here, flat all-reduce) reach the same weights."""
    import torch.distributed as dist
    from fairfedmed_amd import federated as F
    args = F.FedArgs(num_users=2, frac=1.0, round=2, shared_half_s=True, seed=0)
    tr = _fed_setup()
    init = {k: v.clone() for k, v in tr.model.state_dict().items() if k in set(tr.engine.params.keys)}
    hist = F.run_fedotplora(tr, args, log=lambda *_: None)
    assert len(hist["acc"]) == 2 and len(hist["auc"]) == 2 and all(np.isfinite(hist["acc"]))
    moved = sum(float((hist["global_weights"][k].cpu() - init[k].cpu()).abs().sum()) for k in init)
    assert moved > 0, "training did not change the global weights"
    # the same two rounds through the distributed driver
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        tr2 = _fed_setup()
        hist2 = F.run_fedotplora_ranks(tr2, args, log=lambda *_: None)
    finally:
        if created:
            dist.destroy_process_group()
    params = tr2.engine.params
    for k, v in hist["global_weights"].items():
        off, shp = params.offsets[k]
        got = hist2["global_flat"][off:off + v.numel()].view(shp).cpu()
        assert rel(got, v.cpu()) < 1e-5, k
    assert abs(hist["acc"][-1] - hist2["acc"][-1]) < 1e-6


# ----------------------------------------------------------------------------- RN50 backbone
def rn_cfg(bs=6):
    cfg = make_cfg(bs=bs)
    cfg.DATASET.ATTRIBUTES, cfg.DATASET.ATTRIBUTE_TYPE = ["gender"], "gender"        # two groups, as the golden model
    cfg.MODEL.GEOMETRY = C.rn_tiny(rank=4, num_groups=2)
    return cfg


def test_trainer_rn_backbone_trajectory(golden_dir):
    """The registry-built trainer picks the RN50 engine for a ResNet geometry; its steps follow the reference's
    forward_backward trajectory and state_dict() carries the live BatchNorm buffers."""
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.engine_rn import RN50Engine
    from fairfedmed_amd.trainer import SyntheticFedData, _ListDataset, _Loader
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    gold = np.load(os.path.join(golden_dir, "rn_tiny.npz"))
    mcfg = C.rn_tiny(rank=4, num_groups=2)
    cfg = rn_cfg()
    data = SyntheticFedData(mcfg, 1, 1, 1, 6)
    batch = synth.make_batch(mcfg, 6, seed=1234)
    data.fed_train_loader_x_dict[0] = _Loader(_ListDataset([batch], ["gender"], {"gender": 2}))
    cfg.DATA = data
    cfg.MODEL.STATE_DICT = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    tr = build_trainer(cfg)
    assert isinstance(tr.engine, RN50Engine)
    image, label, _, attr = tr.parse_batch_train(batch)
    tr.engine.forward_backward(image, attr, label)        # the golden script's gradient pass: one BatchNorm update
    tr.num_batches = 10 ** 9
    for i, ref in enumerate(meta["rn_tiny_r4g2.traj"]):
        tr.batch_idx = i
        s = tr.forward_backward(batch)
        assert abs(s["loss"] - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (s, ref)
        assert abs(s["acc"] - ref["acc"]) < 1e-3 and abs(s["auc"] - ref["auc"]) < 1e-9
    sd = tr.model.state_dict()
    for k in synth.buffer_keys(mcfg):
        ref = torch.from_numpy(gold[f"rn_tiny_r4g2.post.{k}"]).double()
        assert float((sd[k].cpu().double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-7, k
    # buffers round-trip through load_state_dict without rebuilding the frozen weights
    before = tr.engine.rnw["s1"].data_ptr()
    zero = {k: torch.zeros_like(v) for k, v in sd.items() if k.endswith(("running_mean", "num_batches_tracked"))}
    tr.model.load_state_dict(zero, strict=False)
    assert float(tr.engine.bns[0].run_mean.abs().max()) == 0.0 and int(tr.engine.nbt.sum()) == 0
    assert tr.engine.rnw["s1"].data_ptr() == before


def test_federated_rn_backbone_averages_batchnorm_buffers():
    """Two clients, two rounds on the RN backbone: the BatchNorm running statistics travel with the trainable
    tensors (the reference averages the whole state_dict), one-process and torch.distributed drivers agree."""
    import torch.distributed as dist
    from fairfedmed_amd import federated as F
    from fairfedmed_amd.registry import build_trainer
    from fairfedmed_amd.trainer import SyntheticFedData
    mcfg = C.rn_tiny(rank=4, num_groups=2)

    def setup():
        cfg = rn_cfg()
        cfg.DATASET.USERS = 2
        cfg.TEST.NO_TEST = True
        cfg.TRAIN.METRICS_EVERY = 0
        cfg.DATA = SyntheticFedData(mcfg, 2, 2, 1, 6, attribute="gender", signal=0.3)
        cfg.MODEL.STATE_DICT = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
        return build_trainer(cfg)

    args = F.FedArgs(num_users=2, frac=1.0, round=2, shared_half_s=True, seed=0)
    tr = setup()
    hist = F.run_fedotplora(tr, args, log=lambda *_: None)
    gw = hist["global_weights"]
    bkeys = synth.buffer_keys(mcfg)
    assert all(k in gw for k in bkeys)
    k0 = "image_encoder.bn1.running_mean"
    assert float(gw[k0].abs().max()) > 0                          # synthetic init is 0 mean / 1 var: it moved
    # counters too (as floats, as in the reference): round 0 averages 2 and 2, round 1 averages 4 and 4 and mixes the
    # previous global 2 back in with beta = 0.999 * 1/2
    assert abs(float(gw["image_encoder.bn1.num_batches_tracked"]) - (4 * (1 - 0.4995) + 2 * 0.4995)) < 1e-5
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        tr2 = setup()
        hist2 = F.run_fedotplora_ranks(tr2, args, log=lambda *_: None)
    finally:
        if created:
            dist.destroy_process_group()
    params = tr2.engine.params
    for k, v in gw.items():
        if k in params.offsets:
            off, shp = params.offsets[k]
            got = hist2["global_flat"][off:off + v.numel()].view(shp).cpu()
            assert rel(got, v.cpu()) < 1e-5, k
    live = tr2.engine.buffer_state()                               # loaded from hist2["global_buffers"] at the end
    for k in bkeys:
        if not k.endswith("num_batches_tracked"):
            assert rel(live[k].cpu(), gw[k].cpu()) < 1e-5, k
    assert abs(hist["acc"][-1] - hist2["acc"][-1]) < 1e-6


# ----------------------------------------------------------------------------- on-disk data path
def test_trainer_on_fairfedmed_files_uint8_equals_float32_transport(tmp_path):
    """Two clients read from a FairFedMed tree on disk; shipping uint8 samples (expanded on the GPU) trains to exactly
    the weights that shipping the reference's float32 batches does."""
    from fairfedmed_amd import data as D
    from fairfedmed_amd import federated as F
    from fairfedmed_amd.registry import build_trainer
    import fairfedmed_amd.trainer  # noqa: F401  (registers GLP_OT_SVLoRA)
    D.write_synthetic_fairfedmed(str(tmp_path), sites=2, n_train=16, n_test=8, size=64, seed=4)
    mcfg = C.vit_tiny(rank=4)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="reference")
    hist = {}
    for transport in ("float32", "uint8"):
        cfg = make_cfg(bs=8)
        cfg.DATASET = NS(NAME="FairFedMed", ROOT=str(tmp_path), USERS=2, ATTRIBUTE_TYPE="race",
                         ATTRIBUTES=["race", "gender"], MODALITY_TYPE="slo_fundus")
        cfg.TEST.NO_TEST = True
        cfg.DATA = D.FedData(cfg, transport=transport)
        cfg.MODEL.STATE_DICT = sd
        tr = build_trainer(cfg)
        first = next(iter(tr.fed_train_loader_x_dict[0]))
        assert first["img"].dtype == (torch.uint8 if transport == "uint8" else torch.float32)
        hist[transport] = F.run_fedotplora(tr, F.FedArgs(num_users=2, frac=1.0, round=2, shared_half_s=True, seed=0),
                                           log=lambda *_: None)
    a, b = hist["float32"], hist["uint8"]
    for k, v in a["global_weights"].items():
        assert torch.equal(v, b["global_weights"][k]), k
    assert a["acc"] == b["acc"] and a["auc"] == b["auc"]


def test_cli_runs_the_fairlora_script_on_files(tmp_path):
    """python -m fairfedmed_amd.federated_main with the flags of scripts/fairfedlora_fairfedmed.sh on a FairFedMed tree
    on disk (reduced geometry through cfg_hook): rounds run, per-client final weights are written."""
    from fairfedmed_amd import data as D
    from fairfedmed_amd import federated_main as FM
    D.write_synthetic_fairfedmed(str(tmp_path / "DATA"), sites=3, n_train=16, n_test=8, size=64, seed=9,
                                 attribute_type="language")
    (tmp_path / "tr.yaml").write_text('DATALOADER:\n  TRAIN_X:\n    BATCH_SIZE: 8\n  TEST:\n    BATCH_SIZE: 8\n'
                                      'INPUT:\n  SIZE: (64, 64)\nMODEL:\n  BACKBONE:\n    NAME: "tiny"\n')
    out = tmp_path / "out"
    argv = ["--root", str(tmp_path / "DATA"), "--model", "FedOTPLoRA", "--seed", "1", "--num_users", "3", "--frac", "0.8",
            "--lr", "0.001", "--OT", "None", "--gamma", "0.1", "--trainer", "GLP_OT_SVLoRA", "--round", "2",
            "--stepsize", "200", "--attribute_type", "language", "--attributes", "language", "race", "gender",
            "--n_ctx", "4", "--num_prompt", "2", "--unfreeze_image_encoder", "True", "--lora_rank", "4",
            "--lora_alpha", "2", "--lora_type", "FairLoRA", "--config-file", str(tmp_path / "tr.yaml"),
            "--output-dir", str(out), "--shared_half_s", "True", "--prec", "fp32"]
    lines = []

    def hook(cfg):
        cfg.MODEL.GEOMETRY = C.vit_tiny(rank=4)

    hist = FM.main(argv, log=lambda *a: lines.append(" ".join(str(x) for x in a)), cfg_hook=hook)
    assert len(hist["acc"]) == 2 and all(np.isfinite(hist["auc"]))
    for idx in range(3):
        w = torch.load(out / f"global_client{idx}_final.pth")
        assert "prompt_learner.ctx" in w and any("lora_S" in k for k in w)
        # the reference saves every client's FULL state_dict (federated_main.py:771-774): same keys, same order
        assert list(w.keys()) == list(synth.manifest(C.vit_tiny(rank=4)).keys())
    assert any("Global test acc" in ln for ln in lines) and any("maximum test acc" in ln for ln in lines)
    # --eval-only --model-dir: evaluate every client with weights loaded through the trainer's load_model hook
    from fairfedmed_amd.registry import build_trainer  # noqa: F401
    cfg_seen = {}

    def hook2(cfg):
        hook(cfg)
        cfg_seen["cfg"] = cfg
    FM.main(argv + ["--round", "1"], log=lambda *a: None, cfg_hook=hook2)
    # save a checkpoint from a trainer built the same way, then evaluate it through the command line
    import fairfedmed_amd.trainer as T
    tr = T.GLP_OT_SVLoRA(cfg_seen["cfg"])
    tr.save_model(0, str(tmp_path / "ckpt"), is_best=True)
    ev = FM.main(argv + ["--eval-only", "--model-dir", str(tmp_path / "ckpt")], log=lambda *a: lines.append(" ".join(map(str, a))),
                 cfg_hook=hook)
    assert sorted(ev["eval"]) == [0, 1, 2] and all(len(r) == 4 for r in ev["eval"].values())
    assert any(ln.startswith("client 2: acc") for ln in lines)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_svlora_linear_vs_reference_golden(golden_dir, dtype):
    """SVLoRALinear (--lora_type SVLoRA; trainers/GLP_OT_SVLoRA.py:255-330) on the HIP kernels vs the imported
    reference layer: initial singular values (1-D, linspace(1, 0.1, r)), forward, all gradients."""
    from fairfedmed_amd.model import SVLoRALinear, apply_lora_to_model
    from tests.golden.make_golden import rng_tensor
    gold = np.load(os.path.join(golden_dir, "svlora.npz"))
    L, Bn, fin, fout, r = 50, 4, 128, 192, 8
    lin = torch.nn.Linear(fin, fout)
    lin.weight.data = rng_tensor("svlora.W", (fout, fin)) * fin ** -0.5
    lin.bias.data = rng_tensor("svlora.b", (fout,)) * 0.1
    layer = SVLoRALinear(lin.cuda(), rank=r, alpha=2.0)
    assert tuple(layer.lora_S.weight.shape) == tuple(gold["svlora.s_shape"]) == (r,)
    assert rel(layer.lora_S.weight.detach(), gold["svlora.s_init"]) < 1e-7
    assert float(layer.lora_A.weight.abs().max()) == 0.0
    layer.lora_A.weight.data.copy_(rng_tensor("svlora.A", (fin, r)) * 0.1)
    layer.lora_B.weight.data.copy_(rng_tensor("svlora.B", (r, fout)))
    layer.lora_S.weight.data.copy_(torch.from_numpy(gold["svlora.S"]))
    xin = rng_tensor("svlora.x", (L, Bn, fin)).cuda().to(dtype).requires_grad_(True)
    y = layer(xin)
    y.backward(rng_tensor("svlora.g", (L, Bn, fout)).cuda().to(dtype))
    t1, t2 = (3e-5, 1e-4) if dtype == torch.float32 else (1.5e-2, 4e-2)
    assert rel(y.detach().float().cpu(), gold["svlora.y"]) < t1
    assert rel(xin.grad.float().cpu(), gold["svlora.dx"]) < t1
    assert rel(layer.lora_A.weight.grad, gold["svlora.dA"]) < t2
    assert rel(layer.lora_B.weight.grad, gold["svlora.dB"]) < t2
    assert rel(layer.lora_S.weight.grad, gold["svlora.dS"]) < t2
    # the injection rule knows the three types on the ViT branch
    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mlp = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Linear(128, 64))
    class Mdl(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.image_encoder = torch.nn.Sequential(Blk())
    for typ, cls in (("SVLoRA", "SVLoRALinear"), ("LoRA", "LoRALinear"), ("FairLoRA", "FairLoRALinear")):
        m = Mdl()
        apply_lora_to_model(m, True, rank=4, alpha=2.0, lora_type=typ, num_attrs=3)
        assert type(m.image_encoder[0].mlp[0]).__name__ == cls
    with pytest.raises(NotImplementedError):
        apply_lora_to_model(Mdl(), True, lora_type="DoRA")


def test_cli_under_torch_distributed_run_two_ranks(tmp_path):
    """The command line launched as `python -m torch.distributed.run --nproc-per-node 2 -m fairfedmed_amd.federated_main`
    (one client per rank; here both ranks share the test box's GPU over gloo, FFM_ONE_DEVICE=1): rounds run, ranks agree,
    rank 0 writes global_client{idx}_final.pth."""
    import subprocess
    import sys
    from fairfedmed_amd import data as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    D.write_synthetic_fairfedmed(str(tmp_path / "DATA"), sites=2, n_train=16, n_test=8, size=64, seed=9)
    (tmp_path / "tr.yaml").write_text('DATALOADER:\n  TRAIN_X:\n    BATCH_SIZE: 8\n  TEST:\n    BATCH_SIZE: 8\n'
                                      'INPUT:\n  SIZE: (64, 64)\nMODEL:\n  BACKBONE:\n    NAME: "tiny"\n')
    out = tmp_path / "out"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29518", "-m", "fairfedmed_amd.federated_main",
           "--root", str(tmp_path / "DATA"), "--model", "FedOTPLoRA", "--trainer", "GLP_OT_SVLoRA", "--num_users", "2",
           "--frac", "1.0", "--round", "2", "--OT", "None", "--attribute_type", "race", "--attributes", "race", "gender",
           "--n_ctx", "4", "--num_prompt", "2", "--unfreeze_image_encoder", "True", "--lora_rank", "4", "--lora_alpha", "2",
           "--lora_type", "FairLoRA", "--shared_half_s", "True", "--config-file", str(tmp_path / "tr.yaml"),
           "--output-dir", str(out), "--prec", "fp32"]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, FFM_ONE_DEVICE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "global_test_acc_list:" in r.stdout and "round 1:" in r.stdout
    # rank 0 writes the reference's per-client files here too (full state_dicts rebuilt from the flat buffers)
    for idx in range(2):
        w = torch.load(out / f"global_client{idx}_final.pth")
        assert list(w.keys()) == list(synth.manifest(C.vit_tiny(rank=4)).keys())
        assert all(bool(torch.isfinite(v.float()).all()) for v in w.values())
    assert not (out / "global_flat_final.pth").exists()


def test_cli_rank_path_over_rccl_world_one_equals_the_single_process_driver(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1 -m fairfedmed_amd.federated_main ...` WITHOUT the gloo test rig:
    the rank-parallel driver initialises the nccl backend (RCCL) on this box's GPU, its round-boundary all-reduces run
    through an RCCL communicator, and with one rank the saved per-client weights must equal those of the one-process
    driver (same clients, same order, same optimizer state)."""
    import subprocess
    import sys
    from fairfedmed_amd import data as D
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    D.write_synthetic_fairfedmed(str(tmp_path / "DATA"), sites=2, n_train=16, n_test=8, size=64, seed=9)
    (tmp_path / "tr.yaml").write_text('DATALOADER:\n  TRAIN_X:\n    BATCH_SIZE: 8\n  TEST:\n    BATCH_SIZE: 8\n'
                                      'INPUT:\n  SIZE: (64, 64)\nMODEL:\n  BACKBONE:\n    NAME: "tiny"\n')
    flags = ["--root", str(tmp_path / "DATA"), "--model", "FedOTPLoRA", "--trainer", "GLP_OT_SVLoRA", "--num_users", "2",
             "--frac", "1.0", "--round", "2", "--OT", "None", "--attribute_type", "race", "--attributes", "race", "gender",
             "--n_ctx", "4", "--num_prompt", "2", "--unfreeze_image_encoder", "True", "--lora_rank", "4", "--lora_alpha", "2",
             "--lora_type", "FairLoRA", "--shared_half_s", "True", "--config-file", str(tmp_path / "tr.yaml"),
             "--prec", "fp32", "--save-trainable-only", "--compat-sequential-optimizer"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FFM_ONE_DEVICE")}
    probe = ("import torch.distributed as d, fairfedmed_amd.federated as F; o = F.run_fedotplora_ranks\n"
             "def w(*a, **k):\n    print('BACKEND', d.get_backend(), d.get_world_size(), flush=True); return o(*a, **k)\n"
             "F.run_fedotplora_ranks = w\n"
             "import sys; from fairfedmed_amd.federated_main import main; main(sys.argv[1:])")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29519", "--no-python", sys.executable, "-c", probe] + flags + ["--output-dir", str(tmp_path / "ranks")]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "BACKEND nccl 1" in r.stdout and "global_test_acc_list:" in r.stdout
    r1 = subprocess.run([sys.executable, "-m", "fairfedmed_amd.federated_main"] + flags + ["--output-dir", str(tmp_path / "one")],
                        cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, (r1.stdout[-1500:], r1.stderr[-3000:])
    for idx in range(2):
        a = torch.load(tmp_path / "ranks" / f"global_client{idx}_final.pth")
        b = torch.load(tmp_path / "one" / f"global_client{idx}_final.pth")
        assert "prompt_learner.ctx" in a and set(a) <= set(b)           # (the rank driver keeps the trainable tensors)
        for k in a:
            assert torch.allclose(a[k].float(), b[k].float(), rtol=1e-5, atol=1e-7), k


def test_rn_state_dict_keys_order_dtypes_and_live_buffers():
    """CustomCLIP over the RN50 engine: state_dict() has the reference ResNet's keys in ITS order (tests/golden/
    make_golden.py asserts manifest order == the imported reference's), BatchNorm counters are int64, running
    statistics are live views of the engine's buffers."""
    from fairfedmed_amd.model import CustomCLIP
    mcfg = C.rn_tiny(rank=4, num_groups=2)
    sd = synth.make_state_dict(mcfg, seed=1, lora_init="random")
    model = CustomCLIP(mcfg, sd, dtype=torch.float32, max_images=6)
    got = model.state_dict()
    man = synth.manifest(mcfg)
    assert list(got.keys()) == list(man.keys())
    for k, shp in man.items():
        assert tuple(got[k].shape) == tuple(shp), k
        assert got[k].dtype == (torch.int64 if k.endswith("num_batches_tracked") else torch.float32), k
    assert {n for n, p in model.named_parameters() if p.requires_grad} == set(synth.trainable_keys(mcfg))
    assert {n for n, _ in model.named_buffers()} >= set(synth.buffer_keys(mcfg))
    b = synth.make_batch(mcfg, 6, seed=3)
    model.engine.forward_backward(b["img"].cuda(), b["attrs"].t()[0].cuda(), b["label"].cuda())
    after = model.state_dict()
    k = "image_encoder.layer1.0.bn1.running_mean"
    assert float(after[k].abs().max()) > 0 and int(after["image_encoder.bn1.num_batches_tracked"]) == 1


def test_save_model_load_model_round_trip(tmp_path):
    """Dassl's checkpoint layout (`<dir>/<name>/model.pth.tar-<epoch>`, `model-best.pth.tar`) and the trainer's
    load_model hook (trainers/GLP_OT_SVLoRA.py:1023-1053): a trained trainer's weights come back in a fresh one and
    give the same logits; the fixed token vectors are not taken from the file; a missing file raises."""
    from fairfedmed_amd.trainer import GLP_OT_SVLoRA, SyntheticFedData
    mcfg = C.vit_tiny(rank=4)
    data = SyntheticFedData(mcfg, 1, 3, 1, 8, signal=0.3)
    tr = GLP_OT_SVLoRA(make_cfg(prec="fp32"), data=data)
    tr.fed_before_train()
    tr.train(idx=0, global_epoch=0, is_fed=True)
    tr.save_model(4, str(tmp_path), is_best=True)
    assert os.path.exists(tmp_path / "prompt_learner" / "model.pth.tar-5") and os.path.exists(tmp_path / "image_encoder" / "model-best.pth.tar")
    batch = synth.make_batch(mcfg, 8, seed=99)
    img, attr = batch["img"].cuda(), batch["attrs"].t()[0].cuda()
    want = tr.model_inference(img, attr).clone()
    fresh = GLP_OT_SVLoRA(make_cfg(prec="fp32"), data=data)
    assert not torch.equal(fresh.model_inference(img, attr), want)
    bad = torch.load(tmp_path / "prompt_learner" / "model-best.pth.tar")
    bad["state_dict"]["token_prefix"] = torch.full_like(bad["state_dict"]["token_prefix"], 7.0)    # must be ignored
    torch.save(bad, tmp_path / "prompt_learner" / "model-best.pth.tar")
    fresh.load_model(str(tmp_path))
    assert torch.equal(fresh.model_inference(img, attr), want)
    fresh.load_model(str(tmp_path), epoch=5)
    assert torch.equal(fresh.model_inference(img, attr), want)
    with pytest.raises(FileNotFoundError):
        fresh.load_model(str(tmp_path), epoch=9)
    fresh.load_model("")                                              # no directory: skipped, as in the reference


@pytest.mark.parametrize("tag,mk,names,attribute,dataset", [
    ("adapter_vit", lambda: C.vit_tiny(rank=4), ["NOT Glaucoma", "Glaucoma"], "race", "FairFedMed"),
    ("adapter_rn", lambda: C.rn_tiny(rank=4, num_groups=2), ["NOT Pleural Effusion", "Pleural Effusion"], "gender", "FedChexMimic")])
def test_custom_clip_reference_constructor(golden_dir, tag, mk, names, attribute, dataset):
    """CustomCLIP(cfg, classnames, clip_model) followed by apply_lora_to_model(model, ...) - the reference's call
    sequence (trainers/GLP_OT_SVLoRA.py:575-613, 820, 833-841) - on a CLIP-shaped model: state_dict contract, prompt
    buffers and, with the constructor's random draws replaced by known tensors, the reference's eval logits."""
    from fairfedmed_amd.model import CustomCLIP, apply_lora_to_model
    from tests.test_host_cpu import _ref_style_cfg
    unit = np.load(os.path.join(golden_dir, "unit.npz"))
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))
    mcfg = mk()
    cfg = _ref_style_cfg(mcfg, attribute, dataset)
    model = CustomCLIP(cfg, names, synth.make_clip_model(mcfg, seed=1))
    assert model.dtype == torch.float32 and model.engine.max_images == 6 and model.n_cls == 2
    apply_lora_to_model(model, True, rank=4, alpha=2.0, lora_type="FairLoRA", global_s=False, num_attrs=mcfg.lora.num_groups)
    with pytest.raises(ValueError, match="was built with"):
        apply_lora_to_model(model, True, rank=8, alpha=2.0, lora_type="FairLoRA", num_attrs=mcfg.lora.num_groups)
    sd = model.state_dict()
    assert list(sd.keys()) == meta[f"{tag}.keys"]
    assert np.array_equal(model.tokenized_prompts.numpy(), unit[f"{tag}.tokens"])
    assert np.array_equal(sd["prompt_learner.token_suffix"].cpu().numpy(), unit[f"{tag}.token_suffix"])
    assert {n for n, p in model.named_parameters() if p.requires_grad} == set(synth.trainable_keys(mcfg))
    known = synth.make_state_dict(mcfg, seed=3, lora_init="random")
    model.load_state_dict({k: known[k] for k in synth.trainable_keys(mcfg) if not synth._is_bn_param(k)}, strict=False)
    batch = synth.make_batch(mcfg, 6, seed=77)
    logits = model(batch["img"].cuda(), batch["attrs"].t()[0].cuda())
    assert rel(logits, unit[f"{tag}.logits"]) < 3e-5, rel(logits, unit[f"{tag}.logits"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_fairlora_linear_rank16_wide_output_ds_sums_only_written_partials(dtype):
    """FairLoRALinear.backward at rank 16, 3072 outputs, >= 1024 rows in 16-bit storage: the matrix-core down projection
    writes ceil(M / 16) dS partial rows; the sum over them must not reach into unwritten memory (the allocator's cached
    blocks are filled with NaN in front of the call) and has to match float64 autograd of the reference's formula
    (trainers/GLP_OT_SVLoRA.py:450-482)."""
    from fairfedmed_amd.model import FairLoRALinear
    from fairfedmed_amd import ops
    L, Bn, fin, fout, r, G = 300, 4, 768, 3072, 16, 3
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(L, Bn, fin, generator=gen)
    g = torch.randn(L, Bn, fout, generator=gen) * 0.1
    attr = torch.randint(0, G, (Bn,), generator=gen)
    lin = torch.nn.Linear(fin, fout)
    layer = FairLoRALinear(lin.cuda(), rank=r, alpha=2.0, num_attrs=G)
    A, Sm, Bm = torch.randn(fin, r, generator=gen) * 0.05, torch.rand(G, r, generator=gen), torch.randn(r, fout, generator=gen) * 0.05
    layer.lora_A.weight.data.copy_(A)
    layer.lora_S.weight.data.copy_(Sm)
    layer.lora_B.weight.data.copy_(Bm)
    xin = x.cuda().to(dtype).requires_grad_(True)
    y = layer(xin, attr.cuda())
    nb = ops.lora_down_blocks(L * Bn, fout, r, dtype)
    assert nb == (L * Bn + 15) // 16 < ops.lora_down_blocks_max(L * Bn, fout, r, dtype)
    for n in (nb, ops.lora_down_blocks_max(L * Bn, fout, r, dtype)):     # poison what torch.empty may hand out next
        junk = [torch.full((n, G, r), float("nan"), device="cuda") for _ in range(3)]
        del junk
    y.backward(g.cuda().to(dtype))
    # float64 autograd of the reference's forward on the 16-bit-rounded operands
    xd = x.to(dtype).double()
    Ad, Sd, Bd = (t.double().requires_grad_(True) for t in (A, Sm, Bm))
    pi = torch.full((Bn, G), 0.3 / (G - 1), dtype=torch.float64)
    pi[torch.arange(Bn), attr] = 0.7
    s = pi @ Sd
    yd = xd @ lin.weight.detach().cpu().double().t() + lin.bias.detach().cpu().double() + (2.0 / r) * (((xd @ Ad) * s[None]) @ Bd)
    yd.backward(g.to(dtype).double())
    assert bool(torch.isfinite(layer.lora_S.weight.grad).all())
    assert rel(y.detach().float(), yd.detach()) < 1.5e-2
    assert rel(layer.lora_S.weight.grad, Sd.grad) < 4e-2
    assert rel(layer.lora_A.weight.grad, Ad.grad) < 4e-2
    assert rel(layer.lora_B.weight.grad, Bd.grad) < 4e-2
