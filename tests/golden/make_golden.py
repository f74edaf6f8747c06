#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference).  Nothing here
travels as reference source: the outputs are data (inputs are regenerated from
seeds by fairfedmed_amd.synth; expected outputs are stored as small .npz/.json).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--vitb] [--time-ref]

The reference cannot be imported unmodified: third-party
modules absent from the image are served as stubs by a meta-path finder, and
``Dassl.dassl.engine`` must be imported before the trainer module.
"""
from __future__ import annotations

import argparse
import importlib.abc
import importlib.machinery
import json
import os
import sys
import time
import types
from types import SimpleNamespace as NS
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

# ---------------------------------------------------------------- stubs ----
PREFIXES = ("torchvision", "ftfy", "tensorboard", "torch.utils.tensorboard", "gdown", "fairlearn",
            "aif360", "prettytable", "yacs", "timm", "skimage", "datasets", "cv2")


class _Meta(type):
    def __getattr__(cls, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return MagicMock(name=f"{cls.__name__}.{n}")


class _Stub(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        o = _Meta(n, (), {"__init__": lambda s, *a, **k: None, "__call__": lambda s, *a, **k: None})
        setattr(self, n, o)
        return o


class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if any(name == p or name.startswith(p + ".") for p in PREFIXES):
            return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)


def import_reference():
    sys.meta_path.insert(0, _Finder())
    for k in [k for k in sys.modules if k == "datasets" or k.startswith("datasets.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import ftfy
    ftfy.fix_text = lambda s: s
    from Dassl.dassl.engine import build_trainer  # noqa: F401  (must precede the trainer import)
    import trainers.GLP_OT_SVLoRA as M
    from clip.model import CLIP
    import utils.fed_utils as FU
    from evaluation.metrics import compute_auc
    return M, CLIP, FU, compute_auc


# ------------------------------------------------------------- helpers ----
from fairfedmed_amd import config as C      # noqa: E402
from fairfedmed_amd import synth            # noqa: E402


def ref_cfg(mcfg: C.ModelCfg, lambda_fairness=0.0):
    ot = NS(EPS=mcfg.ot_eps, THRESH=mcfg.ot_thresh, OT=mcfg.ot, TOP_PERCENT=mcfg.ot_top_percent, MAX_ITER=mcfg.ot_max_iter)
    return NS(
        INPUT=NS(PIXEL_MEAN=list(mcfg.pixel_mean), PIXEL_STD=list(mcfg.pixel_std),
                 SIZE=(mcfg.vision.image_size, mcfg.vision.image_size)),
        DATASET=NS(NAME="FairFedMed", MODALITY_TYPE="slo_fundus" if not mcfg.dim_per_3d_slice else "oct_bscans",
                   DIM_PER_3D_SLICE=mcfg.dim_per_3d_slice, ATTRIBUTES=["race"], ATTRIBUTE_TYPE="race"),
        TRAINER=NS(GLP_OT=NS(N_CTX=mcfg.n_ctx, CTX_INIT=False, CSC=False, N=mcfg.n_prompts,
                             CLASS_TOKEN_POSITION="end", EPS=ot.EPS, THRESH=ot.THRESH, OT=ot.OT,
                             TOP_PERCENT=ot.TOP_PERCENT, MAX_ITER=ot.MAX_ITER, PREC="fp32"),
                   GLP_OT_LORA=NS(DISABLE_ATTR=False),
                   LAMBDA_FAIRNESS=lambda_fairness),
    )


def build_reference_model(M, CLIP, mcfg: C.ModelCfg, sd):
    v, t = mcfg.vision, mcfg.text
    dd = {"trainer": "GLP_OT", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0}
    is_rn = isinstance(v, C.ResNetCfg)
    clip_model = CLIP(v.out_dim, v.image_size, tuple(v.layers) if is_rn else v.layers, v.width,
                      None if is_rn else v.patch, t.context_length, 49408, t.width, t.heads, t.layers, dd).float()
    model = M.CustomCLIP(ref_cfg(mcfg), ["NOT Glaucoma", "Glaucoma"], clip_model)
    bn_params = {id(p) for mod in model.modules() if isinstance(mod, torch.nn.BatchNorm2d) for p in mod.parameters()}
    for n, p in model.named_parameters():
        # the freeze loop of GLP_OT_SVLoRA.build_model (:822-829): prompts, the 3D conv and BatchNorm2d stay trainable
        p.requires_grad_("prompt_learner" in n or "proj_per_3d_slice" in n or id(p) in bn_params)
    M.apply_lora_to_model(model, True, rank=mcfg.lora.rank, alpha=mcfg.lora.alpha, lora_type=mcfg.lora.lora_type,
                          global_s=mcfg.lora.global_s, num_attrs=mcfg.lora.num_groups)
    ref_sd = model.state_dict()
    man = synth.manifest(mcfg)
    assert list(ref_sd.keys()) == list(man.keys()), (
        [k for k in ref_sd if k not in man], [k for k in man if k not in ref_sd])
    for k, shp in man.items():
        assert tuple(ref_sd[k].shape) == tuple(shp), (k, ref_sd[k].shape, shp)
    eot = model.tokenized_prompts.argmax(-1).tolist()
    assert eot == list(mcfg.eot) * mcfg.n_prompts, eot
    model.load_state_dict(sd, strict=True)
    return model


def sub(x: torch.Tensor, n=4096):
    """Strided subsample of a flattened tensor (<= n entries) for compact fixtures."""
    f = x.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().astype(np.float32)


def rng_tensor(tag: str, shape, scale=1.0):
    g = synth._rng(tag, 7)
    return torch.from_numpy(g.standard_normal(shape, dtype=np.float32) * np.float32(scale))


# ------------------------------------------------------------ goldens -----
LAYER_CASES = [
    # name, L, Bn(=b*S), in, out, r, G, S, conv_hw
    ("fc_small", 5, 16, 64, 256, 4, 3, 1, None),
    ("fc_vitb", 197, 4, 768, 3072, 8, 3, 1, None),
    ("proj_vitb_slices", 197, 4, 3072, 768, 16, 3, 2, None),
    ("conv1x1", 49, 4, 64, 256, 8, 2, 1, (7, 7)),
]


def layer_inputs(name, L, Bn, fin, fout, r, G, S, hw):
    b = Bn // S
    x = rng_tensor(name + ".x", (L, Bn, fin))
    g = rng_tensor(name + ".g", (L, Bn, fout))
    W = rng_tensor(name + ".W", (fout, fin), fin ** -0.5)
    bias = None if hw else rng_tensor(name + ".b", (fout,), 0.1)
    A = rng_tensor(name + ".A", (fin, r), 0.1)
    Sm = synth.lora_s_init(r, G) + rng_tensor(name + ".S", (G, r), 0.05)
    Bm = rng_tensor(name + ".B", (r, fout))
    attr = torch.from_numpy(synth._rng(name + ".attr", 7).integers(0, G, size=(b,), dtype=np.int64))
    return x, g, W, bias, A, Sm, Bm, attr


def golden_layers(M, out):
    import torch.nn as nn
    for case in LAYER_CASES:
        name, L, Bn, fin, fout, r, G, S, hw = case
        x, g, W, bias, A, Sm, Bm, attr = layer_inputs(*case)
        if hw:
            lin = nn.Conv2d(fin, fout, 1, bias=False)
            lin.weight.data = W.reshape(fout, fin, 1, 1).clone()
            xin = x.permute(1, 2, 0).reshape(Bn, fin, hw[0], hw[1]).clone().requires_grad_(True)
        else:
            lin = nn.Linear(fin, fout)
            lin.weight.data = W.clone()
            lin.bias.data = bias.clone()
            xin = x.clone().requires_grad_(True)
        layer = M.FairLoRALinear(lin, rank=r, alpha=2.0, num_attrs=G)
        layer.lora_A.weight.data = A.clone()
        layer.lora_S.weight.data = Sm.clone()
        layer.lora_B.weight.data = Bm.clone()
        y = layer(xin, attr)
        if hw:
            gy = g.reshape(hw[0], hw[1], Bn, fout).permute(2, 3, 0, 1)
        else:
            gy = g
        y.backward(gy)
        y_tok = y.detach().reshape(Bn, fout, -1).permute(2, 0, 1) if hw else y.detach()
        dx_tok = xin.grad.reshape(Bn, fin, -1).permute(2, 0, 1) if hw else xin.grad
        full = y_tok.numel() <= 65536
        out[f"layer.{name}.y"] = y_tok.numpy() if full else sub(y_tok)
        out[f"layer.{name}.dx"] = dx_tok.numpy() if dx_tok.numel() <= 65536 else sub(dx_tok)
        out[f"layer.{name}.dA"] = layer.lora_A.weight.grad.numpy()
        out[f"layer.{name}.dS"] = layer.lora_S.weight.grad.numpy()
        out[f"layer.{name}.dB"] = layer.lora_B.weight.grad.numpy()
        print("layer", name, "y", tuple(y.shape), "|dA|", float(layer.lora_A.weight.grad.norm()))
        if name == "fc_small":
            # FairLoRALinear.weight(x, attr) (:425-445; plain one-hot mix) with and without the attribute
            out[f"layer.{name}.weight_attr"] = layer.weight(xin.detach(), attr).detach().numpy()
            out[f"layer.{name}.weight_noattr"] = layer.weight(xin.detach(), None).detach().numpy()
            # GLOBAL_S: one more trainable vector added to every sample's singular values (:359-363, 418-422, 467-468)
            lg = M.FairLoRALinear(lin, rank=r, alpha=2.0, global_s=True, num_attrs=G)
            out[f"layer.{name}.gs.sg_init"] = lg.lora_S_global.weight.detach().numpy().copy()
            lg.lora_A.weight.data, lg.lora_S.weight.data, lg.lora_B.weight.data = A.clone(), Sm.clone(), Bm.clone()
            lg.lora_S_global.weight.data = lg.lora_S_global.weight.data * (1.0 + 0.1 * rng_tensor(name + ".Sg", (r,)))
            out[f"layer.{name}.gs.Sg"] = lg.lora_S_global.weight.detach().numpy().copy()
            x2 = x.clone().requires_grad_(True)
            y2 = lg(x2, attr)
            y2.backward(g)
            out[f"layer.{name}.gs.y"] = y2.detach().numpy()
            out[f"layer.{name}.gs.dx"] = x2.grad.numpy()
            for nm in ("A", "S", "B", "S_global"):
                out[f"layer.{name}.gs.d{nm}"] = getattr(lg, "lora_" + nm).weight.grad.numpy()
            out[f"layer.{name}.gs.weight_attr"] = lg.weight(x2.detach(), attr).detach().numpy()


def golden_lora_plain(M, out):
    """LoRALinear (plain LoRA, the RN50 attention-pool projections): forward, input/LoRA gradients, dense weight()."""
    import torch.nn as nn
    L, Bn, fin, fout, r = 50, 4, 128, 192, 8
    x = rng_tensor("lora_plain.x", (L, Bn, fin))
    g = rng_tensor("lora_plain.g", (L, Bn, fout))
    lin = nn.Linear(fin, fout)
    lin.weight.data = rng_tensor("lora_plain.W", (fout, fin)) * fin ** -0.5
    lin.bias.data = rng_tensor("lora_plain.b", (fout,)) * 0.1
    layer = M.LoRALinear(lin, rank=r, alpha=2.0)
    layer.lora_A.weight.data = rng_tensor("lora_plain.A", (fin, r)) * 0.1
    layer.lora_B.weight.data = rng_tensor("lora_plain.B", (r, fout))
    xin = x.clone().requires_grad_(True)
    y = layer(xin)
    y.backward(g)
    out["lora_plain.y"] = y.detach().numpy()
    out["lora_plain.dx"] = xin.grad.numpy()
    out["lora_plain.dA"] = layer.lora_A.weight.grad.numpy()
    out["lora_plain.dB"] = layer.lora_B.weight.grad.numpy()
    out["lora_plain.weight"] = layer.weight(x).detach().numpy()


def golden_ot(M, CLIP):
    """The Sinkhorn / COT logits heads (trainers/GLP_OT_SVLoRA.py:615-675, 713-757) on the tiny ViT: logits, loss, all
    gradients and a 3-step trajectory per head.  Written to tests/golden/ot.npz + ot.json."""
    import dataclasses
    out, meta = {}, {}
    for ot, top in (("Sinkhorn", 1.0), ("COT", 0.8)):
        mcfg = dataclasses.replace(C.vit_tiny(rank=4), ot=ot, ot_top_percent=top)
        golden_model(M, CLIP, mcfg, f"ot_{ot.lower()}", 8, 3, out, meta)
    np.savez_compressed(os.path.join(HERE, "ot.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "ot.json"), "w"), indent=1, sort_keys=True)
    print("ot.npz:", len(out), "arrays")


def golden_svlora():
    """SVLoRALinear (one shared diagonal of singular values; trainers/GLP_OT_SVLoRA.py:255-330): initial lora_S and
    a forward / backward with trained-looking factors.  Written to tests/golden/svlora.npz."""
    import torch.nn as nn
    M, *_ = import_reference() if "trainers.GLP_OT_SVLoRA" not in sys.modules else (sys.modules["trainers.GLP_OT_SVLoRA"],)
    out = {}
    L, Bn, fin, fout, r = 50, 4, 128, 192, 8
    x = rng_tensor("svlora.x", (L, Bn, fin))
    g = rng_tensor("svlora.g", (L, Bn, fout))
    lin = nn.Linear(fin, fout)
    lin.weight.data = rng_tensor("svlora.W", (fout, fin)) * fin ** -0.5
    lin.bias.data = rng_tensor("svlora.b", (fout,)) * 0.1
    layer = M.SVLoRALinear(lin, rank=r, alpha=2.0, global_s=False)
    out["svlora.s_init"] = layer.lora_S.weight.detach().numpy().copy()
    out["svlora.s_shape"] = np.array(layer.lora_S.weight.shape)
    layer.lora_A.weight.data = rng_tensor("svlora.A", (fin, r)) * 0.1
    layer.lora_B.weight.data = rng_tensor("svlora.B", (r, fout))
    layer.lora_S.weight.data = layer.lora_S.weight.data * (1.0 + 0.1 * rng_tensor("svlora.S", (r,)))
    out["svlora.S"] = layer.lora_S.weight.detach().numpy().copy()
    xin = x.clone().requires_grad_(True)
    y = layer(xin)
    y.backward(g)
    out["svlora.y"] = y.detach().numpy()
    out["svlora.dx"] = xin.grad.numpy()
    out["svlora.dA"] = layer.lora_A.weight.grad.numpy()
    out["svlora.dS"] = layer.lora_S.weight.grad.numpy()
    out["svlora.dB"] = layer.lora_B.weight.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "svlora.npz"), **out)
    print("svlora.npz:", {k: v.shape for k, v in out.items()})


def golden_s_init(M, out):
    import torch.nn as nn
    for r in (4, 8, 12, 16, 32):
        for G in (2, 3):
            layer = M.FairLoRALinear(nn.Linear(8, 8), rank=r, alpha=2.0, num_attrs=G)
            out[f"s_init.r{r}.g{G}"] = layer.lora_S.weight.detach().numpy().copy()
            assert float(layer.lora_A.weight.abs().max()) == 0.0


def golden_model(M, CLIP, mcfg, tag, batch_size, steps, out, meta, lora_init="random", num_batches=None, step_size=200):
    sd = synth.make_state_dict(mcfg, seed=1, lora_init=lora_init)
    model = build_reference_model(M, CLIP, mcfg, sd)
    batch = synth.make_batch(mcfg, batch_size, seed=1234)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    n_total = sum(p.numel() for p in model.parameters())
    meta[f"{tag}.trainable_elems"] = n_train
    meta[f"{tag}.total_params"] = n_total
    meta[f"{tag}.trainable_tensors"] = sum(1 for p in model.parameters() if p.requires_grad)

    # trainer-level: object.__new__ + the attributes forward_backward touches
    tr = object.__new__(M.GLP_OT_SVLoRA)
    cfg = ref_cfg(mcfg)
    tr.cfg = cfg
    tr.model = model
    tr.device = torch.device("cpu")
    params = list(model.prompt_learner.parameters()) + list(model.image_encoder.parameters())
    if mcfg.dim_per_3d_slice:
        params += list(model.proj_per_3d_slice.parameters())      # trainers/GLP_OT_SVLoRA.py:862-863
    tr.optim = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4, dampening=0, nesterov=False)
    tr.sched = torch.optim.lr_scheduler.StepLR(tr.optim, step_size=step_size, gamma=0.1)
    from collections import OrderedDict
    # the reference's own wiring (trainers/GLP_OT_SVLoRA.py:866-870): both names carry the SAME optimizer and
    # scheduler, so model_update steps the optimizer twice per batch and update_lr the scheduler twice per epoch
    tr._models, tr._optims, tr._scheds = OrderedDict(), OrderedDict(), OrderedDict()
    tr.register_model("prompt_learner", model.prompt_learner, tr.optim, tr.sched)
    tr.register_model("image_encoder", model.image_encoder, tr.optim, tr.sched)
    tr._writer = None
    tr.num_batches = num_batches if num_batches else steps      # the last step of an "epoch" calls update_lr()
    model.train()

    # step 0 by hand to capture logits and grads before the update
    image, label, _, attr = tr.parse_batch_train(batch)
    t0 = time.time()
    logits = model(image, attr)
    loss = torch.nn.functional.cross_entropy(logits, label)
    tr.optim.zero_grad()
    loss.backward()
    dt = time.time() - t0
    out[f"{tag}.logits"] = logits.detach().numpy()
    meta[f"{tag}.loss0"] = float(loss)
    gn = {}
    for n, p in model.named_parameters():
        if p.requires_grad:
            gr = p.grad if p.grad is not None else torch.zeros_like(p)
            gn[n] = float(gr.norm())
            if p.numel() <= 8192 or tag.startswith("tiny") or tag.startswith("rn"):
                out[f"{tag}.grad.{n}"] = gr.numpy().copy()
            else:
                out[f"{tag}.gradsub.{n}"] = sub(gr, 1024)
    meta[f"{tag}.grad_norms"] = gn
    print(tag, "loss0", float(loss), "fwd+bwd s", round(dt, 2), "trainable", n_train, "total", n_total)
    tr.optim.zero_grad()

    # K-step trajectory through the reference's own forward_backward
    traj = []
    for i in range(steps):
        tr.batch_idx = i % tr.num_batches
        s = tr.forward_backward(batch)
        s["lr_after"] = tr.optim.param_groups[0]["lr"]
        traj.append(s)
        print(tag, "step", i, s)
    meta[f"{tag}.traj"] = traj
    meta[f"{tag}.sched"] = {"last_epoch": tr.sched.last_epoch, "lr": tr.optim.param_groups[0]["lr"],
                            "num_batches": tr.num_batches, "step_size": step_size,
                            "optimizer_steps_per_batch": sum(1 for o in tr._optims.values() if o is not None)}
    post = model.state_dict()
    for k in synth.buffer_keys(mcfg):                              # RN50: BatchNorm running statistics after the steps
        out[f"{tag}.post.{k}"] = post[k].detach().numpy().copy()
    for k in synth.trainable_keys(mcfg):
        if post[k].numel() <= 8192 or tag.startswith("tiny") or tag.startswith("rn"):
            out[f"{tag}.post.{k}"] = post[k].detach().numpy().copy()
    meta[f"{tag}.post_checksum"] = {k: float(post[k].double().sum()) for k in synth.trainable_keys(mcfg)}
    return model


def golden_adapter(M, CLIP, out, meta):
    """CustomCLIP(cfg, classnames, clip_model) (trainers/GLP_OT_SVLoRA.py:575-613) + apply_lora_to_model on a CLIP model
    that holds synth.clip_state_dict's tensors: the tokenised prompts, the token_prefix / token_suffix buffers the
    PromptLearner builds from CLIP's token embedding, the state_dict keys, and the logits once ctx / adapters are set
    to known values.  Both class-name pairs on the path (FairFedMed, FedChexMimic)."""
    import clip.clip as clipmod
    dd = {"trainer": "GLP_OT", "vision_depth": 0, "language_depth": 0, "vision_ctx": 0, "language_ctx": 0}
    for tag, mcfg, names in (("adapter_vit", C.vit_tiny(rank=4), ["NOT Glaucoma", "Glaucoma"]),
                             ("adapter_rn", C.rn_tiny(rank=4, num_groups=2), ["NOT Pleural Effusion", "Pleural Effusion"])):
        v, t = mcfg.vision, mcfg.text
        is_rn = isinstance(v, C.ResNetCfg)
        clip_model = CLIP(v.out_dim, v.image_size, tuple(v.layers) if is_rn else v.layers, v.width,
                          None if is_rn else v.patch, t.context_length, 49408, t.width, t.heads, t.layers, dd).float()
        clip_model.load_state_dict(synth.clip_state_dict(mcfg, seed=1), strict=True)
        cfg = ref_cfg(mcfg)
        model = M.CustomCLIP(cfg, names, clip_model)
        toks = model.tokenized_prompts
        # the pinned token ids of fairfedmed_amd.clip_adapter equal the reference tokenizer's
        from fairfedmed_amd import clip_adapter as A
        assert torch.equal(toks, A.tokenize_prompts(names, mcfg.n_ctx).repeat(mcfg.n_prompts, 1)), names
        out[f"{tag}.tokens"] = toks.numpy()
        out[f"{tag}.token_prefix"] = model.prompt_learner.token_prefix.detach().numpy().copy()
        out[f"{tag}.token_suffix"] = model.prompt_learner.token_suffix.detach().numpy().copy()
        bn_params = {id(p) for mod in model.modules() if isinstance(mod, torch.nn.BatchNorm2d) for p in mod.parameters()}
        for n, p in model.named_parameters():
            p.requires_grad_("prompt_learner" in n or id(p) in bn_params)
        M.apply_lora_to_model(model, True, rank=mcfg.lora.rank, alpha=mcfg.lora.alpha, lora_type="FairLoRA",
                              global_s=False, num_attrs=mcfg.lora.num_groups)
        meta[f"{tag}.keys"] = list(model.state_dict().keys())
        meta[f"{tag}.ctx_std"] = float(model.prompt_learner.ctx.std())
        # known values for everything the constructor draws at random (ctx, lora_B) and a non-trivial lora_A
        known = synth.make_state_dict(mcfg, seed=3, lora_init="random")
        model.load_state_dict({k: known[k] for k in synth.trainable_keys(mcfg) if not synth._is_bn_param(k)}, strict=False)
        batch = synth.make_batch(mcfg, 6, seed=77)
        model.eval()
        with torch.no_grad():
            logits = model(batch["img"], batch["attrs"].t()[0])
        out[f"{tag}.logits"] = logits.numpy()
        print(tag, "logits", logits[:2].tolist())


def golden_fedavg(FU, out, meta):
    G, r = 3, 8
    keys = {"a.lora_S.weight": (G, r), "a.lora_A.weight": (16, r), "prompt_learner.ctx": (2, 4, 8),
            "frozen.weight": (5, 5), "b.lora_S.weight": (G, r)}
    for case, (epoch, shared) in {"e0_shared": (0, True), "e3_shared": (3, True), "e3_plain": (3, False)}.items():
        w = {u: {k: rng_tensor(f"fed.{case}.{u}.{k}", s) for k, s in keys.items()} for u in range(3)}
        w_g = {k: rng_tensor(f"fed.{case}.g.{k}", s) for k, s in keys.items()}
        n_client = [100, 50, 25]
        by_attr = [[50, 30, 20], [10, 20, 20], [5, 5, 15]]
        idxs = [0, 2] if case == "e3_plain" else [0, 1, 2]
        res = FU.average_weights_EMA(w_g, w, idxs, n_client, by_attr, epoch, 10, shared_half_s=shared)
        for k in keys:
            out[f"fed.{case}.{k}"] = res[k].numpy()
        meta[f"fed.{case}"] = {"epoch": epoch, "max_epoch": 10, "shared_half_s": shared, "idxs": idxs,
                               "n_client": n_client, "by_attr": by_attr}


FEDAVG_PLAIN_KEYS = {"a.lora_S.weight": (3, 8), "a.lora_A.weight": (16, 8), "prompt_learner.ctx": (2, 4, 8),
                     "frozen.weight": (5, 5), "b.lora_S.weight": (3, 8)}


def golden_fedavg_plain(FU):
    """utils/fed_utils.py:6-40 (``average_weights``, no EMA): state_dict form with and without the per-attribute
    counts, and the ``islist`` form on one tensor per client -> tests/golden/fedavg_plain.npz"""
    out = {}
    n_client = [100, 50, 25]
    by_attr = [[50, 30, 20], [10, 20, 20], [5, 5, 15]]
    for case, (idxs, attr) in {"all_attr": ([0, 1, 2], True), "two_attr": ([2, 0], True), "all_noattr": ([0, 1, 2], False)}.items():
        w = {u: {k: rng_tensor(f"fedp.{case}.{u}.{k}", s) for k, s in FEDAVG_PLAIN_KEYS.items()} for u in range(3)}
        res = FU.average_weights(w, idxs, n_client, by_attr if attr else None)
        for k in FEDAVG_PLAIN_KEYS:
            out[f"{case}.{k}"] = res[k].numpy()
    wl = {u: rng_tensor(f"fedp.list.{u}", (2, 4, 8)) for u in range(3)}
    out["list"] = FU.average_weights(wl, [1, 2], n_client, islist=True).numpy()
    np.savez_compressed(os.path.join(HERE, "fedavg_plain.npz"), **out)
    print("fedavg_plain.npz:", sorted(out)[:4], "...")


def golden_auc(compute_auc, out, meta):
    g = synth._rng("auc", 7)
    cases = {}
    for name, n in (("n32", 32), ("n200", 200)):
        y = g.integers(0, 2, size=(n,))
        logit = g.standard_normal((n, 2)).astype(np.float32) + y[:, None] * np.array([[-0.5, 0.5]], np.float32)
        if name == "n200":
            logit = np.round(logit, 1)   # force ties
        prob = torch.softmax(torch.from_numpy(logit), -1)
        cases[name] = float(compute_auc(prob, torch.from_numpy(y)))
        out[f"auc.{name}.prob"] = prob.numpy()
        out[f"auc.{name}.y"] = y.astype(np.int64)
    meta["auc"] = cases


def golden_fairness(out, meta):
    """ES-AUC, per-group AUC and between-group disparity from the reference's own functions
    (evaluation/metrics.py:513-552; the fairlearn-based DPD / EOD cannot be generated: fairlearn is absent)."""
    import evaluation.metrics as EM
    g = synth._rng("fair", 11)
    res = {}
    for name, n, G in (("n300g3", 300, 3), ("n120g2", 120, 2)):
        y = g.integers(0, 2, size=(n,))
        attr = g.integers(0, G, size=(n,))
        logit = g.standard_normal((n, 2)).astype(np.float32) \
            + (y[:, None] * np.array([[-0.6, 0.6]], np.float32)) * (1.0 - 0.25 * attr[:, None]).astype(np.float32)
        prob = torch.softmax(torch.from_numpy(logit), -1).numpy()
        overall = float(EM.compute_auc(prob, y))
        aucs = [float(EM.compute_auc(prob[attr == e], y[attr == e])) for e in np.unique(attr)]
        res[name] = {"overall": overall, "group_aucs": aucs,
                     "es_auc": float(EM.equity_scaled_AUC(prob, y, attr, num_classes=2)),
                     "disparity": [float(v) for v in EM.compute_between_group_disparity(aucs, overall)]}
        out[f"fair.{name}.prob"], out[f"fair.{name}.y"], out[f"fair.{name}.attr"] = prob, y.astype(np.int64), attr.astype(np.int64)
    meta["fair"] = res


def golden_dataset():
    """FairFedMedDataset / count_by_attribute of the reference (utils/data_utils.py:559-726,
    Dassl/dassl/data/data_manager.py:443-460) on a synthetic tree written by fairfedmed_amd.data (24x24 samples, so no
    resize: skimage is a stub here).  Stored: what the reference returns, as sums / a few full arrays."""
    import tempfile
    from utils.data_utils import FairFedMedDataset as RefDS
    from Dassl.dassl.data.data_manager import DatasetWrapperAttr as RefWrap
    from fairfedmed_amd import data as D
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for modality in ("slo_fundus", "oct_bscans"):
            root = os.path.join(tmp, modality)
            base = D.write_synthetic_fairfedmed(root, sites=2, n_train=9, n_test=5, size=24, seed=3, modality=modality,
                                                unknown_every=4)
            for site in (1, 2):
                for train in (True, False):
                    ds = RefDS(base, site, attribute_type="race", attributes=["race", "gender"], modality_type=modality,
                               resolution=24, depth=3, train=train)
                    items = [ds[i] for i in range(len(ds))]
                    key = f"{modality}.site{site}.{'train' if train else 'test'}"
                    res[key] = {
                        "len": len(ds), "files": list(ds.data_files), "data_attrs": [int(a) for a in ds.data_attrs],
                        "shape": list(items[0][0].shape), "dtype": str(items[0][0].dtype),
                        "sums": [float(np.asarray(it[0], np.float64).sum()) for it in items],
                        "wsums": [float((np.asarray(it[0], np.float64).reshape(-1)
                                         * np.arange(1, it[0].size + 1)).sum()) for it in items],
                        "labels": [int(it[1]) for it in items], "attrs": [[int(v) for v in it[2]] for it in items],
                        "first_corner": np.asarray(items[0][0])[:, :3, :4].tolist(),
                        "count_race": RefWrap.count_by_attribute_fairfedmed(NS(data_source=ds), "race"),
                        "count_gender": RefWrap.count_by_attribute_fairfedmed(NS(data_source=ds), "gender"),
                    }
        # FedChexMimicDataset (utils/data_utils.py:729-790) + count_by_attribute_fedchexmimic (data_manager.py:462-473):
        # gray PNG / RGB JPEG / RGB PNG -> convert('L') -> float32 -> 3 channels, at the files' own size (no resize)
        base = D.write_synthetic_fedchexmimic(os.path.join(tmp, "chex"), n_train=7, n_test=4, size=20, seed=5)
        from utils.data_utils import FedChexMimicDataset as RefChex
        for site in (1, 2):
            for train in (True, False):
                ds = RefChex(base, site, "gender", ["gender", "race"], resolution=20, depth=3, train=train)
                items = [ds[i] for i in range(len(ds))]
                res[f"chex.site{site}.{'train' if train else 'test'}"] = {
                    "len": len(ds), "files": [str(x) for x in ds.data_files], "data_attrs": [int(a) for a in ds.data_attrs],
                    "shape": list(items[0][0].shape), "dtype": str(items[0][0].dtype),
                    "sums": [float(np.asarray(it[0], np.float64).sum()) for it in items],
                    "wsums": [float((np.asarray(it[0], np.float64).reshape(-1)
                                     * np.arange(1, it[0].size + 1)).sum()) for it in items],
                    "labels": [int(it[1]) for it in items], "label_dtype": str(items[0][1].dtype),
                    "attrs": [[int(v) for v in it[2]] for it in items],
                    "first_corner": np.asarray(items[1][0])[:, :3, :4].tolist(),
                    "count_gender": RefWrap.count_by_attribute_fedchexmimic(NS(data_source=ds), "gender"),
                    "count_race": RefWrap.count_by_attribute_fedchexmimic(NS(data_source=ds), "race"),
                }
    with open(os.path.join(HERE, "dataset.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("dataset.json:", len(res), "cases")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-dataset", action="store_true", help="regenerate tests/golden/dataset.json only")
    ap.add_argument("--only-svlora", action="store_true", help="regenerate tests/golden/svlora.npz only")
    ap.add_argument("--only-ot", action="store_true", help="regenerate tests/golden/ot.npz (Sinkhorn / COT heads) only")
    ap.add_argument("--only-fedavg-plain", action="store_true", help="regenerate tests/golden/fedavg_plain.npz only")
    ap.add_argument("--vitb", action="store_true", help="also generate the ViT-B/16 fixtures (minutes)")
    ap.add_argument("--time-ref", action="store_true", help="time the reference CPU step at bs=32")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    M, CLIP, FU, compute_auc = import_reference()
    if args.only_dataset:
        golden_dataset()
        return
    if args.only_svlora:
        golden_svlora()
        return
    if args.only_ot:
        golden_ot(M, CLIP)
        return
    if args.only_fedavg_plain:
        golden_fedavg_plain(FU)
        return
    golden_dataset()
    golden_svlora()
    golden_ot(M, CLIP)
    golden_fedavg_plain(FU)

    out, meta = {}, {"torch": torch.__version__, "numpy": np.__version__}
    golden_layers(M, out)
    golden_lora_plain(M, out)
    golden_s_init(M, out)
    golden_fedavg(FU, out, meta)
    golden_auc(compute_auc, out, meta)
    golden_fairness(out, meta)
    golden_adapter(M, CLIP, out, meta)
    np.savez_compressed(os.path.join(HERE, "unit.npz"), **out)

    out = {}
    golden_model(M, CLIP, C.vit_tiny(rank=4), "tiny_r4", 8, 3, out, meta)
    golden_model(M, CLIP, C.vit_tiny(rank=8, num_groups=2), "tiny_r8g2", 6, 2, out, meta)
    golden_model(M, CLIP, C.vit_tiny(rank=4), "tiny_refinit", 8, 3, out, meta, lora_init="reference")
    # the other adapter types of apply_lora_to_model (:516-540) and GLOBAL_S, through the reference's trainer
    import dataclasses as _dc
    base = C.vit_tiny(rank=4)
    for tag, lt, gs, G in (("tiny_globals", "FairLoRA", True, 3), ("tiny_svlora", "SVLoRA", False, 1),
                           ("tiny_svlora_globals", "SVLoRA", True, 1), ("tiny_lora", "LoRA", False, 1)):
        mc = _dc.replace(base, lora=_dc.replace(base.lora, lora_type=lt, global_s=gs, num_groups=G))
        golden_model(M, CLIP, mc, tag, 8, 3, out, meta)
    # two local epochs of two batches with StepLR(step_size=2): the scheduler is stepped twice per epoch, so the
    # second epoch already runs at lr * gamma
    golden_model(M, CLIP, C.vit_tiny(rank=4), "tiny_sched", 8, 4, out, meta, num_batches=2, step_size=2)
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **out)

    out = {}   # 3D OCT front end: 6 samples x 2 slice groups of 4 B-scans -> 12 ViT images
    golden_model(M, CLIP, C.vit_tiny_3d(rank=4, dim_per_3d_slice=4), "tiny3d_r4", 6, 3, out, meta)
    np.savez_compressed(os.path.join(HERE, "tiny3d.npz"), **out)

    out = {}   # RN50 trunk (one Bottleneck per stage, RN50's channel widths, 64x64 images), G = 2 as in configs[4]
    golden_model(M, CLIP, C.rn_tiny(rank=4, num_groups=2), "rn_tiny_r4g2", 6, 3, out, meta)
    # the same trunk with identity-skip Bottlenecks (stages (2, 1, 2, 1)): layer1.1 / layer3.1 have no downsample
    golden_model(M, CLIP, C.rn_tiny2(rank=4, num_groups=2), "rn_tiny2_r4g2", 6, 3, out, meta)
    np.savez_compressed(os.path.join(HERE, "rn_tiny.npz"), **out)

    if args.vitb:
        out = {}
        golden_model(M, CLIP, C.vit_b16(rank=8), "vitb_r8", 8, 3, out, meta)
        # the bench workload's batch size (the bf16 panel GEMMs are selected from 6304 token rows on)
        golden_model(M, CLIP, C.vit_b16(rank=8), "vitb_r8_bs32", 32, 2, out, meta)
        np.savez_compressed(os.path.join(HERE, "vitb.npz"), **out)
    else:
        prev = os.path.join(HERE, "meta.json")
        if os.path.exists(prev):
            old = json.load(open(prev))
            for k, v in old.items():
                if k.startswith("vitb") or k.startswith("ref_cpu"):
                    meta.setdefault(k, v)

    if args.time_ref:
        mcfg = C.vit_b16(rank=8)
        sd = synth.make_state_dict(mcfg, seed=1)
        model = build_reference_model(M, CLIP, mcfg, sd)
        batch = synth.make_batch(mcfg, 32, seed=1234)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4)
        attr = batch["attrs"].t()[0]
        times = []
        for i in range(4):
            t0 = time.time()
            loss = torch.nn.functional.cross_entropy(model(batch["img"], attr), batch["label"])
            opt.zero_grad()
            loss.backward()
            opt.step()
            times.append(time.time() - t0)
            print("ref step", i, times[-1])
        meta["ref_cpu_step_s_bs32"] = times
        meta["ref_cpu_threads"] = torch.get_num_threads()

    json.dump(meta, open(os.path.join(HERE, "meta.json"), "w"), indent=1, sort_keys=True)
    print("wrote goldens to", HERE)


if __name__ == "__main__":
    main()
