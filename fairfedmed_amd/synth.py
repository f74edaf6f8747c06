"""Deterministic synthetic weights and batches.

Pretrained CLIP weights and the FairFedMed datasets cannot be fetched (no
network), so every parity and benchmark run uses weights produced by a
counter-based filler keyed by the state_dict key: the same tensors can be
regenerated in this container (for the reference import that makes the golden
vectors) and on the GPU box without shipping 500 MB.

The key set and shapes are those of the reference's ``CustomCLIP`` after
``apply_lora_to_model`` (trainers/GLP_OT_SVLoRA.py:503-573, 575-613;
SURVEY.md §8(b) "state_dict keys").
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

from .config import ModelCfg, ResNetCfg


def manifest(cfg: ModelCfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict key -> shape, in the reference's registration order."""
    v, t, lo = cfg.vision, cfg.text, cfg.lora
    m: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    m["logit_scale"] = ()
    if cfg.dim_per_3d_slice:
        m["proj_per_3d_slice.weight"] = (3, cfg.dim_per_3d_slice, 5, 5)
        m["proj_per_3d_slice.bias"] = (3,)
    nrow = cfg.n_prompts * cfg.n_cls
    m["prompt_learner.ctx"] = (cfg.n_prompts, cfg.n_ctx, t.width)
    m["prompt_learner.token_prefix"] = (nrow, 1, t.width)
    m["prompt_learner.token_suffix"] = (nrow, t.context_length - 1 - cfg.n_ctx, t.width)
    ie = "image_encoder."
    if isinstance(v, ResNetCfg):
        _manifest_resnet(m, cfg)
        _manifest_text(m, cfg)
        return m
    m[ie + "class_embedding"] = (v.width,)
    m[ie + "positional_embedding"] = (v.tokens, v.width)
    m[ie + "proj"] = (v.width, v.out_dim)
    m[ie + "conv1.weight"] = (v.width, 3, v.patch, v.patch)
    m[ie + "ln_pre.weight"] = (v.width,)
    m[ie + "ln_pre.bias"] = (v.width,)
    for i in range(v.layers):
        p = f"{ie}transformer.resblocks.{i}."
        m[p + "attn.in_proj_weight"] = (3 * v.width, v.width)
        m[p + "attn.in_proj_bias"] = (3 * v.width,)
        m[p + "attn.out_proj.weight"] = (v.width, v.width)
        m[p + "attn.out_proj.bias"] = (v.width,)
        m[p + "ln_1.weight"] = (v.width,)
        m[p + "ln_1.bias"] = (v.width,)
        for name, fin, fout in (("c_fc", v.width, 4 * v.width), ("c_proj", 4 * v.width, v.width)):
            q = f"{p}mlp.{name}."
            m[q + "original_linear.weight"] = (fout, fin)
            m[q + "original_linear.bias"] = (fout,)
            m[q + "lora_A.weight"] = (fin, lo.rank)
            _manifest_s(m, q, lo)
            m[q + "lora_B.weight"] = (lo.rank, fout)
        m[p + "ln_2.weight"] = (v.width,)
        m[p + "ln_2.bias"] = (v.width,)
    m[ie + "ln_post.weight"] = (v.width,)
    m[ie + "ln_post.bias"] = (v.width,)
    _manifest_text(m, cfg)
    return m


def _manifest_s(m, q: str, lo) -> None:
    """lora_S (and lora_S_global) of one adapter, in the reference's registration order (A, S, S_global, B).  SVLoRA's
    lora_S and both types' lora_S_global are 1-D [r]: reset_parameters replaces the Embedding weight by a linspace
    (trainers/GLP_OT_SVLoRA.py:294-304, 418-422)."""
    lt = getattr(lo, "lora_type", "FairLoRA")
    if lt == "FairLoRA":
        m[q + "lora_S.weight"] = (lo.num_groups, lo.rank)
    elif lt == "SVLoRA":
        m[q + "lora_S.weight"] = (lo.rank,)
    elif lt != "LoRA":
        raise NotImplementedError(lt)
    if getattr(lo, "global_s", False) and lt != "LoRA":
        m[q + "lora_S_global.weight"] = (lo.rank,)


def _bn(m, p: str, c: int) -> None:
    m[p + "weight"] = (c,)
    m[p + "bias"] = (c,)
    m[p + "running_mean"] = (c,)
    m[p + "running_var"] = (c,)
    m[p + "num_batches_tracked"] = ()


def _manifest_resnet(m, cfg: ModelCfg) -> None:
    """ModifiedResNet_GLP_OT after apply_lora_to_model (clip/model.py:227-301, trainers/GLP_OT_SVLoRA.py:541-573):
    FairLoRALinear on Bottleneck.conv1 / conv3 (1x1), LoRALinear on the four attention-pool projections."""
    v, lo = cfg.vision, cfg.lora
    ie, w = "image_encoder.", v.width
    m[ie + "conv1.weight"] = (w // 2, 3, 3, 3)
    _bn(m, ie + "bn1.", w // 2)
    m[ie + "conv2.weight"] = (w // 2, w // 2, 3, 3)
    _bn(m, ie + "bn2.", w // 2)
    m[ie + "conv3.weight"] = (w, w // 2, 3, 3)
    _bn(m, ie + "bn3.", w)
    inpl = w
    for li, nblk in enumerate(v.layers):
        planes = w * (2 ** li)
        for j in range(nblk):
            stride = 2 if (li > 0 and j == 0) else 1
            p = f"{ie}layer{li + 1}.{j}."
            for name, cin, cout in (("conv1", inpl, planes),):
                q = p + name + "."
                m[q + "original_linear.weight"] = (cout, cin, 1, 1)
                m[q + "lora_A.weight"] = (cin, lo.rank)
                _manifest_s(m, q, lo)
                m[q + "lora_B.weight"] = (lo.rank, cout)
            _bn(m, p + "bn1.", planes)
            m[p + "conv2.weight"] = (planes, planes, 3, 3)
            _bn(m, p + "bn2.", planes)
            q = p + "conv3."
            m[q + "original_linear.weight"] = (planes * 4, planes, 1, 1)
            m[q + "lora_A.weight"] = (planes, lo.rank)
            _manifest_s(m, q, lo)
            m[q + "lora_B.weight"] = (lo.rank, planes * 4)
            _bn(m, p + "bn3.", planes * 4)
            if stride > 1 or inpl != planes * 4:
                m[p + "downsample.0.weight"] = (planes * 4, inpl, 1, 1)
                _bn(m, p + "downsample.1.", planes * 4)
            inpl = planes * 4
    e = v.embed_dim
    ap = ie + "attnpool."
    m[ap + "positional_embedding"] = (v.tokens, e)
    for name, fout in (("k_proj", e), ("q_proj", e), ("v_proj", e), ("c_proj", v.out_dim)):
        q = ap + name + "."
        m[q + "original_linear.weight"] = (fout, e)
        m[q + "original_linear.bias"] = (fout,)
        m[q + "lora_A.weight"] = (e, lo.rank)
        m[q + "lora_B.weight"] = (lo.rank, fout)


def _manifest_text(m, cfg: ModelCfg) -> None:
    v, t = cfg.vision, cfg.text
    te = "text_encoder."
    m[te + "positional_embedding"] = (t.context_length, t.width)
    m[te + "text_projection"] = (t.width, v.out_dim)
    for i in range(t.layers):
        p = f"{te}transformer.resblocks.{i}."
        m[p + "attn.in_proj_weight"] = (3 * t.width, t.width)
        m[p + "attn.in_proj_bias"] = (3 * t.width,)
        m[p + "attn.out_proj.weight"] = (t.width, t.width)
        m[p + "attn.out_proj.bias"] = (t.width,)
        m[p + "ln_1.weight"] = (t.width,)
        m[p + "ln_1.bias"] = (t.width,)
        m[p + "mlp.c_fc.weight"] = (4 * t.width, t.width)
        m[p + "mlp.c_fc.bias"] = (4 * t.width,)
        m[p + "mlp.c_proj.weight"] = (t.width, 4 * t.width)
        m[p + "mlp.c_proj.bias"] = (t.width,)
        m[p + "ln_2.weight"] = (t.width,)
        m[p + "ln_2.bias"] = (t.width,)
    m[te + "ln_final.weight"] = (t.width,)
    m[te + "ln_final.bias"] = (t.width,)


def trainable_keys(cfg: ModelCfg):
    """Keys with requires_grad after the reference's freeze loop + LoRA
    injection (trainers/GLP_OT_SVLoRA.py:822-842): prompt_learner.ctx,
    proj_per_3d_slice.*, and every lora_{A,S,B}."""
    out = []
    for k in manifest(cfg):
        if k == "prompt_learner.ctx" or k.startswith("proj_per_3d_slice.") or ".lora_" in k:
            out.append(k)
        elif _is_bn_param(k):                       # BatchNorm2d weight / bias stay trainable (:825-827)
            out.append(k)
    return out


def _is_bn_param(k: str) -> bool:
    parts = k.split(".")
    return len(parts) >= 2 and (parts[-2].startswith("bn") or (len(parts) >= 3 and parts[-3] == "downsample" and parts[-2] == "1")) \
        and parts[-1] in ("weight", "bias") and k.startswith("image_encoder.")


def buffer_keys(cfg: ModelCfg):
    """Non-parameter state that training changes: BatchNorm running statistics and counters (RN50 only)."""
    return [k for k in manifest(cfg) if k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(key.encode()), seed & 0xFFFFFFFF]))


def lora_s_init(rank: int, num_groups: int) -> torch.Tensor:
    """'same+cycle' initial singular values (trainers/GLP_OT_SVLoRA.py:402-417):
    first r/2 columns = linspace(.5,.1,r/2) shared by all groups, last r/2 =
    the same ramp cyclically shifted by i*((r/2)//G) for group i, times 0.2."""
    assert rank % 2 == 0 and rank >= num_groups
    h = rank // 2
    ramp = torch.linspace(0.5, 0.1, steps=h)
    shift = h // num_groups
    cyc = torch.stack([torch.cat([ramp[i * shift:], ramp[: i * shift]]) for i in range(num_groups)])
    return torch.cat([ramp[None].repeat(num_groups, 1), cyc * 0.2], dim=1)


def _std_for(key: str, cfg: ModelCfg) -> Tuple[str, float, float]:
    """(kind, mean, std) per key; scales follow CLIP.initialize_parameters
    (clip/model.py:533-560) so activations stay O(1) through 12 layers."""
    v, t = cfg.vision, cfg.text
    if isinstance(v, ResNetCfg) and key.startswith("image_encoder."):
        if key.endswith("running_mean"):
            return "const", 0.0, 0.0
        if key.endswith("running_var"):
            return "const", 1.0, 0.0
        if key.endswith("num_batches_tracked"):
            return "const", 0.0, 0.0
        if _is_bn_param(key):
            return ("normal", 1.0, 0.1) if key.endswith("weight") else ("normal", 0.0, 0.1)
        if key.endswith("positional_embedding"):
            return "normal", 0.0, v.embed_dim ** -0.5
        if "attnpool" in key:
            return ("normal", 0.0, v.embed_dim ** -0.5) if key.endswith("weight") else ("normal", 0.0, 0.02)
        return "normal", 0.0, 0.0          # convolutions: He-style std from the fan-in, set by the caller
    width = v.width if key.startswith("image_encoder.") else t.width
    layers = v.layers if key.startswith("image_encoder.") else t.layers
    if key == "logit_scale":
        return "const", float(np.log(1 / 0.07)), 0.0
    if key.endswith("ln_1.weight") or key.endswith("ln_2.weight") or key.endswith("ln_pre.weight") \
            or key.endswith("ln_post.weight") or key.endswith("ln_final.weight"):
        return "normal", 1.0, 0.1
    if ".ln_" in key and key.endswith(".bias"):
        return "normal", 0.0, 0.1
    if key.endswith("attn.in_proj_weight"):
        return "normal", 0.0, width ** -0.5
    if key.endswith("attn.out_proj.weight") or "c_proj" in key and key.endswith("weight") and "lora" not in key:
        return "normal", 0.0, (width ** -0.5) * ((2 * layers) ** -0.5)
    if "c_fc" in key and key.endswith("weight") and "lora" not in key:
        return "normal", 0.0, (2 * width) ** -0.5
    if key.endswith("bias"):
        return "normal", 0.0, 0.02
    if key.endswith("conv1.weight"):
        return "normal", 0.0, 0.02
    if key.endswith("class_embedding") or key.endswith("image_encoder.positional_embedding") or key.endswith("image_encoder.proj"):
        return "normal", 0.0, v.width ** -0.5
    if key.endswith("text_encoder.positional_embedding"):
        return "normal", 0.0, 0.01
    if key.endswith("text_projection"):
        return "normal", 0.0, t.width ** -0.5
    if key.startswith("prompt_learner."):
        return "normal", 0.0, 0.02
    if key == "proj_per_3d_slice.weight":
        return "normal", 0.0, cfg.dim_per_3d_slice ** -0.5
    raise KeyError(key)


def make_state_dict(cfg: ModelCfg, seed: int = 1, lora_init: str = "reference") -> Dict[str, torch.Tensor]:
    """fp32 CPU state_dict with the manifest's keys.

    lora_init="reference": lora_A = 0, lora_B ~ N(0,1), lora_S 'same+cycle'
      (trainers/GLP_OT_SVLoRA.py:380-423) -> the adapter is a no-op at step 0.
    lora_init="random": lora_A ~ N(0, 0.05) and lora_S perturbed, so every
      gradient path (dA, dB, dS, LoRA dx) is non-trivial in a single step.
    """
    sd: Dict[str, torch.Tensor] = OrderedDict()
    lo = cfg.lora
    for key, shape in manifest(cfg).items():
        g = _rng(key, seed)
        if key.endswith("lora_A.weight"):
            if lora_init == "reference":
                x = np.zeros(shape, np.float32)
            else:
                x = g.standard_normal(shape, dtype=np.float32) * 0.05
        elif key.endswith("lora_B.weight"):
            x = g.standard_normal(shape, dtype=np.float32)
        elif key.endswith("lora_S.weight") and len(shape) == 2:
            x = lora_s_init(lo.rank, lo.num_groups).numpy().astype(np.float32)
            if lora_init != "reference":
                x = x + g.standard_normal(shape, dtype=np.float32) * 0.05
        elif key.endswith("lora_S.weight") or key.endswith("lora_S_global.weight"):
            # SVLoRA's shared diagonal and GLOBAL_S: linspace(1, 0.1, r) (trainers/GLP_OT_SVLoRA.py:294-304, 418-422)
            x = np.linspace(1.0, 0.1, lo.rank, dtype=np.float32)
            if lora_init != "reference":
                x = x + g.standard_normal(shape, dtype=np.float32) * 0.05
        else:
            kind, mean, std = _std_for(key, cfg)
            if kind == "normal" and std == 0.0 and len(shape) == 4:       # convolution: He-style std from the fan-in
                std = (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
            if kind == "const":
                x = np.full(shape, mean, np.float32)
            else:
                x = mean + g.standard_normal(shape, dtype=np.float32) * np.float32(std)
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        sd[key] = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).reshape(shape)
    return sd


def make_batch(cfg: ModelCfg, batch: int, seed: int = 1234, signal: float = 0.0, slices: int = 2, overlap: float = 0.0):
    """Synthetic batch in the reference's dict contract:
    img f32 [B,C,H,W] raw 0..255 (C = 3, or slices*dim_per_3d_slice for 3D OCT), label i64 [B], attrs i64 [B,1].

    signal > 0 adds a label-dependent mean shift on a fixed patch mask and a
    group-dependent contrast, so AUC can move off 0.5.

    overlap > 0 (with signal): every sample's shift also carries its own N(0, overlap^2) offset, the same on all pixels of
    the patch - the two classes' patch means OVERLAP, so the task has an irreducible error and the AUC of a converged model
    plateaus near Phi(sqrt(2) signal / overlap) < 1 instead of saturating at 1 (the round-6 RN AUC fixture: a plateau set by
    the data is where a +-0.002 criterion can be asked of 16-bit storage at all, tests/test_auc_parity_gpu.py)."""
    g = np.random.Generator(np.random.Philox(key=[0xBA7C4, seed & 0xFFFFFFFF]))
    v = cfg.vision
    c = 3 if not cfg.dim_per_3d_slice else slices * cfg.dim_per_3d_slice   # 3D OCT: `slices` groups of B-scans
    img = g.random((batch, c, v.image_size, v.image_size), dtype=np.float32)
    label = g.integers(0, cfg.n_cls, size=(batch,), dtype=np.int64)
    attr = g.integers(0, cfg.lora.num_groups, size=(batch,), dtype=np.int64)
    if signal > 0:
        h = v.image_size
        mask = np.zeros((h, h), np.float32)
        mask[h // 4: h // 2, h // 4: h // 2] = 1.0
        contrast = 1.0 - 0.15 * attr.astype(np.float32)
        img = (img - 0.5) * contrast[:, None, None, None] + 0.5
        shift = signal * (label.astype(np.float32) * 2 - 1)
        if overlap > 0:                                # (drawn last: the streams of the other fixtures do not move)
            shift = shift + np.float32(overlap) * g.standard_normal(batch, dtype=np.float32)
        img = img + shift[:, None, None, None] * mask[None, None]
        img = np.clip(img, 0.0, 1.0)
    img = img * np.float32(255.0)
    return {
        "img": torch.from_numpy(img),
        "label": torch.from_numpy(label),
        "attrs": torch.from_numpy(attr[:, None].copy()),
    }


# --------------------------------------------------------------------------
# A CLIP-shaped model for the reference-signature constructor CustomCLIP(cfg, classnames, clip_model)
# --------------------------------------------------------------------------
def clip_state_dict(cfg: ModelCfg, seed: int = 1, vocab: int = 49408) -> "OrderedDict[str, torch.Tensor]":
    """The tensors of make_state_dict(cfg, seed) under clip.model.CLIP's keys (visual.*, transformer.*,
    positional_embedding, ln_final.*, text_projection, logit_scale; clip/model.py:453-531) plus a token embedding
    [vocab, width] ~ N(0, 0.02) (:539)."""
    sd = make_state_dict(cfg, seed=seed, lora_init="reference")
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        if ".lora_" in k or k.startswith("prompt_learner.") or k.startswith("proj_per_3d_slice."):
            continue
        if k.startswith("image_encoder."):
            out["visual." + k[len("image_encoder."):].replace(".original_linear.", ".")] = v
        elif k.startswith("text_encoder."):
            out[k[len("text_encoder."):]] = v
        else:
            out[k] = v                                              # logit_scale
    g = _rng("token_embedding.weight", seed)
    out["token_embedding.weight"] = torch.from_numpy(
        g.standard_normal((vocab, cfg.text.width), dtype=np.float32) * np.float32(0.02))
    return out


def make_clip_model(cfg: ModelCfg, seed: int = 1):
    """An nn.Module with clip.model.CLIP's state_dict and the attributes CustomCLIP.__init__ / PromptLearner.__init__
    read from it (trainers/GLP_OT_SVLoRA.py:69-128, 575-613): dtype, visual.input_resolution, ln_final.weight,
    token_embedding(ids), logit_scale."""
    import torch.nn as nn
    from .model import _Node, _register
    root = _Node()
    for k, v in clip_state_dict(cfg, seed).items():
        _register(root, k, v.clone(), trainable=False, buffer=k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    root.visual.input_resolution = cfg.vision.image_size
    root.dtype = torch.float32
    emb = root.token_embedding.weight
    root.token_embedding.forward = lambda ids, _w=emb: _w[ids]
    return root
