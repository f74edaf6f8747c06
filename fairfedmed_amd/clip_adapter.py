"""``CustomCLIP(cfg, classnames, clip_model)``: the reference's constructor signature
(trainers/GLP_OT_SVLoRA.py:575-613) on top of the HIP engine.

``clip_model`` is anything with the interface of the reference's ``clip.model.CLIP`` (clip/model.py:453-531): a
``state_dict()`` with the keys ``visual.*``, ``transformer.*``, ``token_embedding.weight``, ``positional_embedding``,
``ln_final.*``, ``text_projection``, ``logit_scale``.  The geometry is read from the tensor shapes exactly as
``build_model`` does (clip/model.py:633-670); the keys are renamed to ``CustomCLIP``'s (``image_encoder.*``,
``text_encoder.*``), the prompt learner's tensors are built as ``PromptLearner.__init__`` builds them (:69-128:
``ctx ~ N(0, 0.02)`` of shape [N, n_ctx, d], ``token_prefix`` / ``token_suffix`` = the token embedding of
"X X X X <classname>." without the context positions) and the adapters are initialised as ``apply_lora_to_model`` would
(:503-573; settings from ``cfg.TRAINER.GLP_OT_LORA``), so ``apply_lora_to_model(model, ...)`` afterwards only has to
check that it was asked for the same thing.

The BPE tokenizer is load-time tooling and out of scope (its vocabulary is a third-party data file); the token ids of
the prompts of the two datasets on the path are pinned below (asserted equal to the
reference's tokenizer when the goldens are generated).  Other class names need ``tokenize=`` (e.g. the reference's
``clip.tokenize``).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

from . import config as C
from .synth import lora_s_init, manifest

Tensor = torch.Tensor

SOT, EOT, X_TOKEN, DOT = 49406, 49407, 343, 269
# BPE ids of the class names of FairFedMed ("NOT Glaucoma" / "Glaucoma") and FedChexMimic
CLASSNAME_TOKENS = {
    "NOT Glaucoma": [783, 39171, 19620],
    "Glaucoma": [39171, 19620],
    "NOT Pleural Effusion": [783, 926, 33948, 1490, 9364],
    "Pleural Effusion": [926, 33948, 1490, 9364],
}
MODALITIES_3D = {"oct_bscans", "oct_bscans_3d", "mac_onh", "onh_mac"}


def tokenize_prompts(classnames: Sequence[str], n_ctx: int, context_length: int = 77,
                     tokenize: Optional[Callable[[str], Tensor]] = None) -> Tensor:
    """clip.tokenize("X X X X <name>.") for every class (trainers/GLP_OT_SVLoRA.py:108-112): int64 [n_cls, 77],
    <start> X*n_ctx name . <end> then zeros."""
    rows = []
    for name in classnames:
        name = name.replace("_", " ")
        if tokenize is not None:
            rows.append(torch.as_tensor(tokenize(" ".join(["X"] * n_ctx) + " " + name + ".")).reshape(-1).long())
            continue
        if name not in CLASSNAME_TOKENS:
            raise NotImplementedError(f"token ids of class name {name!r} are not pinned: pass tokenize=clip.tokenize")
        ids = [SOT] + [X_TOKEN] * n_ctx + CLASSNAME_TOKENS[name] + [DOT, EOT]
        rows.append(torch.tensor(ids + [0] * (context_length - len(ids)), dtype=torch.long))
    return torch.stack(rows)


def geometry_from_clip(sd: Dict[str, Tensor]) -> Tuple[object, C.TextCfg]:
    """(VisionCfg | ResNetCfg, TextCfg) from a CLIP state_dict, as clip/model.py:633-658 derives them."""
    embed_dim = sd["text_projection"].shape[1]
    if "visual.proj" in sd:
        width = sd["visual.conv1.weight"].shape[0]
        layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        patch = sd["visual.conv1.weight"].shape[-1]
        grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        vision = C.VisionCfg(image_size=patch * grid, patch=patch, width=width, layers=layers, heads=width // 64,
                             out_dim=embed_dim)
    else:
        counts = tuple(len(set(k.split(".")[2] for k in sd if k.startswith(f"visual.layer{b}"))) for b in (1, 2, 3, 4))
        width = sd["visual.layer1.0.conv1.weight"].shape[0]
        ow = round((sd["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
        assert ow * ow + 1 == sd["visual.attnpool.positional_embedding"].shape[0]
        vision = C.ResNetCfg(image_size=ow * 32, width=width, layers=counts, out_dim=embed_dim)
    tw = sd["ln_final.weight"].shape[0]
    text = C.TextCfg(context_length=sd["positional_embedding"].shape[0], width=tw, heads=tw // 64,
                     layers=len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks"))))
    return vision, text


def model_cfg_from_reference(cfg, classnames: Sequence[str], clip_sd: Dict[str, Tensor], tokens: Tensor) -> C.ModelCfg:
    """ModelCfg from the reference's config tree (the fields CustomCLIP.__init__ / build_model read) + CLIP's shapes."""
    vision, text = geometry_from_clip(clip_sd)
    got, lora = cfg.TRAINER.GLP_OT, getattr(cfg.TRAINER, "GLP_OT_LORA", None)
    assert cfg.INPUT.SIZE[0] == vision.image_size, \
        f"cfg_imsize ({cfg.INPUT.SIZE[0]}) must equal to clip_imsize ({vision.image_size})"       # :79
    # The other prompt variants of PromptLearner (trainers/GLP_OT_SVLoRA.py:84-175) are not merely unused by the FairLoRA
    # scripts: with this trainer's N prompts they do not run in the reference either.  CTX_INIT leaves a 2-D ctx that
    # forward() permutes as 4-D (:133-136); CSC allocates ctx [n_cls, n_ctx, d] and views it as [N * n_cls, ...]
    # (consistent only when N == n_cls); 'middle' / 'front' assemble n_cls prompts (`for i in range(self.n_cls)`,
    # :152-175) for N * n_cls tokenised rows.  There is no behaviour to match, so they stay an explicit error here.
    if getattr(got, "CTX_INIT", False) or getattr(got, "CSC", False):
        raise NotImplementedError("CTX_INIT / CSC prompts: no FairLoRA script sets them, and the reference's PromptLearner "
                                  "does not run with them for N > 1 prompts (ctx shape mismatch in forward())")
    if getattr(got, "CLASS_TOKEN_POSITION", "end") != "end":
        raise NotImplementedError(f"CLASS_TOKEN_POSITION={got.CLASS_TOKEN_POSITION!r}: the reference assembles n_cls prompts "
                                  "for N * n_cls tokenised rows in that branch; only 'end' is a working configuration")
    is3d = getattr(cfg.DATASET, "MODALITY_TYPE", "slo_fundus") in MODALITIES_3D
    if lora is not None and getattr(lora, "UNFREEZE_IMAGE_ENCODER", True):
        ltype = getattr(lora, "TYPE", "FairLoRA")
        if getattr(lora, "DISABLE_ATTR", False) or ltype != "FairLoRA":
            G = 1
        else:
            from .trainer import ATTRIBUTE_GROUPS
            G = len(ATTRIBUTE_GROUPS[cfg.DATASET.NAME][cfg.DATASET.ATTRIBUTE_TYPE])
        lo = C.LoraCfg(rank=lora.RANK, alpha=lora.ALPHA, num_groups=G, lora_type=ltype,
                       global_s=bool(getattr(lora, "GLOBAL_S", False)))
    else:
        raise NotImplementedError("UNFREEZE_IMAGE_ENCODER must be set: without it no adapter is injected")
    return C.ModelCfg(vision=vision, text=text, lora=lo, n_prompts=got.N, n_ctx=got.N_CTX, n_cls=len(classnames),
                      eot=tuple(int(i) for i in tokens.argmax(-1)),
                      pixel_mean=tuple(cfg.INPUT.PIXEL_MEAN), pixel_std=tuple(cfg.INPUT.PIXEL_STD),
                      dim_per_3d_slice=cfg.DATASET.DIM_PER_3D_SLICE if is3d else 0,
                      ot=str(getattr(got, "OT", "None")), ot_eps=float(getattr(got, "EPS", 0.1)),
                      ot_thresh=float(getattr(got, "THRESH", 1e-3)), ot_max_iter=int(getattr(got, "MAX_ITER", 100)),
                      ot_top_percent=float(getattr(got, "TOP_PERCENT", 1.0)))


def state_dict_from_clip(mcfg: C.ModelCfg, clip_sd: Dict[str, Tensor], tokens: Tensor) -> "OrderedDict[str, Tensor]":
    """CustomCLIP's state_dict (manifest order) from CLIP's tensors + freshly initialised prompt / adapter tensors."""
    out: "OrderedDict[str, Tensor]" = OrderedDict()
    lo = mcfg.lora
    emb = clip_sd["token_embedding.weight"].float()
    embedding = emb[tokens.repeat(mcfg.n_prompts, 1)]                      # [N * n_cls, 77, d]  (:109-113)
    for key, shape in manifest(mcfg).items():
        if key == "prompt_learner.ctx":
            t = torch.empty(shape)
            torch.nn.init.normal_(t, std=0.02)                             # :98
        elif key == "prompt_learner.token_prefix":
            t = embedding[:, :1, :].clone()                                # SOS  (:118)
        elif key == "prompt_learner.token_suffix":
            t = embedding[:, 1 + mcfg.n_ctx:, :].clone()                   # class name, EOS, padding  (:119)
        elif key == "proj_per_3d_slice.weight":
            t = torch.empty(shape)
            torch.nn.init.normal_(t, std=mcfg.dim_per_3d_slice ** -0.5)    # :592-594
        elif key == "proj_per_3d_slice.bias":
            t = torch.zeros(shape)
        elif key.endswith("lora_A.weight"):
            t = torch.zeros(shape)                                         # reset_parameters: A = 0, B ~ N(0, 1)
        elif key.endswith("lora_B.weight"):
            t = torch.empty(shape)
            torch.nn.init.normal_(t)
        elif key.endswith("lora_S.weight") and len(shape) == 2:
            t = lora_s_init(lo.rank, lo.num_groups)                        # 'same+cycle' (:402-417)
        elif key.endswith("lora_S.weight") or key.endswith("lora_S_global.weight"):
            t = torch.linspace(1, 0.1, steps=lo.rank)                      # :294-304, 418-422
        elif key.startswith("image_encoder."):
            t = clip_sd["visual." + key[len("image_encoder."):].replace(".original_linear.", ".")]
        elif key.startswith("text_encoder.transformer."):
            t = clip_sd[key[len("text_encoder."):]]
        elif key.startswith("text_encoder."):
            t = clip_sd[key[len("text_encoder."):]]                        # positional_embedding, ln_final.*, text_projection
        elif key == "logit_scale":
            t = clip_sd["logit_scale"]
        else:
            raise KeyError(key)
        t = t.detach()
        out[key] = t.to(torch.int64) if key.endswith("num_batches_tracked") else t.float().reshape(shape).cpu().clone()
    return out


def from_reference_args(cfg, classnames: Sequence[str], clip_model, tokenize=None):
    """(ModelCfg, state_dict, tokenized_prompts [N * n_cls, 77]) for CustomCLIP(cfg, classnames, clip_model)."""
    clip_sd = {k: v.detach().cpu() for k, v in clip_model.state_dict().items()}
    tokens = tokenize_prompts(classnames, cfg.TRAINER.GLP_OT.N_CTX, clip_sd["positional_embedding"].shape[0], tokenize)
    mcfg = model_cfg_from_reference(cfg, classnames, clip_sd, tokens)
    sd = state_dict_from_clip(mcfg, clip_sd, tokens)
    return mcfg, sd, tokens.repeat(mcfg.n_prompts, 1)
