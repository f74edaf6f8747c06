"""Shape/config records for the FairLoRA hot path.

These are the few values of the reference's yacs tree that the hot path reads
(SURVEY.md §5 "Config / flags"): CLIP geometry (clip/model.py:453-531),
TRAINER.GLP_OT.{N, N_CTX}, TRAINER.GLP_OT_LORA.{RANK, ALPHA}, the number of
demographic groups (trainers/GLP_OT_SVLoRA.py:775-790) and the SGD/StepLR
settings (configs/trainers/GLP_OT/vit_b16_oph.yaml:15-22).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Tuple, Union

# CLIP normalisation constants used by every FairLoRA yaml
# (configs/trainers/GLP_OT/vit_b16_oph.yaml:9-13).
CLIP_PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass(frozen=True)
class VisionCfg:
    image_size: int = 224
    patch: int = 16
    width: int = 768
    layers: int = 12
    heads: int = 12
    out_dim: int = 512

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def tokens(self) -> int:  # L = grid^2 + 1 (class token)
        return self.grid * self.grid + 1

    @property
    def head_dim(self) -> int:
        return self.width // self.heads


@dataclass(frozen=True)
class ResNetCfg:
    """CLIP's ModifiedResNet (clip/model.py:227-301): 3-conv stem, Bottleneck stages with anti-aliasing average
    pools, attention pool returning all (H/32)^2 + 1 tokens.  heads and embed_dim follow CLIP.__init__ (:474-486)."""
    image_size: int = 224
    width: int = 64
    layers: Tuple[int, int, int, int] = (3, 4, 6, 3)
    out_dim: int = 1024

    @property
    def embed_dim(self) -> int:
        return self.width * 32

    @property
    def heads(self) -> int:
        return self.width * 32 // 64

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.heads

    @property
    def spacial(self) -> int:
        return self.image_size // 32

    @property
    def tokens(self) -> int:
        return self.spacial * self.spacial + 1


@dataclass(frozen=True)
class TextCfg:
    context_length: int = 77
    width: int = 512
    heads: int = 8
    layers: int = 12


@dataclass(frozen=True)
class LoraCfg:
    rank: int = 8
    alpha: float = 2.0
    num_groups: int = 3       # G = len(retrieval_attributes(ATTRIBUTE_TYPE))
    lambda_group: float = 0.7  # trainers/GLP_OT_SVLoRA.py:459
    # TRAINER.GLP_OT_LORA.TYPE (trainers/GLP_OT_SVLoRA.py:516-540): 'FairLoRA' (every script), 'SVLoRA' (one shared
    # diagonal, lora_S is a 1-D [r] tensor after reset_parameters) or 'LoRA' (no lora_S).  The last two ignore the
    # attribute: set num_groups = 1.
    lora_type: str = "FairLoRA"
    # TRAINER.GLP_OT_LORA.GLOBAL_S: one more trainable vector lora_S_global [r] (1-D after reset_parameters,
    # :418-422 / :300-304) added to every sample's singular values: s_b = pi_b S + S_global
    global_s: bool = False

    @property
    def scaling(self) -> float:  # trainers/GLP_OT_SVLoRA.py:346
        return self.alpha / self.rank


@dataclass(frozen=True)
class OptimCfg:
    lr: float = 1e-3
    momentum: float = 0.9
    weight_decay: float = 5e-4
    stepsize: int = 200
    gamma: float = 0.1


@dataclass(frozen=True)
class ModelCfg:
    vision: Union[VisionCfg, ResNetCfg] = field(default_factory=VisionCfg)
    text: TextCfg = field(default_factory=TextCfg)
    lora: LoraCfg = field(default_factory=LoraCfg)
    n_prompts: int = 2         # TRAINER.GLP_OT.N
    n_ctx: int = 4             # TRAINER.GLP_OT.N_CTX
    n_cls: int = 2
    # EOT position of each of the n_cls tokenised prompts (argmax of token ids,
    # trainers/GLP_OT_SVLoRA.py:64).  Defaults: "X X X X NOT Glaucoma." -> 9,
    # "X X X X Glaucoma." -> 8.
    eot: Tuple[int, ...] = (9, 8)
    pixel_mean: Tuple[float, float, float] = CLIP_PIXEL_MEAN
    pixel_std: Tuple[float, float, float] = CLIP_PIXEL_STD
    # 3D OCT: slices are grouped by `dim_per_3d_slice` channels
    # (trainers/GLP_OT_SVLoRA.py:585-595); 0 = 2D input.
    dim_per_3d_slice: int = 0
    # logits head (TRAINER.GLP_OT.OT): 'None' (every FairLoRA script), 'Sinkhorn' or 'COT' with the plan's
    # parameters EPS / THRESH / MAX_ITER / TOP_PERCENT (trainers/GLP_OT_SVLoRA.py:604-613)
    ot: str = "None"
    ot_eps: float = 0.1
    ot_thresh: float = 1e-3
    ot_max_iter: int = 100
    ot_top_percent: float = 1.0

    def to_dict(self):
        return asdict(self)


def vit_b16(rank: int = 8, alpha: float = 2.0, num_groups: int = 3) -> ModelCfg:
    """CLIP ViT-B/16 FairLoRA (BASELINE.json configs[1..3])."""
    return ModelCfg(lora=LoraCfg(rank=rank, alpha=alpha, num_groups=num_groups))


def vit_tiny(rank: int = 4, alpha: float = 2.0, num_groups: int = 3) -> ModelCfg:
    """A small geometry with the same structure, for fast parity tests:
    64x64 image, 16 patches + cls = 17 tokens, width 128 (2 heads of 64),
    2 vision layers; text width 128 (2 heads), 2 layers, context 77 (the
    reference's tokenizer always pads prompts to 77, clip/clip.py:180-220)."""
    return ModelCfg(
        vision=VisionCfg(image_size=64, patch=16, width=128, layers=2, heads=2, out_dim=128),
        text=TextCfg(context_length=77, width=128, heads=2, layers=2),
        lora=LoraCfg(rank=rank, alpha=alpha, num_groups=num_groups),
        eot=(9, 8),
    )


def vit_tiny_3d(rank: int = 4, dim_per_3d_slice: int = 4, num_groups: int = 3) -> ModelCfg:
    """vit_tiny with the 3D OCT front end (DATASET.MODALITY_TYPE 'oct_bscans'): every group of
    `dim_per_3d_slice` B-scans goes through the trainable 5x5 conv (trainers/GLP_OT_SVLoRA.py:634-639)."""
    import dataclasses
    return dataclasses.replace(vit_tiny(rank=rank, num_groups=num_groups), dim_per_3d_slice=dim_per_3d_slice)


def rn50(rank: int = 8, alpha: float = 8.0, num_groups: int = 2) -> ModelCfg:
    """CLIP RN50 FairLoRA (BASELINE.json configs[4]): FairLoRA on the 1x1 convolutions conv1/conv3 of every
    Bottleneck, plain LoRA on the attention pool, train-mode BatchNorm (trainers/GLP_OT_SVLoRA.py:541-573, 822-829)."""
    return ModelCfg(vision=ResNetCfg(), lora=LoraCfg(rank=rank, alpha=alpha, num_groups=num_groups))


def rn_tiny(rank: int = 4, alpha: float = 2.0, num_groups: int = 2) -> ModelCfg:
    """RN50's channel widths (the GEMM wants K % 64 == 0) with one Bottleneck per stage on 64x64 images:
    4 + 1 attention-pool tokens, 256-d joint space, the tiny text tower."""
    return ModelCfg(vision=ResNetCfg(image_size=64, width=64, layers=(1, 1, 1, 1), out_dim=256),
                    text=TextCfg(context_length=77, width=128, heads=2, layers=2),
                    lora=LoraCfg(rank=rank, alpha=alpha, num_groups=num_groups), eot=(9, 8))


def rn_tiny2(rank: int = 4, alpha: float = 2.0, num_groups: int = 2) -> ModelCfg:
    """rn_tiny with identity-skip Bottlenecks: stages (2, 1, 2, 1), so layer1.1 and layer3.1 have no downsample
    path (clip/model.py:41-60; 12 of RN50's 16 blocks are of that kind)."""
    import dataclasses
    base = rn_tiny(rank=rank, alpha=alpha, num_groups=num_groups)
    return dataclasses.replace(base, vision=dataclasses.replace(base.vision, layers=(2, 1, 2, 1)))


def vit_tiny_lora(lora_type: str = "FairLoRA", global_s: bool = False, rank: int = 4) -> ModelCfg:
    """vit_tiny with another adapter type of apply_lora_to_model (trainers/GLP_OT_SVLoRA.py:516-540) and / or GLOBAL_S;
    LoRA and SVLoRA have no demographic groups (num_groups = 1)."""
    import dataclasses
    base = vit_tiny(rank=rank)
    G = base.lora.num_groups if lora_type == "FairLoRA" else 1
    return dataclasses.replace(base, lora=dataclasses.replace(base.lora, lora_type=lora_type, global_s=global_s, num_groups=G))
