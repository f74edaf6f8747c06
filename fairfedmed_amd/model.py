"""Module-level API of the reference's FairLoRA model on top of the HIP engine.

  * ``CustomCLIP`` — same ``state_dict`` keys, shapes and order as the
    reference's ``CustomCLIP`` after ``apply_lora_to_model``
    (trainers/GLP_OT_SVLoRA.py:503-613), so that
    ``federated_main.py`` / ``fed_utils.average_weights*`` see identical
    dictionaries.  Trainable parameters are views of the engine's flat buffer;
    ``forward(image, attr)`` runs the HIP engine (GPU only, no fallback).
  * ``FairLoRALinear`` — the wrapped-linear layer with the reference's
    constructor/attributes (trainers/GLP_OT_SVLoRA.py:333-482); forward and
    backward go through the C ABI (ffm_lora_down / ffm_gemm_nt /
    ffm_lora_grad_partial), the dense dW is never formed.
  * ``apply_lora_to_model`` — the reference's injection rule for the ViT
    backbone (every nn.Linear under ``.mlp.``, :516).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .config import ModelCfg
from .engine import FairLoRAEngine
from .synth import buffer_keys, lora_s_init, manifest, trainable_keys

Tensor = torch.Tensor


class _Node(nn.Module):
    """Generic container so that parameter paths reproduce the reference's keys."""


def _register(root: nn.Module, key: str, tensor: Tensor, trainable: bool, buffer: bool) -> None:
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=trainable))


class CustomCLIP(nn.Module):
    """CLIP image encoder (+FairLoRA) + prompt learner + text encoder + logits head.

    Two constructor forms:
      CustomCLIP(ModelCfg, state_dict, dtype=, max_images=, device=)   geometry + a state_dict with CustomCLIP's keys;
      CustomCLIP(cfg, classnames, clip_model, ...)                     the reference's signature
          (trainers/GLP_OT_SVLoRA.py:575-613): a yacs-style config tree, the class names and a CLIP model; see
          fairfedmed_amd/clip_adapter.py.  dtype defaults to the config's TRAINER.GLP_OT.PREC then (fp32 / amp -> float32,
          otherwise bf16), max_images to max(train, test batch size) when the config names them."""

    def __init__(self, cfg, state_dict, clip_model=None, dtype=None, max_images: Optional[int] = None,
                 device: str = "cuda:0", tokenize=None):
        super().__init__()
        if not isinstance(cfg, ModelCfg):
            from .clip_adapter import from_reference_args
            if clip_model is None:
                raise TypeError("CustomCLIP(cfg, classnames, clip_model): clip_model is missing")
            ref_cfg, classnames = cfg, list(state_dict)
            cfg, state_dict, self.tokenized_prompts = from_reference_args(ref_cfg, classnames, clip_model, tokenize)
            if dtype is None:
                from .trainer import resolve_precision
                dtype = resolve_precision(ref_cfg.TRAINER.GLP_OT.PREC)
            if max_images is None:
                try:
                    max_images = max(ref_cfg.DATALOADER.TRAIN_X.BATCH_SIZE, ref_cfg.TEST.BATCH_SIZE)
                except AttributeError:
                    max_images = 32
            self.n_cls, self.N = len(classnames), cfg.n_prompts
        dtype = torch.bfloat16 if dtype is None else dtype
        max_images = 32 if max_images is None else max_images
        self.cfg = cfg
        from .engine_rn import create_engine
        self.engine = create_engine(cfg, state_dict, dtype=dtype, max_images=max_images, device=device)
        train = set(trainable_keys(cfg))
        dev = self.engine.device
        # RN50: BatchNorm running statistics are buffers that training changes; they alias the engine's tensors
        live = self.engine.buffer_views() if hasattr(self.engine, "buffer_views") else {}
        for key, shape in manifest(cfg).items():
            if key in train:
                t = self.engine.params.view(key)                   # view of the flat fp32 buffer
            elif key in live:
                t = live[key]
            else:
                t = state_dict[key].detach().to(dev, torch.float32).reshape(shape).clone()
            _register(self, key, t, key in train, buffer=key.startswith("prompt_learner.token_") or key in live)
        for key in train:                                          # .grad = view of the flat grad buffer
            self.get_parameter(key).grad = self.engine.params.view(key, "grad")

    @property
    def dtype(self):
        return self.engine.dtype

    def forward(self, image: Tensor, attr: Optional[Tensor] = None) -> Tensor:
        """logits [B, n_cls] (trainers/GLP_OT_SVLoRA.py:677-763, OT='None')."""
        return self.engine.forward(image, attr)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        res = super().load_state_dict(state_dict, strict=strict, assign=False)
        train = set(self.engine.params.keys) | set(buffer_keys(self.cfg))   # live tensors: nothing to rebuild
        if any(k not in train for k in state_dict):
            # frozen tensors may have changed: rebuild their compute-dtype copies (W and W^T)
            self.engine.load_frozen(self.state_dict())
        return res


# --------------------------------------------------------------------------
# Stand-alone FairLoRA layer
# --------------------------------------------------------------------------
class _FairLoRAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, W, Wt, bias, A, S, Bm, attr32, rows_per_sample, scaling, lam):
        M, K = x2d.shape
        N, r, G = W.shape[0], A.shape[1], S.shape[0]
        t = torch.empty(M, r, device=x2d.device)
        ts = torch.empty(M, r, device=x2d.device)
        ops.lora_down(x2d, A, False, S, attr32, r, G, rows_per_sample, scaling, lam, t, ts)
        y = torch.empty(M, N, device=x2d.device, dtype=x2d.dtype)
        ops.gemm_nt(x2d, W, y, bias=bias, ts=ts, lw=Bm)
        ctx.save_for_backward(x2d, Wt, A, S, Bm, t, ts, attr32 if attr32 is not None else torch.empty(0))
        ctx.meta = (rows_per_sample, scaling, lam, attr32 is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x2d, Wt, A, S, Bm, t, ts, attr32 = ctx.saved_tensors
        rps, scaling, lam, has_attr = ctx.meta
        attr32 = attr32 if has_attr else None
        g = g.contiguous()
        M, K = x2d.shape
        N, r, G = g.shape[1], A.shape[1], S.shape[0]
        dev = g.device
        u, us = torch.empty(M, r, device=dev), torch.empty(M, r, device=dev)
        nb = ops.lora_down_blocks(M, N, r, g.dtype)
        ds_part = torch.empty(nb, G, r, device=dev)
        ops.lora_down(g, Bm, True, S, attr32, r, G, rps, scaling, lam, u, us, t, ds_part)
        dx = torch.empty(M, K, device=dev, dtype=g.dtype)
        ops.gemm_nt(g, Wt, dx, ts=us, lw=A, lw_is_kr=True)
        ns = ops.lora_grad_splits(M)
        part = torch.empty(ns * max(K, N) * r, device=dev)
        dA, dB, dS = torch.empty_like(A), torch.empty_like(Bm), torch.empty_like(S)
        ops.lora_grad_partial(x2d, us, r, part)
        ops.reduce_partials(part, ns, K * r, dA)
        ops.lora_grad_partial(g, ts, r, part)
        ops.reduce_partials(part, ns, N * r, dB, transpose_K=N, transpose_r=r)
        ops.reduce_partials(ds_part, nb, G * r, dS)
        return dx, None, None, None, dA, dS, dB, None, None, None, None


class _Emb(nn.Module):
    """nn.Embedding-shaped holder (the reference stores LoRA matrices as Embedding.weight)."""

    def __init__(self, rows: int, cols: int):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(rows, cols))


class FairLoRALinear(nn.Module):
    def __init__(self, original_linear: nn.Linear, rank: int = 4, alpha: float = 0.4, global_s: bool = False,
                 num_attrs: int = 1):
        super().__init__()
        assert num_attrs > 0, "Number of attributes must be provided!"
        self.is_1x1_conv = isinstance(original_linear, nn.Conv2d)
        if self.is_1x1_conv:
            # RN50 form (trainers/GLP_OT_SVLoRA.py:344-352, 469-480): a 1x1 convolution is the same product on the
            # rows (b, h, w) of the NHWC view of the feature map
            if original_linear.kernel_size != (1, 1) or original_linear.stride != (1, 1) or original_linear.groups != 1:
                raise NotImplementedError("FairLoRA wraps 1x1, stride-1 convolutions only")
            fin, fout = original_linear.in_channels, original_linear.out_channels
        else:
            fin, fout = original_linear.in_features, original_linear.out_features
        self.original_linear = original_linear
        self.rank, self.alpha, self.scaling = rank, alpha, alpha / rank
        self.global_s, self.num_attrs = global_s, num_attrs
        self.lora_A, self.lora_S = _Emb(fin, rank), _Emb(num_attrs, rank)
        if global_s:
            self.lora_S_global = _Emb(1, rank)                     # registered between lora_S and lora_B (:359-363)
        self.lora_B = _Emb(rank, fout)
        dev = original_linear.weight.device
        self.to(dev)
        for p in self.original_linear.parameters():
            p.requires_grad = False
        self.reset_parameters()
        self._cache = None

    def reset_parameters(self):
        """A = 0, B ~ N(0,1), S 'same+cycle' (trainers/GLP_OT_SVLoRA.py:380-423)."""
        nn.init.zeros_(self.lora_A.weight)
        self.lora_S.weight.data.copy_(lora_s_init(self.rank, self.num_attrs))
        if self.global_s:                                          # a 1-D [r] tensor replaces the [1, r] weight (:418-422)
            self.lora_S_global.weight.data = torch.linspace(1, 0.1, steps=self.rank, device=self.lora_S.weight.device)
        nn.init.normal_(self.lora_B.weight)

    def _s(self) -> Tensor:
        """[G, r] singular values the kernels mix per sample; under GLOBAL_S s_b = pi_b S + S_global = pi_b (S + S_global)
        because the mix sums to one (autograd splits the gradient between the two tensors)."""
        return self.lora_S.weight + self.lora_S_global.weight[None] if self.global_s else self.lora_S.weight

    def weight(self, x: Tensor, attr: Optional[Tensor] = None) -> Tensor:
        """Per-sample dense weights W + scaling (A diag(s_b) B)^T, [x.shape[1], out, in]
        (trainers/GLP_OT_SVLoRA.py:425-445).  As in the reference this mixes with the PLAIN one-hot (s_b = S[attr_b], or
        the uniform mean without attr), not with forward()'s 0.7 / 0.3 mix, and it is differentiable in the LoRA
        factors.  A materialisation helper off the hot path (the reference itself only consumes LoRALinear.weight(),
        in the attention pool): plain tensor algebra on whatever device the parameters live on."""
        if attr is not None:
            pi = torch.nn.functional.one_hot(attr, num_classes=self.num_attrs).to(x.device, x.dtype)
        else:
            pi = torch.ones(1, self.num_attrs, device=x.device, dtype=x.dtype) / self.num_attrs
        s = pi @ self.lora_S.weight
        if self.global_s:
            s = s + self.lora_S_global.weight[None]
        s = s[:, None].repeat(1, x.shape[1] // s.shape[0], 1).flatten(0, 1)          # every sample's slices (:437-438)
        dw = (self.lora_A.weight[None] * s[:, None, :]) @ self.lora_B.weight          # [rows, in, out]
        w = self.original_linear.weight
        return w.reshape(w.shape[0], -1)[None] + self.scaling * dw.permute(0, 2, 1)

    def bias(self):
        return self.original_linear.bias

    def _frozen(self, dtype):
        w = self.original_linear.weight
        w = w.reshape(w.shape[0], -1)
        if self._cache is None or self._cache[0] != (w.data_ptr(), w._version, dtype):
            self._cache = ((w.data_ptr(), w._version, dtype), ops.cast_from_f32(w.detach().float(), dtype),
                           ops.transpose_cast(w.detach().float(), dtype))
        return self._cache[1], self._cache[2]

    def forward(self, x: Tensor, attr: Optional[Tensor] = None) -> Tensor:
        """x: [L, Bn, in] token-major as in the reference (clip/model.py:438); Bn = b*S."""
        if not x.is_cuda:
            raise RuntimeError("FairLoRALinear runs on the MI355X HIP kernels only; there is no CPU path")
        if self.is_1x1_conv:
            Bn, fin, H, Wd = x.shape                       # x: [Bn, C, H, W] -> rows (b, h, w)
            b = Bn if attr is None else attr.shape[0]
            W, Wt = self._frozen(x.dtype)
            x2d = x.permute(0, 2, 3, 1).reshape(Bn * H * Wd, fin).contiguous()
            bias = None if self.original_linear.bias is None else self.original_linear.bias.detach().float()
            a32 = None if attr is None else attr.to(torch.int32).contiguous()
            y = _FairLoRAFn.apply(x2d, W, Wt, bias, self.lora_A.weight, self._s(), self.lora_B.weight, a32,
                                  H * Wd * (Bn // b), self.scaling, 0.7)
            return y.reshape(Bn, H, Wd, -1).permute(0, 3, 1, 2)
        L, Bn, fin = x.shape
        b = Bn if attr is None else attr.shape[0]
        S = Bn // b
        W, Wt = self._frozen(x.dtype)
        x2d = x.permute(1, 0, 2).reshape(Bn * L, fin).contiguous()          # image-major rows
        bias = None if self.original_linear.bias is None else self.original_linear.bias.detach().float()
        a32 = None if attr is None else attr.to(torch.int32).contiguous()
        y = _FairLoRAFn.apply(x2d, W, Wt, bias, self.lora_A.weight, self._s(), self.lora_B.weight, a32,
                              L * S, self.scaling, 0.7)
        return y.reshape(Bn, L, -1).permute(1, 0, 2)


class LoRALinear(nn.Module):
    """Plain LoRA, y = x W^T + b + (alpha / r) (x A) B (trainers/GLP_OT_SVLoRA.py:203-252; the RN50 attention pool's
    q/k/v/c projections).  It is the FairLoRA product with one group and s = 1, so it runs on the same kernels."""

    def __init__(self, original_linear: nn.Linear, rank: int = 4, alpha: float = 0.04):
        super().__init__()
        self.original_linear = original_linear
        self.rank, self.alpha, self.scaling = rank, alpha, alpha / rank
        self.lora_A, self.lora_B = _Emb(original_linear.in_features, rank), _Emb(rank, original_linear.out_features)
        self.to(original_linear.weight.device)
        self.register_buffer("_ones", torch.ones(1, rank, device=original_linear.weight.device), persistent=False)
        for p in self.original_linear.parameters():
            p.requires_grad = False
        self.reset_parameters()
        self._cache = None

    def reset_parameters(self):
        nn.init.zeros_(self.lora_A.weight)
        nn.init.normal_(self.lora_B.weight)

    def weight(self, x=None, attr=None):
        """Dense W + scaling (A B)^T, as the attention pool consumes it (clip/model.py:90-93)."""
        return self.original_linear.weight + self.scaling * (self.lora_A.weight @ self.lora_B.weight).t()

    def bias(self):
        return self.original_linear.bias

    _frozen = FairLoRALinear._frozen

    def forward(self, x: Tensor, attr: Optional[Tensor] = None) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError("LoRALinear runs on the MI355X HIP kernels only; there is no CPU path")
        lead, fin = x.shape[:-1], x.shape[-1]
        W, Wt = self._frozen(x.dtype)
        x2d = x.reshape(-1, fin).contiguous()
        bias = None if self.original_linear.bias is None else self.original_linear.bias.detach().float()
        y = _FairLoRAFn.apply(x2d, W, Wt, bias, self.lora_A.weight, self._ones, self.lora_B.weight, None,
                              x2d.shape[0], self.scaling, 0.7)
        return y.reshape(*lead, -1)


class SVLoRALinear(nn.Module):
    """One shared diagonal of singular values, y = x W^T + b + (alpha / r) ((x A) * s) B
    (trainers/GLP_OT_SVLoRA.py:255-330): FairLoRA with a single group.  As in the reference, ``lora_S.weight`` is a
    1-D tensor of length r after ``reset_parameters`` (linspace(1, 0.1, r) replaces the [r, 1] embedding weight)."""

    def __init__(self, original_linear: nn.Linear, rank: int = 4, alpha: float = 0.4, global_s: bool = False):
        super().__init__()
        self.original_linear = original_linear
        self.rank, self.alpha, self.scaling, self.global_s = rank, alpha, alpha / rank, global_s
        self.lora_A, self.lora_S = _Emb(original_linear.in_features, rank), _Emb(rank, 1)
        if global_s:
            self.lora_S_global = _Emb(rank, 1)                     # :272-273
        self.lora_B = _Emb(rank, original_linear.out_features)
        self.to(original_linear.weight.device)
        for p in self.original_linear.parameters():
            p.requires_grad = False
        self.reset_parameters()
        self._cache = None

    def reset_parameters(self):
        nn.init.zeros_(self.lora_A.weight)
        self.lora_S.weight.data = torch.linspace(1, 0.1, steps=self.rank, device=self.lora_S.weight.device)
        if self.global_s:                                          # :300-304
            self.lora_S_global.weight.data = torch.linspace(1, 0.1, steps=self.rank, device=self.lora_S.weight.device)
        nn.init.normal_(self.lora_B.weight)

    _frozen = FairLoRALinear._frozen

    def forward(self, x: Tensor, attr: Optional[Tensor] = None) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError("SVLoRALinear runs on the MI355X HIP kernels only; there is no CPU path")
        lead, fin = x.shape[:-1], x.shape[-1]
        W, Wt = self._frozen(x.dtype)
        x2d = x.reshape(-1, fin).contiguous()
        bias = None if self.original_linear.bias is None else self.original_linear.bias.detach().float()
        s = self.lora_S.weight + self.lora_S_global.weight if self.global_s else self.lora_S.weight     # :307-309
        y = _FairLoRAFn.apply(x2d, W, Wt, bias, self.lora_A.weight, s.view(1, -1), self.lora_B.weight,
                              None, x2d.shape[0], self.scaling, 0.7)
        return y.reshape(*lead, -1)


def apply_lora_to_model(model: nn.Module, unfreeze_image_encoder: bool, rank: int = 4, alpha: float = 0.04,
                        lora_type: str = "FairLoRA", global_s: bool = False, num_attrs: int = 1) -> None:
    """The reference's injection rules (trainers/GLP_OT_SVLoRA.py:503-573) for modules under 'image_encoder.':
    ViT - every nn.Linear whose name contains '.mlp.' becomes a FairLoRALinear; ResNet - every 1x1 nn.Conv2d named
    '*conv*' under 'layer*' becomes a FairLoRALinear (downsample.0 is not named conv and stays), every nn.Linear
    of 'attnpool' a plain LoRALinear."""
    if lora_type not in ("LoRA", "SVLoRA", "FairLoRA"):
        raise NotImplementedError(lora_type)
    if isinstance(model, CustomCLIP):
        # the engine-backed CustomCLIP carries its adapters from construction (its image encoder is a sequence of HIP
        # launches, not nn.Linear modules to wrap): the reference's call sequence CustomCLIP(...) ->
        # apply_lora_to_model(model, ...) is honoured by checking that the request matches what was built
        lo = model.cfg.lora
        want = (int(rank), float(alpha), lora_type, bool(global_s) and lora_type != "LoRA",
                int(num_attrs) if lora_type == "FairLoRA" else 1)
        have = (lo.rank, float(lo.alpha), lo.lora_type, bool(lo.global_s), lo.num_groups)
        if not unfreeze_image_encoder:
            raise NotImplementedError("unfreeze_image_encoder=False: the engine always carries the adapters")
        if want != have:
            raise ValueError(f"apply_lora_to_model asks for (rank, alpha, type, global_s, groups) = {want}, the model "
                             f"was built with {have}: set cfg.TRAINER.GLP_OT_LORA accordingly")
        return
    for name, module in dict(model.named_modules()).items():
        if not (unfreeze_image_encoder and name.startswith("image_encoder.")):
            continue
        if isinstance(module, nn.Linear) and ".mlp." in name:
            if lora_type == "LoRA":
                new = LoRALinear(module, rank=rank, alpha=alpha)
            elif lora_type == "SVLoRA":
                new = SVLoRALinear(module, rank=rank, alpha=alpha, global_s=global_s)
            else:
                new = FairLoRALinear(module, rank=rank, alpha=alpha, global_s=global_s, num_attrs=num_attrs)
        elif name.startswith("image_encoder.layer") or name.startswith("image_encoder.attnpool"):
            if "attnpool" in name and isinstance(module, nn.Linear):
                new = LoRALinear(module, rank=rank, alpha=alpha)
            elif isinstance(module, nn.Conv2d) and "conv" in name and tuple(module.weight.shape[-2:]) == (1, 1):
                if lora_type != "FairLoRA":
                    raise NotImplementedError(lora_type)           # :561-567: the ResNet branch only knows FairLoRA
                new = FairLoRALinear(module, rank=rank, alpha=alpha, global_s=global_s, num_attrs=num_attrs)
            else:
                continue
        else:
            continue
        parent = model
        parts = name.split(".")
        for p in parts[:-1]:
            parent = getattr(parent, p)
        setattr(parent, parts[-1], new)
