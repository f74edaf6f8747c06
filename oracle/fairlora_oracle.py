"""ORACLE — CPU restatement of the reference's FairLoRA hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``fairfedmed_amd``) never does and fails loudly when its
HIP library is missing.

What it restates (paths relative to the reference repo, Harvard-AI-and-Robotics-Lab/FairFedMed):
  * trainers/GLP_OT_SVLoRA.py:333-482   FairLoRALinear (forward + hand-derived backward)
  * trainers/GLP_OT_SVLoRA.py:46-66,131-152  TextEncoder / PromptLearner (class token at "end")
  * trainers/GLP_OT_SVLoRA.py:677-763   CustomCLIP.forward with OT='None'
  * trainers/GLP_OT_SVLoRA.py:883-975   forward_backward (loss, SGD step, metrics)
  * clip/model.py:304-374,413-449       LayerNorm / QuickGELU / MLP / ResidualAttentionBlock /
                                        ModifiedVisionTransformer
  * Dassl/dassl/optim/optimizer.py:105-113  torch.optim.SGD(momentum, weight_decay)
  * utils/fed_utils.py:42-100           average_weights_EMA
  * evaluation/metrics.py:340-356       compute_auc (binary, sklearn roc_auc_score)

It is written as plain functions over a ``state_dict`` (dict of fp32 CPU
tensors with the reference's keys) so that it shares no module code with the
reference.  Dense arithmetic uses the same third-party library the reference
uses (PyTorch CPU kernels, fp32).

Parity pin: the reference has no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself, imported in the build container with random weights from
``fairfedmed_amd.synth`` — see ``tests/golden/make_golden.py`` (generator) and
``tests/test_oracle_golden.py`` (check).  Tolerance there: fp32 rtol 1e-5.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# FairLoRA layer
# --------------------------------------------------------------------------
def group_mix(attr: Optional[Tensor], num_groups: int, lambda_group: float = 0.7,
              dtype=torch.float32) -> Tensor:
    """pi[b, g]: 0.7 on the sample's own group, 0.3/(G-1) on the others;
    uniform 1/G when attr is None (trainers/GLP_OT_SVLoRA.py:453-462)."""
    if attr is None:
        return torch.full((1, num_groups), 1.0 / num_groups, dtype=dtype)
    onehot = F.one_hot(attr.long(), num_classes=num_groups).to(dtype)
    return onehot * lambda_group + (1 - onehot) * (1 - lambda_group) / (num_groups - 1)


def fairlora_linear(x: Tensor, W: Tensor, b: Optional[Tensor], A: Tensor, S: Optional[Tensor], Bm: Tensor,
                    attr: Optional[Tensor], scaling: float, S_global: Optional[Tensor] = None) -> Tensor:
    """y = x W^T + b + scaling * ((x A) * s_b) B,  s_b = pi_b S (+ S_global under GLOBAL_S, :467-468).

    S [G, r]: FairLoRALinear.  S [r] (1-D): SVLoRALinear, one diagonal shared by all samples, the attribute is
    ignored (:307-311).  S None: LoRALinear, s = 1 (:241-242).

    x is [L, Bn, in] (token-major like the reference, clip/model.py:438) or,
    for a 1x1 conv, [b, c_in, h, w] viewed as [hw, b, c_in]
    (trainers/GLP_OT_SVLoRA.py:469-471).  Each sample's s_b is repeated over
    its slices when x carries Bn = b*S rows (:474-475).
    """
    conv = W.dim() == 4
    if conv:
        y = F.conv2d(x, W, b)
        bb, c_in, h, w = x.shape
        x = x.reshape(bb, c_in, h * w).permute(2, 0, 1)
    else:
        y = F.linear(x, W, b)
    if S is None:
        s = torch.ones(1, A.shape[1], dtype=x.dtype)
    elif S.dim() == 1:
        s = S[None]
    else:
        s = group_mix(attr, S.shape[0], dtype=x.dtype) @ S          # [b, r]
    if S_global is not None:
        s = s + S_global[None]
    num_slices = x.shape[1] // s.shape[0]
    s = s[:, None, :].repeat(1, num_slices, 1).flatten(0, 1)        # [Bn, r]
    t = x @ A                                                      # [L, Bn, r]
    dy = ((t * s[None]) @ Bm) * scaling
    if conv:
        dy = dy.reshape(h, w, bb, -1).permute(2, 3, 0, 1)
    return y + dy


def fairlora_dense_weight(W: Tensor, A: Tensor, S: Tensor, Bm: Tensor, attr: Optional[Tensor], scaling: float,
                          num_rows: int, S_global: Optional[Tensor] = None) -> Tensor:
    """FairLoRALinear.weight(x, attr) (trainers/GLP_OT_SVLoRA.py:425-445): per-sample dense weights
    W + scaling (A diag(s_b) B)^T, [num_rows, out, in] with num_rows = x.shape[1].  NOTE the reference's weight() mixes
    with the PLAIN one-hot (s_b = S[attr_b]; uniform 1/G without attr), not with the 0.7 / 0.3 mix of forward()."""
    G = S.shape[0]
    if attr is not None:
        pi = F.one_hot(attr.long(), num_classes=G).to(W.dtype)
    else:
        pi = torch.full((1, G), 1.0 / G, dtype=W.dtype)
    s = pi @ S
    if S_global is not None:
        s = s + S_global[None]
    s = s[:, None, :].repeat(1, num_rows // s.shape[0], 1).flatten(0, 1)          # [rows, r]
    dw = (A[None] * s[:, None, :]) @ Bm                                            # [rows, in, out]
    return W[None] + scaling * dw.permute(0, 2, 1)


def fairlora_backward(x: Tensor, g: Tensor, W: Tensor, A: Tensor, S: Tensor, Bm: Tensor,
                      attr: Optional[Tensor], scaling: float):
    """Hand-derived gradients of ``fairlora_linear`` (linear form, x [L,Bn,in],
    g = dL/dy [L,Bn,out]); checked against autograd in tests.

      t = x A, u = g B^T
      dA = scaling * sum_{l,b} x^T (u * s_b)          [in, r]
      dB = scaling * sum_{l,b} (t * s_b)^T g          [r, out]
      dS = pi^T (scaling * sum_l t * u)  per sample   [G, r]
      dx = g W + scaling * (u * s_b) A^T
    """
    G = S.shape[0]
    pi = group_mix(attr, G, dtype=x.dtype)                          # [b, G]
    s = pi @ S
    num_slices = x.shape[1] // s.shape[0]
    s_full = s[:, None, :].repeat(1, num_slices, 1).flatten(0, 1)   # [Bn, r]
    t = x @ A
    u = g @ Bm.t()
    us = u * s_full[None]
    ts = t * s_full[None]
    dA = scaling * torch.einsum("lbi,lbr->ir", x, us)
    dB = scaling * torch.einsum("lbr,lbo->ro", ts, g)
    tu = scaling * (t * u).sum(0)                                   # [Bn, r]
    tu = tu.reshape(s.shape[0], num_slices, -1).sum(1)              # [b, r]
    dS = pi.expand(s.shape[0], G).t() @ tu
    dx = g @ W + scaling * (us @ A.t())
    return dx, dA, dS, dB


# --------------------------------------------------------------------------
# CLIP pieces
# --------------------------------------------------------------------------
def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """clip/model.py:304-310 — statistics in fp32, eps 1e-5."""
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5).to(x.dtype)


def quick_gelu(x: Tensor) -> Tensor:
    """clip/model.py:313-315."""
    return x * torch.sigmoid(1.702 * x)


def mha(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor, heads: int,
        mask: Optional[Tensor]) -> Tensor:
    """nn.MultiheadAttention(x, x, x, need_weights=False) with x [L, N, E]
    (clip/model.py:350-352): packed in-projection, q scaled by hd^-0.5,
    additive mask, softmax, out-projection."""
    L, N, E = x.shape
    hd = E // heads
    qkv = F.linear(x, in_w, in_b)
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(L, N * heads, hd).transpose(0, 1) * (hd ** -0.5)
    k = k.reshape(L, N * heads, hd).transpose(0, 1)
    v = v.reshape(L, N * heads, hd).transpose(0, 1)
    s = q @ k.transpose(1, 2)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(0, 1).reshape(L, N, E)
    return F.linear(o, out_w, out_b)


def vision_block(x: Tensor, sd: Dict[str, Tensor], p: str, heads: int, attr: Optional[Tensor],
                 scaling: float) -> Tensor:
    """ResidualAttentionBlock with FairLoRA-wrapped MLP (clip/model.py:325-357)."""
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
    x = x + mha(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads, None)
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])

    def lora(name, inp):
        q = f"{p}mlp.{name}."
        return fairlora_linear(inp, sd[q + "original_linear.weight"], sd[q + "original_linear.bias"],
                               sd[q + "lora_A.weight"], sd.get(q + "lora_S.weight"), sd[q + "lora_B.weight"],
                               attr, scaling, sd.get(q + "lora_S_global.weight"))

    h = lora("c_fc", h)
    h = quick_gelu(h)
    h = lora("c_proj", h)
    return x + h


# --------------------------------------------------------------------------
# RN50 trunk (clip/model.py:11-118, 227-301) with the FairLoRA / LoRA wrappers of
# trainers/GLP_OT_SVLoRA.py:541-573; BatchNorm in train mode updates its running
# statistics IN PLACE in `sd`, as nn.BatchNorm2d does during the reference's forward.
# --------------------------------------------------------------------------
class _StoreBF16(torch.autograd.Function):
    """A tensor written to memory as bfloat16 and read back (value AND gradient): what a 16-bit engine does to every
    activation it keeps.  Not part of the reference's algorithm - see STORE below."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


# Control for the bf16 parity bounds of the RN tower (tests/test_fullsize_gpu.py): None = the reference's fp32
# algorithm.  Set to `store_bf16` the SAME fp32 algorithm rounds every stored activation of the ResNet trunk (convolution
# outputs, BatchNorm / ReLU outputs, pooled maps, block outputs) and its gradient to bfloat16 - an independent statement of
# "this network with 16-bit activation storage", against which the HIP bf16 engine's distance from the fp32 oracle
# can be judged (ReLU masks flip under 2^-9 perturbations; the fp32-vs-bf16 gradient cosine of a random-weight RN50 is
# set by that, not by any kernel).
STORE = None


def store_bf16(x: Tensor) -> Tensor:
    return _StoreBF16.apply(x)


def _st(x: Tensor) -> Tensor:
    return x if STORE is None else STORE(x)


def batch_norm(sd: Dict[str, Tensor], p: str, x: Tensor, training: bool) -> Tensor:
    if training:
        sd[p + "num_batches_tracked"] += 1
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training, 0.1, 1e-5)


def bottleneck(sd: Dict[str, Tensor], p: str, x: Tensor, attr: Optional[Tensor], stride: int, scaling: float,
               training: bool) -> Tensor:
    """Bottleneck.forward (clip/model.py:41-60): all convolutions have stride 1, an average pool follows conv2 when
    stride > 1, and the downsample path is avgpool -> 1x1 conv -> BN."""
    def lora_conv(name, inp):
        q = p + name + "."
        return fairlora_linear(inp, sd[q + "original_linear.weight"], None, sd[q + "lora_A.weight"],
                               sd[q + "lora_S.weight"], sd[q + "lora_B.weight"], attr, scaling,
                               sd.get(q + "lora_S_global.weight"))

    out = _st(F.relu(batch_norm(sd, p + "bn1.", _st(lora_conv("conv1", x)), training)))
    out = _st(F.relu(batch_norm(sd, p + "bn2.", _st(F.conv2d(out, sd[p + "conv2.weight"], None, padding=1)), training)))
    if stride > 1:
        out = _st(F.avg_pool2d(out, stride))
    out = batch_norm(sd, p + "bn3.", _st(lora_conv("conv3", out)), training)
    identity = x
    if p + "downsample.0.weight" in sd:
        identity = _st(F.avg_pool2d(x, stride)) if stride > 1 else x
        identity = _st(batch_norm(sd, p + "downsample.1.", _st(F.conv2d(identity, sd[p + "downsample.0.weight"])), training))
    return _st(F.relu(out + identity))


def attention_pool(sd: Dict[str, Tensor], p: str, x: Tensor, heads: int, scaling: float) -> Tensor:
    """AttentionPool2d.forward (clip/model.py:75-118) with the dense LoRA weights W + scaling (A B)^T of
    LoRALinear.weight (trainers/GLP_OT_SVLoRA.py:235-236): returns all HW + 1 tokens, [HW+1, B, out]."""
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(2, 0, 1)                  # NCHW -> (HW) N C
    x = torch.cat([x.mean(dim=0, keepdim=True), x], dim=0)
    x = x + sd[p + "positional_embedding"][:, None, :]

    def dense(name):
        q = p + name + "."
        return (sd[q + "original_linear.weight"] + scaling * (sd[q + "lora_A.weight"] @ sd[q + "lora_B.weight"]).t(),
                sd[q + "original_linear.bias"])

    (wq, bq), (wk, bk), (wv, bv), (wc, bc) = dense("q_proj"), dense("k_proj"), dense("v_proj"), dense("c_proj")
    L, N, E = x.shape
    hd = E // heads
    q = F.linear(x, wq, bq).reshape(L, N * heads, hd).transpose(0, 1) * (hd ** -0.5)
    k = F.linear(x, wk, bk).reshape(L, N * heads, hd).transpose(0, 1)
    v = F.linear(x, wv, bv).reshape(L, N * heads, hd).transpose(0, 1)
    o = (torch.softmax(q @ k.transpose(1, 2), dim=-1) @ v).transpose(0, 1).reshape(L, N, E)
    return F.linear(o, wc, bc)


def resnet_forward(sd: Dict[str, Tensor], image: Tensor, attr: Optional[Tensor], cfg, training: bool = True) -> Tensor:
    """ModifiedResNet_GLP_OT.forward (clip/model.py:270-301): [B,3,H,W] -> [HW/1024 + 1, B, out_dim]."""
    v, ie, sc = cfg.vision, "image_encoder.", cfg.lora.scaling
    x = image
    for i, stride in ((1, 2), (2, 1), (3, 1)):
        x = _st(F.relu(batch_norm(sd, f"{ie}bn{i}.", _st(F.conv2d(x, sd[f"{ie}conv{i}.weight"], None, stride=stride, padding=1)),
                                  training)))
    x = _st(F.avg_pool2d(x, 2))
    for li, nblk in enumerate(v.layers):
        for j in range(nblk):
            x = bottleneck(sd, f"{ie}layer{li + 1}.{j}.", x, attr, 2 if (li > 0 and j == 0) else 1, sc, training)
    return attention_pool(sd, ie + "attnpool.", x, v.heads, sc)


def vision_forward(sd: Dict[str, Tensor], image: Tensor, attr: Optional[Tensor], cfg, training: bool = True) -> Tensor:
    """ModifiedVisionTransformer.forward (clip/model.py:430-449): returns ALL
    tokens projected to out_dim, [L, B, out].  (RN50 configs dispatch to resnet_forward.)"""
    v = cfg.vision
    if hasattr(v, "embed_dim"):
        return resnet_forward(sd, image, attr, cfg, training)
    ie = "image_encoder."
    x = F.conv2d(image, sd[ie + "conv1.weight"], None, stride=v.patch)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd[ie + "class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + sd[ie + "positional_embedding"]
    x = layer_norm(x, sd[ie + "ln_pre.weight"], sd[ie + "ln_pre.bias"])
    x = x.permute(1, 0, 2)
    for i in range(v.layers):
        x = vision_block(x, sd, f"{ie}transformer.resblocks.{i}.", v.heads, attr, cfg.lora.scaling)
    x = x.permute(1, 0, 2)
    x = layer_norm(x, sd[ie + "ln_post.weight"], sd[ie + "ln_post.bias"])
    x = x @ sd[ie + "proj"]
    return x.permute(1, 0, 2)


def text_forward(sd: Dict[str, Tensor], cfg) -> Tensor:
    """PromptLearner.forward (class token position 'end',
    trainers/GLP_OT_SVLoRA.py:131-152) + TextEncoder.forward (:55-66).
    Returns [N, n_cls, out]."""
    t = cfg.text
    ctx = sd["prompt_learner.ctx"]                                   # [N, n_ctx, w]
    N, n_ctx, w = ctx.shape
    ctx = ctx.unsqueeze(0).expand(cfg.n_cls, -1, -1, -1).permute(1, 0, 2, 3)
    ctx = ctx.contiguous().view(N * cfg.n_cls, n_ctx, w)
    prompts = torch.cat([sd["prompt_learner.token_prefix"], ctx, sd["prompt_learner.token_suffix"]], dim=1)
    te = "text_encoder."
    x = prompts + sd[te + "positional_embedding"]
    x = x.permute(1, 0, 2)
    L = x.shape[0]
    mask = torch.full((L, L), float("-inf")).triu_(1)               # clip/model.py:562-568
    for i in range(t.layers):
        p = f"{te}transformer.resblocks.{i}."
        h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        x = x + mha(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                    sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], t.heads, mask)
        h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        h = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        h = quick_gelu(h)
        h = F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
        x = x + h
    x = x.permute(1, 0, 2)
    x = layer_norm(x, sd[te + "ln_final.weight"], sd[te + "ln_final.bias"])
    eot = torch.tensor(list(cfg.eot) * cfg.n_prompts)               # tokenized_prompts.argmax(-1)
    x = x[torch.arange(x.shape[0]), eot] @ sd[te + "text_projection"]
    return x.view(cfg.n_prompts, cfg.n_cls, -1)


def preprocess(sd: Dict[str, Tensor], image: Tensor, cfg) -> Tensor:
    """/255, optional per-slice 5x5 conv + min-max, CLIP mean/std
    (trainers/GLP_OT_SVLoRA.py:678-693)."""
    b, c, h, w = image.shape
    x = image / 255.0
    if cfg.dim_per_3d_slice:
        x = x.reshape(-1, cfg.dim_per_3d_slice, h, w)
        x = F.conv2d(x, sd["proj_per_3d_slice.weight"], sd["proj_per_3d_slice.bias"], padding=2)
        mn = x.amin(dim=(1, 2, 3), keepdim=True)
        mx = x.amax(dim=(1, 2, 3), keepdim=True)
        x = (x - mn) / (mx - mn + 1e-5)
    mean = torch.tensor(cfg.pixel_mean).reshape(1, -1, 1, 1)
    std = torch.tensor(cfg.pixel_std).reshape(1, -1, 1, 1)
    return (x - mean) / std


def sinkhorn_plan(K: Tensor, u: Tensor, v: Tensor, thresh: float, max_iter: int) -> Tensor:
    """CustomCLIP.Sinkhorn (trainers/GLP_OT_SVLoRA.py:615-634): K [P, M, N], u [P, M], v [P, N]; ONE stopping test for
    the whole batch (the mean of |r - r0| over all P problems)."""
    r, c = torch.ones_like(u), torch.ones_like(v)
    for _ in range(max_iter):
        r0 = r
        r = u / torch.matmul(K, c.unsqueeze(-1)).squeeze(-1)
        c = v / torch.matmul(K.permute(0, 2, 1).contiguous(), r.unsqueeze(-1)).squeeze(-1)
        if float((r - r0).abs().mean()) < thresh:
            break
    return torch.matmul(r.unsqueeze(-1), c.unsqueeze(-2)) * K


def cot_plan(a: Tensor, b: Tensor, K: Tensor, thresh: float, max_iter: int) -> Tensor:
    """CustomCLIP.entropic_COT_fast (:636-675): the partial-transport scaling iterations on the Gibbs kernel K."""
    dx, dy = torch.ones_like(a), torch.ones_like(b)
    Kp = torch.matmul(torch.diag_embed(1 / a, dim1=1), K)
    Kq = torch.matmul(torch.diag_embed(1 / b, dim1=1), K.permute(0, 2, 1))
    u, v, cpt = dx, dy, 0
    while cpt < max_iter:
        v0 = v
        u = torch.minimum(torch.div(dx, torch.matmul(Kp, v.unsqueeze(-1)).squeeze(-1)), dx)
        v = torch.div(dy, torch.matmul(Kq, u.unsqueeze(-1)).squeeze(-1))
        cpt += 1
        if float((v - v0).abs().mean()) < thresh:
            break
    return torch.matmul(torch.matmul(torch.diag_embed(u, dim1=1), K), torch.diag_embed(v, dim1=1))


def clip_logits(sd: Dict[str, Tensor], image: Tensor, attr: Optional[Tensor], cfg, training: bool = True) -> Tensor:
    """CustomCLIP.forward (trainers/GLP_OT_SVLoRA.py:677-763) with cfg.ot in {'None', 'Sinkhorn', 'COT'}; the
    transport plan is built under no_grad, as in the reference.  `training` only matters for the BatchNorm layers of
    the RN50 trunk."""
    b = image.shape[0]
    x = preprocess(sd, image, cfg)
    feats = vision_forward(sd, x, attr, cfg, training)              # [L, B*S, d]
    feats = feats[1:]                                               # drop the class token (:696-697)
    M = feats.shape[0]
    text = text_forward(sd, cfg)                                    # [N, n_cls, d]
    feats = F.normalize(feats, dim=2)
    text = F.normalize(text, dim=2)
    sim = torch.einsum("mbd,ncd->mnbc", feats, text).contiguous()
    sim = sim.view(M, cfg.n_prompts, -1).permute(2, 0, 1)          # [B*S*n_cls, M, N]
    ot = getattr(cfg, "ot", "None")
    if ot == "None":
        sim_op = sim.mean(dim=(1, 2))
    else:
        N = cfg.n_prompts
        xx = torch.full((sim.shape[0], M), 1.0 / M, dtype=sim.dtype)
        yy = torch.full((sim.shape[0], N), 1.0 / N, dtype=sim.dtype)
        with torch.no_grad():
            KK = torch.exp(-(1.0 - sim) / cfg.ot_eps)
            if ot == "Sinkhorn":
                T = sinkhorn_plan(KK, xx, yy, cfg.ot_thresh, cfg.ot_max_iter)
            elif ot == "COT":
                yy = yy * min(float(torch.sum(xx)), cfg.ot_top_percent)           # (:724-726)
                T = cot_plan(xx, yy, KK, cfg.ot_thresh, cfg.ot_max_iter)
            else:
                raise NotImplementedError(ot)
        sim_op = torch.sum(T * sim, dim=(1, 2))
    sim_op = sim_op.contiguous().view(b, -1, cfg.n_cls).mean(1)     # average the slices (:753-754)
    return sd["logit_scale"].exp() * sim_op


# --------------------------------------------------------------------------
# Trainer step (forward_backward + SGD), metrics
# --------------------------------------------------------------------------
def _rank_auc(score: np.ndarray, pos: np.ndarray) -> float:
    """Mann-Whitney AUC with mid-ranks for ties == sklearn's trapezoidal ROC area."""
    p = np.asarray(score, dtype=np.float64)
    y = np.asarray(pos).astype(np.int64)
    n1 = int(y.sum())
    n0 = len(y) - n1
    order = np.argsort(p, kind="mergesort")
    ranks = np.empty(len(p), dtype=np.float64)
    sp = p[order]
    i = 0
    while i < len(sp):
        j = i
        while j + 1 < len(sp) and sp[j + 1] == sp[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y == 1].sum() - n1 * (n1 + 1) / 2.0) / (n0 * n1))


def auc_binary(prob: np.ndarray, label: Sequence[int]) -> float:
    """compute_auc for two classes (evaluation/metrics.py:340-356): the labels
    are one-hot encoded and roc_auc_score(average='macro') averages the AUC of
    each softmax column against its own indicator.  The two columns are fp32
    and ties in one need not be ties in the other, so both are ranked.
    A single-class batch reports 1 (trainers/GLP_OT_SVLoRA.py:965-967)."""
    prob = np.asarray(prob)
    y = np.asarray(label).astype(np.int64)
    if y.min() == y.max():
        return 1.0
    if prob.ndim == 1:
        return _rank_auc(prob, y == 1)
    return 0.5 * (_rank_auc(prob[:, 0], y == 0) + _rank_auc(prob[:, 1], y == 1))


def fairness_term(logits: Tensor, label: Tensor, attr: Tensor) -> Tensor:
    """The detached 'confidence' fairness term (trainers/GLP_OT_SVLoRA.py:930-944)."""
    probs = torch.softmax(logits, dim=1)
    correct = probs[torch.arange(len(label)), label]
    vals = [1 - correct[attr == g].mean() for g in torch.unique(attr)]
    vals = torch.tensor([float(v) for v in vals])
    return torch.mean(torch.abs(vals - vals.mean()))


class SgdState:
    """torch.optim.SGD(momentum, weight_decay, dampening=0) restated
    (Dassl/dassl/optim/optimizer.py:105-113): d = g + wd*p; first step
    buf = d, afterwards buf = mu*buf + d; p -= lr*buf."""

    def __init__(self, lr=1e-3, momentum=0.9, weight_decay=5e-4):
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.buf: Dict[str, Tensor] = {}

    def step(self, sd: Dict[str, Tensor], grads: Dict[str, Tensor], repeats: int = 1) -> None:
        """`repeats` = how many registered model names share the optimizer: TrainerBase.model_update calls
        optim.step() once per name on the same gradients (Dassl/dassl/engine/trainer.py:333-337), twice with
        UNFREEZE_IMAGE_ENCODER (trainers/GLP_OT_SVLoRA.py:866-870)."""
        for _ in range(repeats):
            for k, g in grads.items():
                d = g + self.weight_decay * sd[k]
                if k not in self.buf:
                    self.buf[k] = d.clone()
                else:
                    self.buf[k] = self.momentum * self.buf[k] + d
                sd[k] = sd[k] - self.lr * self.buf[k]


class StepLRState:
    """torch.optim.lr_scheduler.StepLR over an SgdState (Dassl/dassl/optim/lr_scheduler.py:100-115):
    lr = lr0 * gamma ** (last_epoch // step_size).  TrainerBase.update_lr steps the scheduler of EVERY registered
    model name (Dassl/dassl/engine/trainer.py:253-258); the two names of the FairLoRA run share one scheduler, so an
    epoch end advances it by two."""

    def __init__(self, opt: "SgdState", step_size: int, gamma: float = 0.1):
        self.opt, self.lr0, self.step_size, self.gamma, self.last_epoch = opt, opt.lr, step_size, gamma, 0

    def step(self, repeats: int = 1) -> None:
        self.last_epoch += repeats
        self.opt.lr = self.lr0 * self.gamma ** (self.last_epoch // self.step_size)


def loss_and_grads(sd: Dict[str, Tensor], batch, cfg, trainable: List[str], lambda_fairness: float = 0.0):
    """One forward + backward of GLP_OT_SVLoRA.forward_backward's fp32 branch
    (trainers/GLP_OT_SVLoRA.py:900-950): returns (loss, logits, grads)."""
    work = {k: v for k, v in sd.items()}
    leaves = {}
    for k in trainable:
        leaves[k] = sd[k].detach().clone().requires_grad_(True)
        work[k] = leaves[k]
    image, label = batch["img"], batch["label"]
    attr = batch["attrs"].t()[0] if "attrs" in batch and batch["attrs"] is not None else None
    logits = clip_logits(work, image, attr, cfg)
    cls_loss = F.cross_entropy(logits, label)
    loss = cls_loss
    if attr is not None and lambda_fairness != 0.0:
        loss = cls_loss + lambda_fairness * fairness_term(logits.detach(), label, attr)
    if not torch.isfinite(loss).all():
        raise FloatingPointError("Loss is infinite or NaN!")         # Dassl/dassl/engine/trainer.py:260-262
    loss.backward()
    grads = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])).detach()
             for k in trainable}
    return loss.detach(), logits.detach(), grads


def train_step(sd: Dict[str, Tensor], opt: SgdState, batch, cfg, trainable: List[str], optimizer_steps: int = 2,
               sched: Optional[StepLRState] = None, last_batch: bool = False):
    """forward_backward: loss -> backward -> model_update (-> update_lr after the epoch's last batch); returns the
    summary dict (trainers/GLP_OT_SVLoRA.py:959-973).  optimizer_steps = 2 is the reference's run configuration (both
    'prompt_learner' and 'image_encoder' registered with the one optimizer and scheduler)."""
    loss, logits, grads = loss_and_grads(sd, batch, cfg, trainable)
    opt.step(sd, grads, optimizer_steps)
    if sched is not None and last_batch:
        sched.step(optimizer_steps)
    prob = torch.softmax(logits, -1)
    acc = float((logits.argmax(-1) == batch["label"]).float().mean() * 100.0)
    auc = auc_binary(prob.numpy(), batch["label"].numpy())
    return {"loss": float(loss), "acc": acc, "auc": auc}, logits, grads


# --------------------------------------------------------------------------
# Round boundary: FedAvg + EMA (utils/fed_utils.py:42-100)
# --------------------------------------------------------------------------
def average_weights_ema(w_g: Dict[str, Tensor], w: Dict[int, Dict[str, Tensor]], idxs_users: Sequence[int],
                        n_client: Sequence[int], n_client_by_attr, epoch: int, max_epoch: int,
                        beta: float = 0.999, shared_half_s: bool = False) -> Dict[str, Tensor]:
    total = sum(n_client[r] for r in idxs_users)
    by_attr = None
    if n_client_by_attr is not None:
        by_attr = torch.tensor(n_client_by_attr)
        total_by_attr = by_attr[list(idxs_users)].sum(0)
    out: Dict[str, Tensor] = {}
    G = None if by_attr is None else by_attr.shape[1]
    for key in w[idxs_users[0]]:
        acc = None
        for u in idxs_users:
            f = n_client[u] / total
            x = w[u][key]
            if by_attr is not None and "lora_S" in key and x.shape[0] == G:
                term = x * (by_attr[u] / total_by_attr)[:, None]
            else:
                term = x * f
            acc = term if acc is None else acc + term
        if shared_half_s and by_attr is not None and "lora_S" in key and acc.shape[0] == G:
            g_, d_ = acc.shape
            acc = torch.cat([acc[:, : d_ // 2].mean(0, keepdim=True).repeat(g_, 1), acc[:, d_ // 2:]], dim=1)
        bd = beta * (epoch / max(max_epoch, 1))
        out[key] = (1 - bd) * acc + bd * w_g[key]
    return out
