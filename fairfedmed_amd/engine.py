"""FairLoRA local-training engine: one training step of the reference's
``GLP_OT_SVLoRA.forward_backward`` (trainers/GLP_OT_SVLoRA.py:883-975) as a
fixed sequence of HIP kernel launches over preallocated HBM buffers.

Data layout in HBM (all row-major):
  * token matrices [rows, width], row = image*L + token (image-major);
  * frozen weights twice, in the compute dtype: W [out, in] for the forward
    product and W^T [in, out] for the dX product (both are the "B[N,K]" operand
    of ffm_gemm_nt, K contiguous) — weights are frozen so W^T is built once;
  * every trainable tensor (prompt ctx, lora_A/S/B of the 24 wrapped linears)
    is a view into ONE flat fp32 buffer `flat`, with matching `grad` and
    `momentum` buffers: the SGD step is one kernel and the FedAvg exchange is
    one all-reduce of `flat`;
  * activations saved for the backward pass live in per-layer buffers sized
    once for `max_images` (no allocation inside a step).

Backward computes dX only (every dense weight is frozen): LoRA gradients come
from rank-r reductions (ffm_lora_grad_partial) and the LoRA dx term is an
epilogue of the dX GEMM, so the dense dW = A diag(s) B is never formed.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from ._lib import is16 as _is16
from . import ops
from .config import ModelCfg
from .synth import manifest, trainable_keys

Tensor = torch.Tensor


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class FlatParams:
    """All trainable tensors as views of one flat fp32 buffer (+ grad, momentum)."""

    def __init__(self, cfg: ModelCfg, device):
        shapes = manifest(cfg)
        self.keys = trainable_keys(cfg)
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for k in self.keys:
            shp = tuple(shapes[k])
            n = 1
            for s in shp:
                n *= s
            self.offsets[k] = (off, shp)
            off = _round_up(off + n, 4)
        self.numel = off
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.momentum = torch.zeros(off, device=device, dtype=torch.float32)
        self.steps = 0

    def view(self, key: str, which: str = "flat") -> Tensor:
        off, shp = self.offsets[key]
        n = 1
        for s in shp:
            n *= s
        return getattr(self, which)[off:off + n].view(shp)

    def load(self, sd: Dict[str, Tensor]) -> None:
        for k in self.keys:
            self.view(k).copy_(sd[k].to(self.flat.device, torch.float32))

    def lora_s_offsets(self) -> Tensor:
        return torch.tensor([self.offsets[k][0] for k in self.keys if k.endswith("lora_S.weight")],
                            dtype=torch.int64, device=self.flat.device)


class SOperands:
    """The singular-value operand S [Geff, r] (and the tensor that receives dS) of every adapter, by prefix
    ('...mlp.c_fc.'), for the adapter types of apply_lora_to_model (trainers/GLP_OT_SVLoRA.py:516-540):

      FairLoRA            lora_S [G, r] itself (a view of the flat buffer), gradient straight into its grad view;
      SVLoRA              lora_S [r] viewed as one group [1, r] (one diagonal shared by all samples, :307-311);
      LoRA                a constant row of ones (:241-242), dS discarded;
      ... + GLOBAL_S      s_b = pi_b S + S_global (:300-304, 418-422, 467-468).  The group mix sums to one, so this is
                          the same kernel call on S_eff[g] = S[g] + S_global, with dS[g] = dS_eff[g] and
                          dS_global = sum_g dS_eff[g]: S_eff is refreshed once per step (`prepare`) and the two gradients
                          are scattered after the reduction (`finish`), three small launches for all adapters together.
    """

    def __init__(self, params: FlatParams, prefixes: List[str], cfg: ModelCfg, device):
        lo = cfg.lora
        self.params, self.r = params, lo.rank
        self.type, self.glob = getattr(lo, "lora_type", "FairLoRA"), bool(getattr(lo, "global_s", False))
        if self.type not in ("FairLoRA", "SVLoRA", "LoRA"):
            raise NotImplementedError(self.type)                   # trainers/GLP_OT_SVLoRA.py:533-534
        if self.type != "FairLoRA" and lo.num_groups != 1:
            raise ValueError(f"lora_type {self.type} has no demographic groups: set num_groups = 1")
        self.G = lo.num_groups
        self.glob = self.glob and self.type != "LoRA"
        self.index = {p: i for i, p in enumerate(prefixes)}
        n, G, r = len(prefixes), self.G, self.r
        f32 = torch.float32
        if self.type == "LoRA":
            self.ones = torch.ones(1, r, device=device, dtype=f32)
            self.scratch = torch.zeros(n, 1, r, device=device, dtype=f32)
        if self.glob:
            base = torch.arange(G * r, device=device)
            self.idx_S = torch.stack([params.offsets[p + "lora_S.weight"][0] + base for p in prefixes])       # [n, G*r]
            self.idx_G = torch.stack([params.offsets[p + "lora_S_global.weight"][0] + base[:r] for p in prefixes])
            self.eff = torch.zeros(n, G, r, device=device, dtype=f32)
            self.deff = torch.zeros(n, G, r, device=device, dtype=f32)
        self.prefixes = prefixes

    def op(self, prefix: str) -> Tensor:
        i = self.index[prefix]
        if self.type == "LoRA":
            return self.ones
        if self.glob:
            return self.eff[i]
        return self.params.view(prefix + "lora_S.weight").view(self.G, self.r)

    def grad(self, prefix: str) -> Tensor:
        i = self.index[prefix]
        if self.type == "LoRA":
            return self.scratch[i]
        if self.glob:
            return self.deff[i]
        return self.params.view(prefix + "lora_S.weight", "grad").view(self.G, self.r)

    def prepare(self) -> None:
        if self.glob:
            flat = self.params.flat
            torch.add(flat[self.idx_S].view(-1, self.G, self.r), flat[self.idx_G].view(-1, 1, self.r), out=self.eff)

    def finish(self) -> None:
        if self.glob:
            g = self.params.grad
            g[self.idx_S.view(-1)] = self.deff.view(-1)
            g[self.idx_G.view(-1)] = self.deff.sum(1).view(-1)


def _ds_rows(rows: int, N: int, K: int, rank: int, dtype, packed: Optional[bool] = None, dgelu: bool = False) -> int:
    """dS partial rows a FairLoRA dX GEMM writes (packed=None: the larger of the kernels ffm_gemm_nt may pick)."""
    fl = L.EPI_LORA | L.EPI_LORA_KR | L.EPI_RANKOP | (L.EPI_DGELU if dgelu else 0)
    if packed is None:
        return max(ops.gemm_tiles_m(rows, N, K, fl, rank, dtype, p) for p in (False, True))
    return ops.gemm_tiles_m(rows, N, K, fl, rank, dtype, packed)


_PACKED = ("w_in", "w_in_t", "w_out", "w_out_t", "w_fc", "w_fc_t", "w_proj", "w_proj_t", "w_in_ln", "w_fc_ln")


@dataclass
class _Block:
    """Frozen weights of one residual attention block in the compute dtype."""
    w_in: Tensor
    w_in_t: Tensor
    b_in: Tensor
    w_out: Tensor
    w_out_t: Tensor
    b_out: Tensor
    ln1_w: Tensor
    ln1_b: Tensor
    ln2_w: Tensor
    ln2_b: Tensor
    w_fc: Tensor
    w_fc_t: Tensor
    b_fc: Tensor
    w_proj: Tensor
    w_proj_t: Tensor
    b_proj: Tensor
    lora: Optional[Dict[str, str]] = None      # role -> flat key, vision blocks only
    packed: Optional[Dict[str, Tensor]] = None # bf16 vision blocks: the eight weights in MFMA-fragment order (ops.pack_b)
    # bf16 vision blocks, ln_1 folded into the qkv product (FFM_EPI_LNIN): gamma-scaled in_proj weight, its row sums c
    # and d = W beta + b (frozen: built once at load time)
    w_in_ln: Optional[Tensor] = None
    c_in: Optional[Tensor] = None
    d_in: Optional[Tensor] = None
    # ... and ln_2 folded into the c_fc product
    w_fc_ln: Optional[Tensor] = None
    c_fc: Optional[Tensor] = None
    d_fc: Optional[Tensor] = None

    def pk(self, name: str) -> Optional[Tensor]:
        return self.packed[name] if self.packed else None


class _Stack:
    """A transformer tower (vision with FairLoRA, or text) with its saved activations."""

    def __init__(self, width: int, heads: int, layers: int, tokens: int, max_images: int, causal: bool,
                 rank: int, dtype, device, x3: bool = False):
        self.width, self.heads, self.layers, self.L = width, heads, layers, tokens
        self.causal, self.rank, self.dtype = causal, rank, dtype
        # FFM_TEXT_W16=1: the frozen weights of a float32 tower that runs beside a 16-bit vision tower (x3) as IEEE half in
        # memory (FFM_F32_X3_W16) - see FairLoRAEngine.__init__; measured: no gain, off by default
        self.wdtype = torch.float16 if (x3 and os.environ.get("FFM_TEXT_W16", "0") == "1") else dtype
        # x3: float32 tower whose products run on the bf16 matrix cores as hi/lo pairs (FFM_F32_X3)
        # (ops.gemm_nt is looked up per call: bench.py wraps it to time the launches)
        # FFM_GELU_DERIV=1: the MLP's saved tensor is quick_gelu'(pre) instead of pre (ffm_gemm_args.gelu_deriv): the forward's
        # epilogue has the sigmoid in hand and nothing but the dX product of c_proj reads the pre-activation.  Measured
        # (round 4, serial rocprofv3 of one call each): dX(c_proj) 48.86 -> 48.48 us, c_fc forward 47.18 -> 48.43 us, the step
        # 4.869 -> 4.874 ms (three alternating pairs): the backward epilogue is not bound by the derivative's exp + rcp, and
        # the forward pays for the extra arithmetic.  Off by default; bit-identical in fp32 either way.
        deriv = os.environ.get("FFM_GELU_DERIV", "0") == "1"

        # x3 (the text tower, 40 token rows): scratch for the products that are split over K across the grid
        # (ffm_gemm_args.sk_part, csrc/gemm_skinny.hip: N = 512 against K = 1536 / 2048 - c_proj forward, dX(c_fc), dX(qkv));
        # one buffer per tower: its launches are ordered on the tower's stream
        rows_max = max_images * tokens
        self.sk_part = None
        if x3 and rows_max <= 48:
            need = max(ops.gemm_splitk_floats(rows_max, n_, k_, self.wdtype == torch.float16)
                       for n_, k_ in ((width, 4 * width), (width, 3 * width), (4 * width, width), (3 * width, width), (width, width)))
            if need > 0:
                self.sk_part = torch.empty(need, device=device, dtype=torch.float32)

        def _gemm(*a, **k):
            if deriv and (k.get("gelu_out") is not None or k.get("dgelu_aux") is not None):
                k["gelu_deriv"] = True
            if self.sk_part is not None:
                k["sk_part"] = self.sk_part
            return ops.gemm_nt(*a, x3=x3, **k)
        self.gemm = _gemm
        self.blocks: List[_Block] = []
        T = max_images * tokens
        self.max_rows = T
        e = lambda *s: torch.zeros(*s, device=device, dtype=dtype)
        f = lambda *s: torch.zeros(*s, device=device, dtype=torch.float32)
        w = width
        self.x = [e(T, w) for _ in range(layers + 1)]        # block inputs; x[layers] = tower output
        self.xm = [e(T, w) for _ in range(layers)]           # after the attention residual
        self.qkv = [e(T, 3 * w) for _ in range(layers)]
        self.o = [e(T, w) for _ in range(layers)]
        self.lse = [f(max_images * heads * tokens) for _ in range(layers)]
        self.st1 = [(f(T), f(T)) for _ in range(layers)]
        self.st2 = [(f(T), f(T)) for _ in range(layers)]
        self.h2 = [e(T, w) for _ in range(layers)]
        self.pre = [e(T, 4 * w) for _ in range(layers)]
        self.act = [e(T, 4 * w) for _ in range(layers)]
        # partial row sums {sum, sum of squares} of every block input, left behind by its producer (FFM_EPI_ROWSTATS /
        # embed_lnpre) for the ln_1 that is folded into the qkv product: up to 8 column tiles
        self.rowp = [f(8 * T * 2) for _ in range(layers + 1)] if (rank and _is16(dtype)) else None
        self.rowp2 = [f(8 * T * 2) for _ in range(layers)] if (rank and _is16(dtype)) else None   # ... of xm (ln_2)
        # ln_2's BACKWARD folded into dX(c_proj) / dX(c_fc) (FFM_EPI_LNB_STAT / FFM_EPI_LNB_APPLY): the producer's partial row
        # sums, consumed by the very next launch - one buffer per tower; rows -> column tiles of the producer (0: not folded)
        # (ln_1's: two partial rows per head from the attention backward kernels, consumed by dX(qkv))
        self.lnb_part = f(max(8, 2 * heads) * T * 2) if (rank and _is16(dtype)) else None
        self.foldb = {}
        self.foldb1 = {}
        self.fold = {}                                       # rows -> (np of the c_proj forward, ok) decision cache
        self.fold2 = {}                                      # rows -> np of the out-proj forward (ln_2 into c_fc)
        if rank:
            self.t1 = [f(T, rank) for _ in range(layers)]
            self.ts1 = [f(T, rank) for _ in range(layers)]
            self.t2 = [f(T, rank) for _ in range(layers)]
            self.ts2 = [f(T, rank) for _ in range(layers)]
        # transient buffers
        self.h = e(T, w)
        self.g = e(T, w)          # running gradient w.r.t. the residual stream
        self.g1 = e(T, w)
        self.dh = e(T, w)
        self.do = e(T, w)
        self.dqkv = e(T, 3 * w)
        self.dpre = e(T, 4 * w)
        self.delta = f(max_images * heads * tokens)
        if rank:
            self.u = f(T, rank)
            # The LoRA-gradient reductions trail the dX chain on a side stream, so everything they
            # read is kept per layer: gradient of the block output, dL/d(pre), and both us tensors.
            self.g_l = [e(T, w) for _ in range(layers)]
            self.dpre_l = [e(T, 4 * w) for _ in range(layers)]
            self.us2 = [f(T, rank) for _ in range(layers)]
            self.us1 = [f(T, rank) for _ in range(layers)]
            # per-layer partial sums of the LoRA gradients; all of them are reduced by ONE launch
            # (ffm_reduce_partials_multi) at the end of the backward pass
            ns = ops.lora_grad_splits(T)
            nb = max(ops.lora_down_blocks_max(T, w, rank, dtype), ops.lora_down_blocks_max(T, 4 * w, rank, dtype),
                     _ds_rows(T, 4 * w, w, rank, dtype, dgelu=True), _ds_rows(T, w, 4 * w, rank, dtype))
            self.part = [{"fc_A": f(ns * w * rank), "fc_B": f(ns * 4 * w * rank),
                          "proj_A": f(ns * 4 * w * rank), "proj_B": f(ns * w * rank),
                          "fc_S": f(nb * 8 * rank), "proj_S": f(nb * 8 * rank)} for _ in range(layers)]
            self.plans = {}
            self.lgrad = {}                                      # rows -> row tiles of the FFM_EPI_LGRAD partials (0: not served)


class FairLoRAEngine:
    """HIP execution engine for CustomCLIP(+FairLoRA) training and inference."""

    @property
    def grad_scale(self) -> float:
        """fp16 mode: the factor on dloss/dlogits in front of the backward pass, taken out of the fp32 gradients behind it.
        It lives in DEVICE memory (``scale_state``, ffm_loss_scale in include/ffm_hip.h) and moves by itself: an overflowed
        backward pass skips its SGD step and halves it.  Reading it synchronises; 1.0 in the other storage types."""
        return float(self.scale_state[0]) if self.scale_state is not None else 1.0

    @grad_scale.setter
    def grad_scale(self, value: float) -> None:
        value = float(value)
        if not (value > 0.0) or value != value or value == float("inf"):
            raise ValueError(f"grad_scale must be a positive finite number, got {value!r}")
        if self.scale_state is None:
            if value != 1.0:
                raise ValueError("grad_scale exists in the IEEE-half mode only")
            return
        # recorded plans read the scale through the state's address: nothing to re-record
        self.scale_state[:2].copy_(torch.tensor([value, 1.0 / value]))
        self.scale_state[5:6].fill_(max(value, float(self.scale_state[5])))

    def overflow_steps(self) -> int:
        """fp16 mode: training steps skipped so far because their scaled gradients left half's range (host sync)."""
        return int(self.scale_state[4]) if self.scale_state is not None else 0

    def __init__(self, cfg: ModelCfg, state_dict: Dict[str, Tensor], dtype=torch.bfloat16, max_images: int = 32,
                 device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("FairLoRAEngine needs an MI355X GPU; there is no CPU path")
        from . import _lib
        _lib.load()                                   # fail loudly if the HIP library is missing
        if cfg.vision.head_dim != 64 or cfg.text.width // cfg.text.heads != 64:
            raise ValueError("attention kernels are built for head_dim 64 (all CLIP towers)")
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self.max_images = max_images
        v, t = cfg.vision, cfg.text
        self.params = FlatParams(cfg, self.device)
        self.params.load(state_dict)
        self.n_text = cfg.n_prompts * cfg.n_cls
        # Text tower: the mask is causal and only the EOT row of each prompt is read (clip/model.py:562-568,
        # trainers/GLP_OT_SVLoRA.py:62-64), so tokens after the last EOT position influence neither the features nor
        # the ctx gradient.  The tower runs on the first max(eot)+1 tokens of the 77 (10 for the four FairFedMed
        # prompts): bit-identical rows, 7.7x fewer of them, and far less interference with the vision chain.
        self.txt_len = min(t.context_length, max(cfg.eot) + 1)
        assert self.txt_len >= 1 + cfg.n_ctx
        # the text-tail kernels index x[eot_row[p]] without a range check (csrc/text.hip): an EOT position outside the
        # rows the tower keeps would read another prompt's row
        if not all(0 <= int(e) < self.txt_len for e in cfg.eot):
            raise ValueError(f"EOT positions {list(cfg.eot)} must lie inside the {self.txt_len} text rows that are kept "
                             f"(context_length {t.context_length})")
        # ... and ALWAYS in float32, also in the bf16 throughput mode: the two classes' prompts differ in a few tokens, so
        # the logit difference l1 - l0 = e^ls <f, t1 - t0> rides on the small difference of two nearly equal text features,
        # and 2^-9 roundings of the text activations AND of the text weights land on it many times amplified.  Measured on
        # the tiny model (tools/auc_diag.py, AUC after equal rounds against the reference): text tower in bf16 0.0026 off,
        # f32 activations on bf16 weights 0.0020, all f32 0.0005.  Beside a bf16 vision tower the 40-row products run on
        # the bf16 matrix cores as hi/lo pairs (FFM_F32_X3, csrc/gemm_skinny.hip) instead of the 16x slower f32 MFMA.
        # Round 4, FFM_TEXT_W16=1: the FROZEN weights of the tower as IEEE half (FFM_F32_X3_W16: 11 significant bits, what
        # the reference's own PREC="fp16" holds, clip/model.py:609-630; split exactly into the bf16 hi + lo pair in the
        # kernel), 151 instead of 302 MB of text weights per step beside the vision chain; activations, gradients and
        # accumulation stay float32.  Measured (three alternating pairs, one call): 4.590 against 4.592 ms per step -
        # with two column tiles per block the weights are not what the chain feels any more.
        # Off by default: float32 weights.
        self.txt = _Stack(t.width, t.heads, t.layers, self.txt_len, self.n_text, True, 0, torch.float32, self.device,
                          x3=(_is16(dtype)))
        dev, f32 = self.device, torch.float32
        self._init_vision(max_images)                 # tower-specific buffers (ViT here, RN50 in engine_rn.py)
        self.load_frozen(state_dict)
        self._init_vision_late()
        self.feat = torch.zeros(max_images * v.tokens, v.out_dim, device=dev, dtype=dtype)
        self.dfeat = torch.zeros_like(self.feat)
        self.fbar = torch.zeros(max_images, v.out_dim, device=dev, dtype=f32)
        self.rnorm = torch.zeros(max_images * v.tokens, device=dev, dtype=f32)
        self.logits_img = torch.zeros(max_images, cfg.n_cls, device=dev, dtype=f32)
        self.dlogits_img = torch.zeros_like(self.logits_img)
        self.logits = torch.zeros(max_images, cfg.n_cls, device=dev, dtype=f32)
        self.prob = torch.zeros_like(self.logits)
        self.loss = torch.zeros(1, device=dev, dtype=f32)
        self.finite = torch.ones(1, device=dev, dtype=torch.int32)
        # IEEE-half storage: dloss/dlogits is scaled by 2^k before the backward pass (every backward kernel is linear in
        # the incoming gradient) and the factor comes out of the fp32 gradient buffer in front of the SGD step.  Unscaled,
        # the 16-bit activation gradients of ViT-B/16 sit in half's subnormals (2.8 % error on the lora_S gradient norms
        # at batch 8; 0.3 % scaled: tests/test_engine_gpu.py).  Round 6: the scale is DYNAMIC and device-resident
        # (ops.loss_scale / unscale_check / sgd_momentum_gated): a backward pass whose scaled gradients overflow skips its
        # SGD step and halves the scale - no host sync, recorded plans stay valid - instead of raising; after
        # FFM_F16_GROWTH_INTERVAL (2000) good steps it doubles again, up to the initial value FFM_F16_GRAD_SCALE (4096).
        self.scale_state = None
        if dtype == torch.float16:
            s0 = float(os.environ.get("FFM_F16_GRAD_SCALE", "4096"))
            self.scale_state = torch.tensor([s0, 1.0 / s0, 1.0, 0.0, 0.0, s0, float(os.environ.get("FFM_F16_GROWTH_INTERVAL", "2000")), 1.0],
                                            device=dev, dtype=f32)
        self.dtbar = torch.zeros(cfg.n_cls, v.out_dim, device=dev, dtype=f32)
        self.attr_i32 = torch.zeros(max_images, device=dev, dtype=torch.int32)
        self.label_buf = torch.zeros(max_images, device=dev, dtype=torch.int64)
        self._counts_buf = None                       # enable_step_counts()
        self.tbar_buf = torch.zeros(cfg.n_cls, v.out_dim, device=dev, dtype=f32)
        self.ot = cfg.ot if cfg.ot != "None" else None
        if self.ot:
            # Sinkhorn / COT logits heads (trainers/GLP_OT_SVLoRA.py:615-675, 713-757; csrc/head_ot.hip)
            if self.ot not in ops.OT_MODES:
                raise NotImplementedError(cfg.ot)
            P, M, N = max_images * cfg.n_cls, v.tokens - 1, cfg.n_prompts
            self.dtn = torch.zeros(N * cfg.n_cls, v.out_dim, device=dev, dtype=f32)
            self.ot_sim = torch.zeros(P * M * N, device=dev, dtype=f32)
            self.ot_T = torch.zeros_like(self.ot_sim)
            self.ot_errs = torch.zeros(cfg.ot_max_iter * P, device=dev, dtype=f32)
            self.ot_istop = torch.zeros(1, device=dev, dtype=torch.int32)
            self.ot_tsum = torch.zeros(P, device=dev, dtype=f32)
            self.ot_dtn_part = torch.zeros(max_images * N * cfg.n_cls * v.out_dim, device=dev, dtype=f32)
        self.step_plans: Dict[tuple, list] = {}
        self.use_replay = True                        # replay recorded launch plans (host-side "graph")
        # the two ends of the text tower (csrc/text.hip): absolute EOT row of every prompt, the un-normalised / normalised
        # text features, 1 / norm and ln_final's statistics kept for the way back, the projection's input gradient
        self.eot_rows = torch.tensor([i * self.txt_len + cfg.eot[i % cfg.n_cls] for i in range(self.n_text)],
                                     device=dev, dtype=torch.int32)
        self.tf_buf = torch.zeros(self.n_text, v.out_dim, device=dev, dtype=f32)
        self.tn_all = torch.zeros(self.n_text, v.out_dim, device=dev, dtype=f32)
        self.t_rnorm = torch.zeros(self.n_text, device=dev, dtype=f32)
        self.t_stats = torch.zeros(self.n_text, 2, device=dev, dtype=f32)
        self.t_dy = torch.zeros(self.n_text, t.width, device=dev, dtype=f32)
        if self.ot:
            self.tn_buf = self.tn_all                     # the transport heads read every prompt's normalised feature
        # The text tower (308 token rows) is latency-bound and independent of the vision tower until the
        # logits head, so it runs on its own HIP stream beside it (forward and backward).
        self.side = self._side0 = torch.cuda.Stream(device=self.device)
        self.grad_stream = self._grad0 = torch.cuda.Stream(device=self.device)
        self.ev_layer = [torch.cuda.Event() for _ in range(self._n_layer_events())]
        self.ev_grads = torch.cuda.Event()
        self.ev_tail = torch.cuda.Event()
        self.ev_tail0 = torch.cuda.Event()
        self.ev_text_fwd = torch.cuda.Event()
        self.ev_head_bwd = torch.cuda.Event()
        self.ev_text_bwd = torch.cuda.Event()
        self.ev_start = torch.cuda.Event()
        self.ev_pack = torch.cuda.Event()
        self._pack_event = None

    # ---------------------------------------------------- vision tower hooks --
    def _init_vision(self, max_images: int) -> None:
        cfg, v, dtype, dev = self.cfg, self.cfg.vision, self.dtype, self.device
        self.vis = _Stack(v.width, v.heads, v.layers, v.tokens, max_images, False, cfg.lora.rank, dtype, dev)
        P = v.grid * v.grid
        self.cols = torch.zeros(max_images * P, 3 * v.patch * v.patch, device=dev, dtype=dtype)
        self.patch_out = torch.zeros(max_images * P, v.width, device=dev, dtype=dtype)
        self.hpost = torch.zeros(max_images * v.tokens, v.width, device=dev, dtype=dtype)
        self.post_stats = (torch.zeros(max_images * v.tokens, device=dev), torch.zeros(max_images * v.tokens, device=dev))
        f32 = torch.float32
        self.is3d = cfg.dim_per_3d_slice > 0
        if self.is3d:
            D, H = cfg.dim_per_3d_slice, v.image_size
            nblk = ops.slice_blocks(H, H)
            self.conv_out = torch.zeros(max_images, 3, H, H, device=dev, dtype=f32)
            self.dconv = torch.zeros_like(self.conv_out)
            self.mm_part = torch.zeros(max_images * nblk * 2, device=dev, dtype=f32)
            self.mnmx = torch.zeros(max_images, 2, device=dev, dtype=f32)
            self.mm_cnt = torch.zeros(max_images, 2, device=dev, dtype=torch.int32)
            self.ab_part = torch.zeros(max_images * ops.slice_bwd_ab_blocks() * 2, device=dev, dtype=f32)
            self.gmm = torch.zeros(max_images, 2, device=dev, dtype=f32)
            self.wpart = torch.zeros(max_images * ops.slice_wgrad_blocks(H, H) * (3 * D * 25 + 3), device=dev, dtype=f32)
            self.dcols = torch.zeros_like(self.cols)
            self.dpatch = torch.zeros_like(self.patch_out)

    def _init_vision_late(self) -> None:
        """After the frozen weights are loaded: the packed rank operands of the FairLoRA GEMMs."""
        cfg, v, dtype, dev = self.cfg, self.cfg.vision, self.dtype, self.device
        # FairLoRA down projections ride inside the GEMMs (FFM_EPI_RANKOP) when the rank fits one MFMA tile
        self.fused_rank = 0 < cfg.lora.rank <= 16
        # where a block's LoRA-gradient reductions start (_stack_backward); unset: by row count
        self.red_at = int(os.environ["FFM_RED_AT"]) if "FFM_RED_AT" in os.environ else None
        self.use_lgrad = os.environ.get("FFM_LGRAD", "1") != "0"    # the two large reductions inside the dX product of c_proj
        ie = "image_encoder.transformer.resblocks."
        self.sops = SOperands(self.params, [f"{ie}{i}.mlp.c_{n}." for i in range(v.layers) for n in ("fc", "proj")],
                              cfg, dev)
        if self.fused_rank:
            w = v.width
            ent = []
            self.rk = []
            self.lw_wide = []
            for blk in self.vis.blocks:
                pk = {"fc_A": torch.zeros(16, w, device=dev, dtype=dtype),
                      "proj_A": torch.zeros(16, 4 * w, device=dev, dtype=dtype),
                      "fc_B": torch.zeros(16, 4 * w, device=dev, dtype=dtype),
                      "proj_B": torch.zeros(16, w, device=dev, dtype=dtype)}
                self.rk.append(pk)
                # the same matrices as [K, 32] rows: the `lw` operand of the GEMM whose OUTPUT columns they span
                wd = {role: torch.zeros(buf.shape[1], 32, device=dev, dtype=dtype) for role, buf in pk.items()} \
                    if _is16(dtype) else {}
                self.lw_wide.append(wd)
                for role, buf in pk.items():
                    # ln_2's backward fold (_fold_ln2_bwd): the rank operand of dX(c_fc) carries W gamma and W beta + b as rows
                    # 14 / 15 - its t[14] / t[15] are the two row sums against those vectors, out of the matrix cores; every
                    # consumer masks the slots beyond r (rank <= 14)
                    extra = (blk.c_fc, blk.d_fc) if (role == "fc_B" and _is16(dtype) and blk.c_fc is not None and cfg.lora.rank <= 14) else None
                    ent.append((self.params.view(blk.lora[role]), role.endswith("_B"), buf, wd.get(role), None, extra))
                if _is16(dtype):
                    # ln_2 folded into c_fc: the rank operand gamma-scaled, and its two correction rows (ops.LnIn.rk)
                    pk["fc_A_ln"] = torch.zeros(16, w, device=dev, dtype=dtype)
                    pk["fc_A_lnrk"] = torch.zeros(32, device=dev, dtype=torch.float32)
                    ent.append((self.params.view(blk.lora["fc_A"]), False, pk["fc_A_ln"], None,
                                (blk.ln2_w, blk.ln2_b, pk["fc_A_lnrk"])))
            self.pack_plan = ops.PackPlan(ent, dtype, dev)

    def _n_layer_events(self) -> int:
        return self.cfg.vision.layers

    # ------------------------------------------------------------ weights --
    def _w(self, x: Tensor, dtype=None) -> Tensor:
        return ops.cast_from_f32(x.to(self.device, torch.float32).contiguous(), dtype or self.dtype)

    def _wt(self, x: Tensor, dtype=None) -> Tensor:
        return ops.transpose_cast(x.to(self.device, torch.float32).contiguous(), dtype or self.dtype)

    def _f(self, x: Tensor) -> Tensor:
        return x.to(self.device, torch.float32).contiguous().clone()

    def _load_stack(self, stack: _Stack, sd, prefix: str, lora: bool) -> None:
        old = stack.blocks if stack.blocks else None
        W = lambda x: self._w(x, stack.wdtype)
        WT = lambda x: self._wt(x, stack.wdtype)
        stack.blocks = []
        for i in range(stack.layers):
            p = f"{prefix}transformer.resblocks.{i}."
            fc = "mlp.c_fc.original_linear." if lora else "mlp.c_fc."
            pj = "mlp.c_proj.original_linear." if lora else "mlp.c_proj."
            blk = _Block(
                w_in=W(sd[p + "attn.in_proj_weight"]), w_in_t=WT(sd[p + "attn.in_proj_weight"]),
                b_in=self._f(sd[p + "attn.in_proj_bias"]),
                w_out=W(sd[p + "attn.out_proj.weight"]), w_out_t=WT(sd[p + "attn.out_proj.weight"]),
                b_out=self._f(sd[p + "attn.out_proj.bias"]),
                ln1_w=self._f(sd[p + "ln_1.weight"]), ln1_b=self._f(sd[p + "ln_1.bias"]),
                ln2_w=self._f(sd[p + "ln_2.weight"]), ln2_b=self._f(sd[p + "ln_2.bias"]),
                w_fc=W(sd[p + fc + "weight"]), w_fc_t=WT(sd[p + fc + "weight"]),
                b_fc=self._f(sd[p + fc + "bias"]),
                w_proj=W(sd[p + pj + "weight"]), w_proj_t=WT(sd[p + pj + "weight"]),
                b_proj=self._f(sd[p + pj + "bias"]),
            )
            if lora and _is16(stack.dtype):
                w32 = sd[p + "attn.in_proj_weight"].to(self.device, torch.float32)
                g1, b1 = blk.ln1_w, blk.ln1_b
                blk.w_in_ln = W(w32 * g1[None, :])
                blk.c_in = blk.w_in_ln.float().sum(1).contiguous()        # row sums of the weight AS ROUNDED
                blk.d_in = (w32 @ b1 + blk.b_in).contiguous()
                wf32 = sd[p + fc + "weight"].to(self.device, torch.float32)
                blk.w_fc_ln = W(wf32 * blk.ln2_w[None, :])
                blk.c_fc = blk.w_fc_ln.float().sum(1).contiguous()
                blk.d_fc = (wf32 @ blk.ln2_b + blk.b_fc).contiguous()
            if lora:
                blk.lora = {f"{n}_{m}": f"{p}mlp.c_{n}.lora_{m}.weight" for n in ("fc", "proj") for m in "ASB"}
            if old is not None:
                # keep the device addresses stable (recorded launch plans hold raw pointers): refresh in place
                for name, val in vars(blk).items():
                    if isinstance(val, torch.Tensor):
                        getattr(old[i], name).copy_(val)
                blk = old[i]
            if lora and _is16(stack.dtype):
                # frozen weights in MFMA-fragment order for the panel GEMM (csrc/gemm_panel_impl.h)
                if blk.packed is None:
                    blk.packed = {}
                for name in _PACKED:
                    blk.packed[name] = ops.pack_b(getattr(blk, name), blk.packed.get(name))
            stack.blocks.append(blk)

    def load_frozen(self, sd: Dict[str, Tensor]) -> None:
        """(Re)build the compute-dtype copies of every frozen tensor."""
        te = "text_encoder."
        self._load_stack(self.txt, sd, te, False)
        def put(name, val):                                       # stable addresses across reloads
            cur = getattr(self, name, None)
            if cur is None:
                setattr(self, name, val)
            elif isinstance(val, tuple):
                for c, n in zip(cur, val):
                    c.copy_(n)
            else:
                cur.copy_(val)

        self._load_vision_frozen(sd, put)
        put("logit_scale", self._f(sd["logit_scale"].reshape(1)))
        # text side constants stay fp32 (tiny): prompt pieces, ln_final, projection
        put("tok_prefix", self._f(sd["prompt_learner.token_prefix"]))
        put("tok_suffix", self._f(sd["prompt_learner.token_suffix"]))
        put("txt_pos", self._f(sd[te + "positional_embedding"]))
        put("lnfinal", (self._f(sd[te + "ln_final.weight"]), self._f(sd[te + "ln_final.bias"])))
        put("text_proj", self._f(sd[te + "text_projection"]))

    def _load_vision_frozen(self, sd: Dict[str, Tensor], put) -> None:
        cfg, v = self.cfg, self.cfg.vision
        ie = "image_encoder."
        self._load_stack(self.vis, sd, ie, True)
        put("conv_w", self._w(sd[ie + "conv1.weight"].reshape(v.width, -1)))
        if cfg.dim_per_3d_slice:
            put("conv_w_t", self._wt(sd[ie + "conv1.weight"].reshape(v.width, -1)))   # dX of the patch embedding
        put("cls", self._w(sd[ie + "class_embedding"]))
        put("pos", self._w(sd[ie + "positional_embedding"]))
        put("lnpre", (self._f(sd[ie + "ln_pre.weight"]), self._f(sd[ie + "ln_pre.bias"])))
        put("lnpost", (self._f(sd[ie + "ln_post.weight"]), self._f(sd[ie + "ln_post.bias"])))
        put("proj", self._w(sd[ie + "proj"]))                     # [width, out]: B operand of dh = df proj^T
        put("proj_t", self._wt(sd[ie + "proj"]))                  # [out, width]: B operand of f = h proj
        # the three frozen weights outside the blocks in MFMA-fragment order too (16-bit modes): patch embedding and the
        # two products of the final projection take the panel kernel where their shape fills the chip
        self.pk_out = getattr(self, "pk_out", {})
        if _is16(self.dtype) and os.environ.get("FFM_PACK_OUT", "1") != "0":
            for name in ("conv_w", "proj", "proj_t"):
                w = getattr(self, name)
                if w.shape[0] % 16 == 0 and w.shape[1] % 32 == 0:
                    self.pk_out[name] = ops.pack_b(w, self.pk_out.get(name))

    # -------------------------------------------------------------- tower --
    def _lora_view(self, blk: _Block, role: str) -> Tensor:
        return self.params.view(blk.lora[role])

    def _S(self, layer: int, which: str) -> Tensor:
        return self.sops.op(f"image_encoder.transformer.resblocks.{layer}.mlp.c_{which}.")

    def _dS(self, layer: int, which: str) -> Tensor:
        return self.sops.grad(f"image_encoder.transformer.resblocks.{layer}.mlp.c_{which}.")

    def _fold_ln1(self, st: _Stack, rows: int) -> int:
        """0: ln_1 runs as its own kernel.  Otherwise the number of partial row sums the c_proj forward of this row
        count leaves per row (its column tiles): both that product and the qkv product are served by the panel kernel
        with the folding epilogues (bf16, fused rank, frozen weights packed)."""
        if st.rowp is None or not getattr(self, "fused_rank", False) or getattr(self, "no_ln_fold", False):
            return 0
        if rows not in st.fold:
            w, r = st.width, st.rank
            f_proj = L.EPI_BIAS | L.EPI_LORA | L.EPI_RESIDUAL | L.EPI_RANKOP | L.EPI_ROWSTATS
            npj = ops.gemm_tiles_n(rows, w, 4 * w, f_proj, r, st.dtype, True)
            nq = ops.gemm_tiles_n(rows, 3 * w, w, L.EPI_BIAS | L.EPI_LNIN, 0, st.dtype, True)
            st.fold[rows] = npj if (npj > 0 and npj <= 8 and nq > 0) else 0
        return st.fold[rows]

    def _fold_ln2(self, st: _Stack, rows: int) -> int:
        """The same for ln_2 in front of c_fc: partial row sums from the out-proj forward, the FairLoRA c_fc product with
        the folding epilogue (its rank operand gamma-scaled), and dA(c_fc) from the raw rows (ffm_lora_grad_partial_ln)."""
        if st.rowp2 is None or not getattr(self, "fused_rank", False) or getattr(self, "no_ln_fold", False):
            return 0
        if rows not in st.fold2:
            w, r = st.width, st.rank
            npo = ops.gemm_tiles_n(rows, w, w, L.EPI_BIAS | L.EPI_RESIDUAL | L.EPI_ROWSTATS, 0, st.dtype, True)
            f_fc = L.EPI_BIAS | L.EPI_LORA | L.EPI_GELU | L.EPI_RANKOP | L.EPI_LNIN
            nf = ops.gemm_tiles_n(rows, 4 * w, w, f_fc, r, st.dtype, True)
            st.fold2[rows] = npo if (npo > 0 and npo <= 8 and nf > 0 and w % 128 == 0 and r <= 16) else 0
        return st.fold2[rows]

    def _fold_ln2_bwd(self, st: _Stack, rows: int, blk: _Block) -> int:
        """ln_2's BACKWARD folded into the two dX products around it (round 6; include/ffm_hip.h FFM_EPI_LNB_*): 0 - the
        LayerNorm backward runs as its own kernel; otherwise the column tiles of the dX product of c_proj, which leaves the
        two row sums beside its FFM_EPI_LGRAD partial products, and the dX product of c_fc stores dL/d x_mid directly.
        Needs the forward fold (its W gamma / W beta + b / A^T gamma / A^T beta vectors), the LGRAD epilogue and both
        kernels for this row count.  FFM_LNB_FOLD=0: off (A/B runs)."""
        if st.lnb_part is None or os.environ.get("FFM_LNB_FOLD", "1") == "0":
            return 0
        if rows not in st.foldb:
            w, r = st.width, st.rank
            n = 0
            if self._fold_ln2(st, rows) and self._lgrad_rows(st, rows, blk) > 0 and r <= 14:
                f1 = L.EPI_LORA | L.EPI_LORA_KR | L.EPI_DGELU | L.EPI_RANKOP | L.EPI_LGRAD | L.EPI_LNB_STAT
                f2 = L.EPI_LORA | L.EPI_LORA_KR | L.EPI_RANKOP | L.EPI_LNB_APPLY
                n1 = ops.gemm_tiles_n(rows, 4 * w, w, f1, r, st.dtype, True)
                n2 = ops.gemm_tiles_n(rows, w, 4 * w, f2, r, st.dtype, True)
                n = n1 if (0 < n1 <= 8 and n2 > 0) else 0
            st.foldb[rows] = n
        return st.foldb[rows]

    def _fold_ln1_bwd(self, st: _Stack, rows: int) -> int:
        """ln_1's BACKWARD folded into the attention backward (row sums per head: ffm_attention_bwd_lnstat) and the dX product
        of the in-projection (FFM_EPI_LNB_APPLY on the plain panel tile): 0, or the number of partial rows (2 heads)."""
        if st.lnb_part is None or os.environ.get("FFM_LNB_FOLD", "1") == "0" or os.environ.get("FFM_LNB_FOLD1", "1") == "0":
            return 0
        if rows not in st.foldb1:
            w = st.width
            n = 0
            if self._fold_ln1(st, rows) and 2 * st.heads <= 24 and ops.attention_bwd_lnstat_ok(st.L, st.causal, st.dtype):
                if ops.gemm_tiles_n(rows, w, 3 * w, L.EPI_LNB_APPLY, 0, st.dtype, True) > 0:
                    n = 2 * st.heads
            st.foldb1[rows] = n
        return st.foldb1[rows]

    def _stack_forward(self, st: _Stack, rows: int, images: int, attr: Optional[Tensor], rows_per_sample: int,
                       save: bool = True) -> Tensor:
        """x[0][:rows] holds the tower input; returns the tower output view."""
        lo = self.cfg.lora
        r, G = st.rank, lo.num_groups
        gemm = st.gemm
        for i, blk in enumerate(st.blocks):
            x, xm = st.x[i][:rows], st.xm[i][:rows]
            qkv, o, h2 = st.qkv[i][:rows], st.o[i][:rows], st.h2[i][:rows]
            pre, act = st.pre[i][:rows], st.act[i][:rows]
            h = st.h[:rows]
            fold = self._fold_ln1(st, rows)
            if fold:
                # ln_1 rides inside the qkv product: raw rows x gamma-scaled weight, normalised in the epilogue with the
                # row sums the producer of x left behind (block 0: embed_lnpre, else the previous block's c_proj)
                ln = ops.LnIn(st.rowp[i], 1 if i == 0 else fold, blk.c_in, st.st1[i][0], st.st1[i][1])
                gemm(x, blk.w_in_ln, qkv, bias=blk.d_in, b_packed=blk.pk("w_in_ln"), ln_in=ln)
            else:
                ops.layernorm_fwd(x, h, blk.ln1_w, blk.ln1_b, st.st1[i][0], st.st1[i][1])
                gemm(h, blk.w_in, qkv, bias=blk.b_in, b_packed=blk.pk("w_in"))
            ops.attention_fwd(qkv, o, st.lse[i], images, st.L, st.heads, st.causal)
            fold2 = self._fold_ln2(st, rows) if r else 0
            gemm(o, blk.w_out, xm, bias=blk.b_out, res=x, b_packed=blk.pk("w_out"), rowstats=st.rowp2[i] if fold2 else None)
            if not fold2:
                ops.layernorm_fwd(xm, h2, blk.ln2_w, blk.ln2_b, st.st2[i][0], st.st2[i][1])
            if r and self.fused_rank:
                if fold2:
                    # ln_2 rides inside the c_fc product (h2 is never written; dA(c_fc) takes the raw rows, _stack_backward)
                    ro = ops.RankOp(self.rk[i]["fc_A_ln"], self._S(i, "fc"), attr, rows_per_sample, lo.scaling,
                                    lo.lambda_group, t_out=st.t1[i], ts_out=st.ts1[i], lw_wide=self.lw_wide[i].get("fc_B"))
                    ln = ops.LnIn(st.rowp2[i], fold2, blk.c_fc, st.st2[i][0], st.st2[i][1], rk=self.rk[i]["fc_A_lnrk"])
                    gemm(xm, blk.w_fc_ln, pre, bias=blk.d_fc, lw=self._lora_view(blk, "fc_B"), gelu_out=act, rankop=ro,
                         b_packed=blk.pk("w_fc_ln"), ln_in=ln)
                else:
                    ro = ops.RankOp(self.rk[i]["fc_A"], self._S(i, "fc"), attr, rows_per_sample, lo.scaling,
                                    lo.lambda_group, t_out=st.t1[i], ts_out=st.ts1[i], lw_wide=self.lw_wide[i].get("fc_B"))
                    gemm(h2, blk.w_fc, pre, bias=blk.b_fc, lw=self._lora_view(blk, "fc_B"), gelu_out=act, rankop=ro,
                         b_packed=blk.pk("w_fc"))
                ro = ops.RankOp(self.rk[i]["proj_A"], self._S(i, "proj"), attr, rows_per_sample,
                                lo.scaling, lo.lambda_group, t_out=st.t2[i], ts_out=st.ts2[i],
                                lw_wide=self.lw_wide[i].get("proj_B"))
                gemm(act, blk.w_proj, st.x[i + 1][:rows], bias=blk.b_proj, lw=self._lora_view(blk, "proj_B"),
                     res=xm, rankop=ro, b_packed=blk.pk("w_proj"),
                     rowstats=st.rowp[i + 1] if (fold and i + 1 < st.layers) else None)
            elif r:
                ops.lora_down(h2, self._lora_view(blk, "fc_A"), False, self._S(i, "fc"), attr, r, G,
                              rows_per_sample, lo.scaling, lo.lambda_group, st.t1[i], st.ts1[i])
                gemm(h2, blk.w_fc, pre, bias=blk.b_fc, ts=st.ts1[i], lw=self._lora_view(blk, "fc_B"),
                            gelu_out=act)
                ops.lora_down(act, self._lora_view(blk, "proj_A"), False, self._S(i, "proj"), attr, r, G,
                              rows_per_sample, lo.scaling, lo.lambda_group, st.t2[i], st.ts2[i])
                gemm(act, blk.w_proj, st.x[i + 1][:rows], bias=blk.b_proj, ts=st.ts2[i],
                            lw=self._lora_view(blk, "proj_B"), res=xm)
            else:
                gemm(h2, blk.w_fc, pre, bias=blk.b_fc, gelu_out=act)
                gemm(act, blk.w_proj, st.x[i + 1][:rows], bias=blk.b_proj, res=xm)
        return st.x[st.layers][:rows]

    def _stack_backward(self, st: _Stack, rows: int, images: int, attr: Optional[Tensor], rows_per_sample: int,
                        need_input_grad: bool, grad_in_last: bool = False) -> Tensor:
        """st.g[:rows] holds dL/d(tower output) (a FairLoRA tower with grad_in_last: st.g_l[layers - 1] does - the buffer the
        last block's reductions read - and no copy is made); on return st.g holds dL/d(tower input) (if need_input_grad).
        LoRA gradients are written into params.grad."""
        lo = self.cfg.lora
        r, G, w = st.rank, lo.num_groups, st.width
        gemm = st.gemm
        g, g1 = st.g[:rows], st.g1[:rows]
        main = torch.cuda.current_stream(self.device)
        for i in range(st.layers - 1, -1, -1):
            blk = st.blocks[i]
            x, xm = st.x[i][:rows], st.xm[i][:rows]
            pre, act, h2 = st.pre[i][:rows], st.act[i][:rows], st.h2[i][:rows]
            last = (i == 0) and not need_input_grad
            if r:
                # gradient w.r.t. this block's output lives in its own buffer (read later by the side stream)
                gi, dpre = st.g_l[i][:rows], st.dpre_l[i][:rows]
                if i == st.layers - 1 and not grad_in_last:
                    self._glue(lambda gi=gi, g=g: gi.copy_(g))
                u, us2, us1 = st.u[:rows], st.us2[i][:rows], st.us1[i][:rows]
                pt = st.part[i]
                # ---- critical path: u = g B^T and dX (+ LoRA dx term), dS partials
                fused = self.fused_rank
                early = last and getattr(self, "tail_early", True)
                if early:
                    # the step's tail: dB(c_proj) needs only gi and the forward's ts2 - it starts beside this block's dX
                    # product instead of behind it
                    self._ev_record(self.ev_tail0, main)
                    self._ev_wait(self.grad_stream, self.ev_tail0)
                    with self._on(self.grad_stream):
                        ops.lora_grad_partial(gi, st.ts2[i][:rows], r, pt["proj_B"])
                # FFM_EPI_LGRAD: dB(c_fc) = dpre^T ts1 and dA(c_proj) = act^T us2 leave with the dX product of c_proj, which
                # holds dpre and (through pre) act in registers - per row tile, into the buffers the two reduction launches
                # they replace would have filled (77 MB per block that the side stream no longer reads beside the chain)
                lg = self._lgrad_rows(st, rows, blk) if fused else 0
                # ln_2's backward rides in this block's two dX products (not block 0 of a tower that stops there)
                lnb = self._fold_ln2_bwd(st, rows, blk) if (fused and lg and not last) else 0
                if fused:
                    ro = ops.RankOp(self.rk[i]["proj_B"], self._S(i, "proj"), attr, rows_per_sample,
                                    lo.scaling, lo.lambda_group, ts_out=us2, t_fwd=st.t2[i][:rows], ds_part=pt["proj_S"],
                                    lw_wide=self.lw_wide[i].get("proj_A"),
                                    lgrad=(st.ts1[i][:rows], pt["fc_B"], pt["proj_A"]) if lg else None)
                    gemm(gi, blk.w_proj_t, dpre, lw=self._lora_view(blk, "proj_A"), lw_is_kr=True,
                                dgelu_aux=pre, rankop=ro, b_packed=blk.pk("w_proj_t"),
                                lnb_stat=ops.LnBwdStat(st.lnb_part) if lnb else None)
                else:
                    ops.lora_down(gi, self._lora_view(blk, "proj_B"), True, self._S(i, "proj"), attr, r, G,
                                  rows_per_sample, lo.scaling, lo.lambda_group, u, us2, st.t2[i][:rows], pt["proj_S"])
                    gemm(gi, blk.w_proj_t, dpre, ts=us2, lw=self._lora_view(blk, "proj_A"), lw_is_kr=True,
                                dgelu_aux=pre)
                if fused and not last:
                    # u1 = dpre B_fc^T rides inside the dX GEMM of c_fc
                    ro = ops.RankOp(self.rk[i]["fc_B"], self._S(i, "fc"), attr, rows_per_sample,
                                    lo.scaling, lo.lambda_group, ts_out=us1, t_fwd=st.t1[i][:rows], ds_part=pt["fc_S"],
                                    lw_wide=self.lw_wide[i].get("fc_A"))
                    if lnb:
                        # ... and stores dL/d x_mid = LayerNorm backward(g_h) + gi directly: no ffm_layernorm_bwd launch
                        gemm(dpre, blk.w_fc_t, g1, lw=self._lora_view(blk, "fc_A"), lw_is_kr=True, rankop=ro,
                             b_packed=blk.pk("w_fc_t"),
                             lnb_apply=ops.LnBwdApply(st.lnb_part, lnb, xm, blk.ln2_w, st.st2[i][0], st.st2[i][1],
                                                      self.rk[i]["fc_A_lnrk"], gi))
                    else:
                        gemm(dpre, blk.w_fc_t, st.dh[:rows], lw=self._lora_view(blk, "fc_A"), lw_is_kr=True,
                             rankop=ro, b_packed=blk.pk("w_fc_t"))
                else:
                    if last:
                        # the dX chain ends with this down projection: the three reductions that do not need its result
                        # start beside it instead of behind it (the step's tail: 32 + 57 us in a row otherwise)
                        self._ev_record(self.ev_tail, main)
                        self._ev_wait(self.grad_stream, self.ev_tail)
                        with self._on(self.grad_stream):
                            if not early:
                                ops.lora_grad_partial(gi, st.ts2[i][:rows], r, pt["proj_B"])
                            if not lg:
                                ops.lora_grad_partial(act, us2, r, pt["proj_A"])
                                ops.lora_grad_partial(dpre, st.ts1[i][:rows], r, pt["fc_B"])
                    ops.lora_down(dpre, self._lora_view(blk, "fc_B"), True, self._S(i, "fc"), attr, r, G,
                                  rows_per_sample, lo.scaling, lo.lambda_group, u, us1, st.t1[i][:rows], pt["fc_S"])
                # ---- off the critical path: the four rank-r gradient reductions of this block
                def reductions(i=i, blk=blk, gi=gi, act=act, dpre=dpre, us2=us2, us1=us1, pt=pt, xm=xm, h2=h2, last=last, lg=lg):
                    # (the step's tail - block 0's last reduction and the sum of its partials - on the chain's own stream
                    # instead of behind two cross-stream events: measured, 4.699 -> 4.692 ms per step, nothing; not kept)
                    self._ev_record(self.ev_layer[i], main)
                    self._ev_wait(self.grad_stream, self.ev_layer[i])
                    with self._on(self.grad_stream):
                        if not last:
                            ops.lora_grad_partial(gi, st.ts2[i][:rows], r, pt["proj_B"])
                            if not lg:
                                ops.lora_grad_partial(act, us2, r, pt["proj_A"])
                                ops.lora_grad_partial(dpre, st.ts1[i][:rows], r, pt["fc_B"])
                        if self._fold_ln2(st, rows):
                            ops.lora_grad_partial_ln(xm, us1, st.st2[i][0], st.st2[i][1], blk.ln2_w, blk.ln2_b, r, pt["fc_A"])
                        else:
                            ops.lora_grad_partial(h2, us1, r, pt["fc_A"])
                        # this block's six gradient tensors are complete: sum their partials now, behind the four reductions
                        # on the same side stream (one launch per block; only block 0's is left when the dX chain ends -
                        # ONE launch for all 72 tensors at the end sat in the step's tail for ~50 us)
                        self._reduce_plan(st, rows, need_input_grad, i).run()
                # red_at: where the side stream may start them (everything they read is kept per layer).  0: behind this
                # block's dX(c_fc) - beside its LayerNorm / out-proj / attention backward; 1: behind its attention backward;
                # 2: behind the whole block - beside the NEXT block's two FairLoRA products, whose main loops live on
                # LDS / L2 operands (the last block's have nothing behind them and always start at once)
                # Measured (one call, alternating): configs[1] (6304 rows) 4.65-4.71 / 4.66-4.67 / 4.75-4.76 ms per step for
                # 0 / 1 / 2; configs[3] (19 700 rows, rank 16: 52 MB of partials per block, whose sum took the attention
                # backward beside it from 49 to 74 us) 12.45 / 12.42-12.46 / 12.37-12.38.
                auto = 2 if rows > 12608 else 0
                red_at = 0 if last or i == 0 else (auto if self.red_at is None else self.red_at)
                if red_at == 0:
                    reductions()
                if last:
                    break
                if not fused:
                    gemm(dpre, blk.w_fc_t, st.dh[:rows], ts=us1, lw=self._lora_view(blk, "fc_A"),
                                lw_is_kr=True)
                gout = st.g_l[i - 1][:rows] if i > 0 else g
            else:
                gi, dpre, gout = g, st.dpre[:rows], g
                lnb = 0
                gemm(gi, blk.w_proj_t, dpre, dgelu_aux=pre)
                gemm(dpre, blk.w_fc_t, st.dh[:rows])
            if not lnb:
                ops.layernorm_bwd(st.dh[:rows], xm, blk.ln2_w, st.st2[i][0], st.st2[i][1], gi, g1)
            gemm(g1, blk.w_out_t, st.do[:rows], b_packed=blk.pk("w_out_t"))
            # ln_1's backward: its two row sums leave with the attention backward (two partial rows per head), and the dX
            # product of the in-projection stores dL/d x = LayerNorm backward(g_h) + g1 directly
            lnb1 = self._fold_ln1_bwd(st, rows) if r else 0
            ops.attention_bwd(st.qkv[i][:rows], st.o[i][:rows], st.do[:rows], st.lse[i], st.delta, st.dqkv[:rows],
                              images, st.L, st.heads, st.causal,
                              ln_stat=(blk.c_in, blk.d_in, st.lnb_part) if lnb1 else None)
            if r and red_at == 1:
                reductions()
            if lnb1:
                gemm(st.dqkv[:rows], blk.w_in_t, gout, b_packed=blk.pk("w_in_t"),
                     lnb_apply=ops.LnBwdApply(st.lnb_part, lnb1, x, blk.ln1_w, st.st1[i][0], st.st1[i][1], None, g1))
            else:
                gemm(st.dqkv[:rows], blk.w_in_t, st.dh[:rows], b_packed=blk.pk("w_in_t"))
                ops.layernorm_bwd(st.dh[:rows], x, blk.ln1_w, st.st1[i][0], st.st1[i][1], g1, gout)
            if r and red_at == 2:
                reductions()
        if r:
            if self.sops.glob:
                self._glue(self.sops.finish, self.grad_stream)     # dS_eff -> dS, dS_global
            self._ev_record(self.ev_grads, self.grad_stream)
            self._ev_wait(main, self.ev_grads)
        return g

    def _lgrad_rows(self, st: _Stack, rows: int, blk: _Block) -> int:
        """Row tiles of the FFM_EPI_LGRAD partial products for this row count, 0 when the kernel that serves the dX product
        of c_proj has no such epilogue (then the two reduction launches run) or FFM_LGRAD=0 asks for the launches."""
        key = (rows, blk.packed is not None)
        if key not in st.lgrad:
            n = 0
            deriv = os.environ.get("FFM_GELU_DERIV", "0") == "1"   # (the saved tensor is then gelu'(pre): no activation to recompute)
            if self.use_lgrad and not deriv and blk.packed is not None and _is16(self.dtype) and st.rank % 4 == 0:
                n = max(0, ops.gemm_lgrad_rows(rows, 4 * st.width, st.width, st.rank, self.dtype, True))
                if n > ops.lora_grad_splits(rows):            # (the partial buffers are sized for the reduction kernel's splits)
                    n = 0
            st.lgrad[key] = n
        return st.lgrad[key]

    def _reduce_plan(self, st: _Stack, rows: int, full_bwd: bool, layer: int):
        """Descriptor table (built once per row count and block) that sums the block's partials into params.grad."""
        key = (rows, full_bwd, layer)
        if key not in st.plans:
            r, G, w = st.rank, self.cfg.lora.num_groups, st.width
            nsp = ops.lora_grad_splits(rows)
            ent = []
            for li, (blk, pt) in enumerate(zip(st.blocks, st.part)):
                if li != layer:
                    continue
                gv = lambda role: self.params.view(blk.lora[role], "grad")
                # dS partial rows: GEMM row tiles when the down projection is fused, lora_down blocks otherwise
                # (block 0's c_fc has no dX GEMM, so it always uses the stand-alone kernel)
                # dS partial rows written by the dX GEMMs (c_proj: N = 4w, K = w; c_fc: N = w, K = 4w)
                pk = st.blocks[li].packed is not None
                nb_p = _ds_rows(rows, 4 * w, w, r, self.dtype, pk, dgelu=True) if self.fused_rank \
                    else ops.lora_down_blocks(rows, w, r, self.dtype)
                nb_f = _ds_rows(rows, w, 4 * w, r, self.dtype, pk) if (self.fused_rank and (li > 0 or full_bwd)) \
                    else ops.lora_down_blocks(rows, 4 * w, r, self.dtype)
                # (FFM_EPI_LGRAD: proj_A and fc_B hold one partial per row tile of the dX product of c_proj)
                nlg = (self._lgrad_rows(st, rows, st.blocks[li]) if self.fused_rank else 0) or nsp
                ent += [(pt["proj_S"], nb_p, G * r, self._dS(li, "proj"), 0, 0), (pt["fc_S"], nb_f, G * r, self._dS(li, "fc"), 0, 0),
                        (pt["proj_B"], nsp, w * r, gv("proj_B"), w, r), (pt["proj_A"], nlg, 4 * w * r, gv("proj_A"), 0, 0),
                        (pt["fc_B"], nlg, 4 * w * r, gv("fc_B"), 4 * w, r), (pt["fc_A"], nsp, w * r, gv("fc_A"), 0, 0)]
            st.plans[key] = ops.ReducePlan(ent, self.device)
        return st.plans[key]

    # ------------------------------------------------------------ replay --
    # A training step is a fixed sequence of C launches over static buffers plus a few host-side
    # actions (event record/wait, the PyTorch glue at the ends of the text tower).  The helpers below
    # execute an action and, while a recording is active, also append it to the replay plan.
    def _ev_record(self, ev, stream) -> None:
        ev.record(stream)
        ops.record_callable(lambda: ev.record(stream))

    def _ev_wait(self, stream, ev) -> None:
        stream.wait_event(ev)
        ops.record_callable(lambda: stream.wait_event(ev))

    def _glue(self, fn, stream=None) -> None:
        def run():
            if stream is None:
                fn()
            else:
                with torch.cuda.stream(stream):
                    fn()
        run()
        ops.record_callable(run)

    class _on:
        """Make `stream` current for the C launches issued inside (their stream pointer is baked into the
        recorded launch, so nothing is recorded for the context itself)."""

        def __init__(self, stream):
            self.ctx = torch.cuda.stream(stream)

        def __enter__(self):
            return self.ctx.__enter__()

        def __exit__(self, *a):
            return self.ctx.__exit__(*a)

    # --------------------------------------------------------------- text --
    def _text_forward(self, with_grad: bool = True, stream=None) -> None:
        """prompts = [prefix, ctx, suffix] + pos -> text tower -> EOT gather, ln_final, projection, normalise (, mean over
        the prompts) -> tbar_buf [n_cls, D], or with a transport head every prompt's normalised feature tn_buf [N*n_cls, D]
        (trainers/GLP_OT_SVLoRA.py:131-152, 55-66, 709-718): HIP kernels end to end (csrc/text.hip)."""
        cfg = self.cfg
        rows = self.n_text * self.txt_len
        ops.text_embed(self.tok_prefix, self.params.view("prompt_learner.ctx"), self.tok_suffix, self.txt_pos, self.txt.x[0],
                       cfg.n_cls, self.txt_len)
        self._stack_forward(self.txt, rows, self.n_text, None, self.txt_len)
        ops.text_tail_fwd(self.txt.x[self.txt.layers], self.eot_rows, self.lnfinal[0], self.lnfinal[1], self.text_proj, self.tf_buf,
                          self.tn_all, self.t_rnorm, self.t_stats, None if self.ot else self.tbar_buf, cfg.n_prompts, cfg.n_cls)

    def _text_backward(self, stream=None) -> None:
        """dtbar (or dtn) -> gradient of the tower's EOT rows -> tower backward -> d ctx."""
        cfg = self.cfg
        rows = self.n_text * self.txt_len
        ops.text_tail_bwd(self.txt.x[self.txt.layers], self.eot_rows, self.lnfinal[0], self.text_proj, self.tn_all, self.t_rnorm,
                          self.t_stats, None if self.ot else self.dtbar, self.dtn if self.ot else None, self.t_dy, self.txt.g,
                          cfg.n_prompts, cfg.n_cls, self.txt_len)
        self._stack_backward(self.txt, rows, self.n_text, None, self.txt_len, True)
        ops.text_ctx_grad(self.txt.g, self.params.view("prompt_learner.ctx", "grad"), cfg.n_cls, self.txt_len)

    # ------------------------------------------------------------- vision --
    def _as_f32(self, image: Tensor) -> Tensor:
        """uint8 transport (fairfedmed_amd.data, transport="uint8"): expand on the GPU to the float32 batch the
        reference's loader ships - a single SLO / X-ray channel is repeated to 3 (utils/data_utils.py:676-679)."""
        if image.dtype != torch.uint8:
            return image
        if not image.is_cuda or image.dim() != 4:
            raise TypeError("uint8 images must be CUDA tensors [B, C, H, W]")
        b, c1 = image.shape[:2]
        rep = 1 if self.cfg.dim_per_3d_slice else (3 // c1 if c1 in (1, 3) else 1)
        stage = getattr(self, "_u8_stage", None)
        shape = (b, c1 * rep) + tuple(image.shape[2:])
        if stage is None or stage.shape[0] < b or tuple(stage.shape[1:]) != shape[1:]:
            stage = self._u8_stage = torch.empty(shape, device=self.device, dtype=torch.float32)
        return ops.expand_u8(image.contiguous(), stage[:b], rep)

    def _check_batch(self, image: Tensor) -> Tuple[int, int]:
        cfg, v = self.cfg, self.cfg.vision
        if not image.is_cuda or image.dtype != torch.float32:
            raise TypeError("image must be a float32 CUDA tensor of raw 0..255 values")
        b, c, h, w = image.shape
        D = cfg.dim_per_3d_slice
        if h != v.image_size or w != v.image_size or (not D and c != 3) or (D and c % D):
            raise ValueError(f"expected [B,{'S*%d' % D if D else 3},{v.image_size},{v.image_size}], "
                             f"got {tuple(image.shape)}")
        S = c // D if D else 1                       # slices per sample: the ViT batch is b*S (:683-684)
        if b * S > self.max_images:
            raise ValueError(f"{b * S} ViT images exceed the engine's max_images={self.max_images}")
        return b, S

    def _load_inputs(self, image: Tensor, attr: Optional[Tensor], label: Optional[Tensor]):
        """Per-step inputs -> static buffers (these three launches are the only ones not replayed)."""
        cfg, v = self.cfg, self.cfg.vision
        image = self._as_f32(image)
        b, S = self._check_batch(image)
        images = b * S
        P = v.grid * v.grid
        if self.is3d:
            # trainable 5x5 slice conv + per-image min-max (trainers/GLP_OT_SVLoRA.py:681-690), then the patches
            self._image = image.contiguous()
            ops.slice_conv_fwd(self._image, self.params.view("proj_per_3d_slice.weight"),
                               self.params.view("proj_per_3d_slice.bias"), self.conv_out[:images], self.mm_part,
                               self.mnmx, self.mm_cnt, cfg.dim_per_3d_slice)
            ops.patchify_minmax(self.conv_out[:images], self.mnmx, self.mm_cnt, self.cols[:images * P], v.patch,
                                cfg.pixel_mean, cfg.pixel_std)
        else:
            ops.patchify(image.contiguous(), self.cols[:images * P], v.patch, cfg.pixel_mean, cfg.pixel_std)
        if attr is not None:
            self.attr_i32[:b].copy_(attr)
        if label is not None:
            self.label_buf[:b].copy_(label)
        return b, S

    def _vision_forward(self, b: int, S: int, has_attr: bool, wait=None) -> None:
        cfg, v = self.cfg, self.cfg.vision
        images = b * S
        P, L = v.grid * v.grid, v.tokens
        rows = images * L
        a32 = self.attr_i32[:b] if has_attr else None
        if self.sops.glob:
            self._glue(self.sops.prepare)             # S_eff = S + S_global
        ops.gemm_nt(self.cols[:images * P], self.conv_w, self.patch_out[:images * P], b_packed=self.pk_out.get("conv_w"))
        ops.embed_lnpre(self.patch_out[:images * P], self.cls, self.pos, self.lnpre[0], self.lnpre[1],
                        self.vis.x[0][:rows], images, L, rowstat=self.vis.rowp[0] if self.vis.rowp is not None else None)
        self._rank_operands_ready()                   # LoRA matrices -> GEMM rank operands (they change every step)
        out = self._stack_forward(self.vis, rows, images, a32, L * S)
        ops.layernorm_fwd(out, self.hpost[:rows], self.lnpost[0], self.lnpost[1], self.post_stats[0],
                          self.post_stats[1])
        ops.gemm_nt(self.hpost[:rows], self.proj_t, self.feat[:rows], b_packed=self.pk_out.get("proj_t"))
        if wait is not None:
            self._ev_wait(torch.cuda.current_stream(self.device), wait)     # text features ready
        self._head_forward(rows, images, L)

    def _pack_rank_operands(self) -> None:
        """ffm_lora_pack_multi over every adapter (one or two launches) on the CURRENT stream."""
        plan = getattr(self, "pack_plan", None)
        if plan is not None and getattr(self, "fused_rank", True):
            plan.run()

    def _rank_operands_ready(self) -> None:
        """In a training step the packing runs at the head of the side stream, beside the patch embedding (27 us off the
        main queue), and the vision chain waits for it here; outside a step (inference) it runs in line."""
        ev = getattr(self, "_pack_event", None)
        if ev is None:
            self._pack_rank_operands()
        else:
            self._ev_wait(torch.cuda.current_stream(self.device), ev)

    def _head_forward(self, rows: int, images: int, L: int) -> None:
        cfg = self.cfg
        if self.ot:
            ops.ot_head_fwd(self.feat[:rows], self.tn_buf, self.logit_scale, self.rnorm, self.ot_sim, self.ot_T, self.ot_errs,
                            self.ot_istop, self.ot_tsum, self.logits_img, images, L, cfg.n_cls, cfg.n_prompts, self.ot,
                            cfg.ot_eps, cfg.ot_thresh, cfg.ot_max_iter, cfg.ot_top_percent)
        else:
            ops.head_fwd(self.feat[:rows], self.tbar_buf, self.logit_scale, self.fbar, self.rnorm, self.logits_img,
                         images, L, cfg.n_cls)

    def _head_backward(self, rows: int, images: int, L: int) -> None:
        cfg = self.cfg
        if self.ot:
            P = cfg.n_prompts * cfg.n_cls
            ops.ot_head_bwd(self.feat[:rows], self.tn_buf, self.logit_scale, self.rnorm, self.ot_T, self.dlogits_img,
                            self.dfeat[:rows], self.ot_dtn_part, images, L, cfg.n_cls, cfg.n_prompts)
            ops.reduce_partials(self.ot_dtn_part, images, P * self.cfg.vision.out_dim, self.dtn.view(-1))
        else:
            ops.head_bwd(self.feat[:rows], self.tbar_buf, self.logit_scale, self.fbar, self.rnorm, self.dlogits_img,
                         self.dfeat[:rows], self.dtbar, images, L, cfg.n_cls)

    def _vision_backward(self, b: int, S: int, has_attr: bool) -> None:
        """dfeat -> gradients of every trainable tensor of the image side (params.grad)."""
        cfg, v = self.cfg, self.cfg.vision
        images, L = b * S, v.tokens
        rows = images * L
        a32 = self.attr_i32[:b] if has_attr else None
        ops.gemm_nt(self.dfeat[:rows], self.proj, self.vis.dh[:rows], b_packed=self.pk_out.get("proj"))
        # (straight into the buffer the last block's LoRA-gradient reductions read: no copy on the main stream)
        gl = self.vis.rank > 0
        ops.layernorm_bwd(self.vis.dh[:rows], self.vis.x[v.layers][:rows], self.lnpost[0], self.post_stats[0],
                          self.post_stats[1], None, (self.vis.g_l[v.layers - 1] if gl else self.vis.g)[:rows])
        self._stack_backward(self.vis, rows, images, a32, L * S, self.is3d, grad_in_last=gl)
        if self.is3d:
            # dL/d(tokens) -> ln_pre / pos-add backward -> dX of the patch embedding (columns of the patches)
            P = v.grid * v.grid
            ops.embed_lnpre_bwd(self.vis.g[:rows], self.patch_out[:images * P], self.pos, self.lnpre[0],
                                self.dpatch[:images * P], images, L)
            ops.gemm_nt(self.dpatch[:images * P], self.conv_w_t, self.dcols[:images * P])

    # ---------------------------------------------------------------- API --
    @torch.no_grad()
    def forward(self, image: Tensor, attr: Optional[Tensor] = None) -> Tensor:
        """CustomCLIP.forward(image, attr) -> logits [B, n_cls] (inference)."""
        if self.sops.type != "FairLoRA":
            attr = None                               # LoRALinear / SVLoRALinear.forward ignore attr (:241, :307)
        b, S = self._load_inputs(image, attr, None)
        self._text_forward(False)
        self._vision_forward(b, S, attr is not None)
        return self.logits_img[:b * S].view(b, S, -1).mean(1)

    def _step_body(self, b: int, S: int, has_attr: bool) -> None:
        cfg, v = self.cfg, self.cfg.vision
        main = torch.cuda.current_stream(self.device)
        images, L = b * S, v.tokens
        rows = images * L
        a32 = self.attr_i32[:b] if has_attr else None
        self._ev_record(self.ev_start, main)
        self._ev_wait(self.side, self.ev_start)           # parameters of the previous step are final
        with self._on(self.side):
            self._pack_rank_operands()
            self._ev_record(self.ev_pack, self.side)
            self._text_forward(True, self.side)
        self._ev_record(self.ev_text_fwd, self.side)
        self._pack_event = self.ev_pack
        try:
            self._vision_forward(b, S, has_attr, wait=self.ev_text_fwd)
        finally:
            self._pack_event = None
        ops.ce_loss(self.logits_img, self.label_buf, self.logits, self.prob, self.loss, self.dlogits_img,
                    self.finite, b, S, cfg.n_cls)
        if self.scale_state is not None:
            ops.loss_scale(self.dlogits_img[:b * S], self.scale_state)
        self._head_backward(rows, images, L)
        self._ev_record(self.ev_head_bwd, main)
        self._ev_wait(self.side, self.ev_head_bwd)
        with self._on(self.side):
            if self._counts_buf is not None:
                # the batch's evaluator counts for the trainer's per-step summary (enable_step_counts): prob is final, and
                # beside the backward pass the 20 us launch costs the chain nothing (behind the SGD step it sat between steps)
                ops.eval_counts(self.prob[:b], self.label_buf[:b], None, 0, method="pairs", out=self._counts_buf)
            self._text_backward(self.side)
        self._ev_record(self.ev_text_bwd, self.side)
        self._vision_backward(b, S, has_attr)
        self._ev_wait(main, self.ev_text_bwd)

    def forward_backward(self, image: Tensor, attr: Optional[Tensor], label: Tensor) -> Dict[str, Tensor]:
        """Forward, CE loss, backward; gradients of every trainable tensor land in params.grad.
        Returns device tensors (no host sync): loss [1], logits [B,n_cls], prob [B,n_cls], finite [1].
        The first call for a batch shape records the step's launch plan; later calls replay it."""
        if self.sops.type != "FairLoRA":
            attr = None
        with torch.no_grad():
            b, S = self._load_inputs(image, attr, label)
            key = (b, S, attr is not None, torch.cuda.current_stream(self.device).cuda_stream)
            plan = self.step_plans.get(key) if self.use_replay else None
            if plan is not None:
                for f in plan:
                    f()
            elif self.use_replay:
                plan = []
                with ops.record(plan):
                    self._step_body(b, S, attr is not None)
                self.step_plans[key] = plan
            else:
                self._step_body(b, S, attr is not None)
            if self.is3d:
                images, P, D = b * S, self.cfg.vision.grid ** 2, self.cfg.dim_per_3d_slice
                ops.slice_bwd(self.dcols[:images * P], self._image, self.conv_out[:images], self.mnmx, self.mm_cnt,
                              self.dconv[:images], self.ab_part, self.gmm, self.wpart, D, self.cfg.vision.patch,
                              self.cfg.pixel_std)
                nw = 3 * D * 25 + 3
                off = self.params.offsets["proj_per_3d_slice.weight"][0]
                assert self.params.offsets["proj_per_3d_slice.bias"][0] == off + nw - 3
                nblk = ops.slice_wgrad_blocks(self.cfg.vision.image_size, self.cfg.vision.image_size)
                ops.reduce_partials(self.wpart, images * nblk, nw, self.params.grad[off:off + nw])
            if self.scale_state is not None:
                ops.unscale_check(self.params.grad, self.scale_state)
        out = {"loss": self.loss, "logits": self.logits[:b], "prob": self.prob[:b], "finite": self.finite}
        if self._counts_buf is not None:
            out["counts"] = self._counts_buf          # ffm_eval_counts of (prob, label): rows {unknown, all}
        return out

    @torch.no_grad()
    def sgd_step(self, lr: float, momentum: float, weight_decay: float, repeats: int = 1) -> None:
        """optim.step(), `repeats` times on the gradients of the last forward_backward (one launch)."""
        p = self.params
        if self.scale_state is not None:                # fp16: skipped when the gradients overflowed; the scale then halves
            ops.sgd_momentum_gated(p.flat, p.grad, p.momentum, lr, momentum, weight_decay, p.steps == 0, repeats, self.scale_state)
        else:
            ops.sgd_momentum(p.flat, p.grad, p.momentum, lr, momentum, weight_decay, p.steps == 0, repeats)
        p.steps += repeats

    def enable_step_counts(self, on: bool = True) -> None:
        """Binary tasks: every training step also leaves ffm_eval_counts(prob, label) in out["counts"] (int64 [2, 10]: the
        integers behind the reference's per-step accuracy / AUC summary, trainers/GLP_OT_SVLoRA.py:959-970), computed on the
        side stream beside the backward pass."""
        want = on and self.cfg.n_cls == 2
        if want != (self._counts_buf is not None):
            self._counts_buf = torch.zeros(2, ops.EVAL_SLOTS, device=self.device, dtype=torch.int64) if want else None
            self.step_plans.clear()

    def set_overlap(self, on: bool) -> None:
        """on: text tower and LoRA-gradient reductions run on their own streams beside the vision chain
        (default).  off: everything on the caller's stream, one kernel at a time (clean per-kernel timings)."""
        cur = torch.cuda.current_stream(self.device)
        if on:
            self.side, self.grad_stream = self._side0, self._grad0
        else:
            self.side = self.grad_stream = cur
        self.step_plans.clear()

    # ------------------------------------------------------------- graph --
    def capture_train_step(self, batch_size: int, lr: float, momentum: float, weight_decay: float) -> "GraphedStep":
        """Capture forward + backward + SGD (all three streams) into one hipGraph.  A step then costs the
        host one graph launch instead of ~450 Python->C calls (the eager loop is host-bound at ~7 ms)."""
        return GraphedStep(self, batch_size, lr, momentum, weight_decay)

    def trainable_state(self) -> Dict[str, Tensor]:
        return {k: self.params.view(k) for k in self.params.keys}


class GraphedStep:
    """One captured training step.  ``run(image, attr, label)`` copies the batch into the static input
    buffers and replays the graph; results are the engine's usual device tensors (loss, logits, prob, finite)."""

    def __init__(self, eng: FairLoRAEngine, batch_size: int, lr: float, momentum: float, weight_decay: float):
        self.eng = eng
        eng.use_replay = False                        # the hipGraph replaces the recorded launch plan
        v, dev = eng.cfg.vision, eng.device
        self.image = torch.zeros(batch_size, 3, v.image_size, v.image_size, device=dev)
        self.attr = torch.zeros(batch_size, device=dev, dtype=torch.int64)
        self.label = torch.zeros(batch_size, device=dev, dtype=torch.int64)
        self.hp = torch.tensor([lr, momentum, weight_decay], device=dev, dtype=torch.float32)
        p = eng.params
        if p.steps == 0:
            p.momentum.zero_()
        keep = (p.flat.clone(), p.momentum.clone(), p.grad.clone())
        # warm-up off the default stream (sets kernel attributes, sizes the allocator), then restore the state
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        p.flat.copy_(keep[0]); p.momentum.copy_(keep[1]); p.grad.copy_(keep[2])
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._body()
        p.flat.copy_(keep[0]); p.momentum.copy_(keep[1]); p.grad.copy_(keep[2])
        torch.cuda.synchronize(dev)

    def _body(self):
        out = self.eng.forward_backward(self.image, self.attr, self.label)
        with torch.no_grad():
            p = self.eng.params
            ops.sgd_momentum_dev(p.flat, p.grad, p.momentum, self.hp)
        return out

    def set_lr(self, lr: float) -> None:
        self.hp[0:1].fill_(lr)

    def run(self, image: Optional[Tensor] = None, attr: Optional[Tensor] = None, label: Optional[Tensor] = None):
        if image is not None:
            self.image.copy_(image, non_blocking=True)
        if attr is not None:
            self.attr.copy_(attr, non_blocking=True)
        if label is not None:
            self.label.copy_(label, non_blocking=True)
        self.graph.replay()
        self.eng.params.steps += 1
        return self.out
