"""RN50 image tower for the FairLoRA engine (BASELINE.json configs[4]).

``ModifiedResNet_GLP_OT`` (clip/model.py:227-301) after ``apply_lora_to_model``
(trainers/GLP_OT_SVLoRA.py:541-573): FairLoRALinear on every Bottleneck's 1x1
conv1 / conv3, LoRALinear on the four attention-pool projections, BatchNorm
weights and biases trainable (:822-829), everything else frozen.

Data layout: activations are NHWC rows, row = (image, y, x), channels contiguous,
so every 1x1 convolution IS ffm_gemm_nt on the rows (with the FairLoRA epilogue),
3x3 convolutions are an implicit GEMM (ffm_conv3x3_nhwc: the GEMM's A loader reads the patch from the
neighbouring pixels, weight re-ordered once to [Cout, (ky, kx, cin)]; the input gradient is the same call on dY), and the attention pool runs on [image*L, E] token rows with
the ViT attention kernels (head_dim 64, 32 heads).  The text tower, logits head,
flat parameter buffer, SGD, streams and launch-plan replay are the base class's.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from ._lib import is16 as _is16
from .engine import FairLoRAEngine, _round_up
from .synth import buffer_keys

Tensor = torch.Tensor


class _Lora:
    """One (Fair)LoRA-wrapped linear over token / pixel rows: y = x W^T (+b) + scaling ((x A) * s_b) B.
    fair=False is the reference's LoRALinear (:218-241): the same update with s = 1 and no groups."""

    def __init__(self, eng: "RN50Engine", prefix: str, K: int, N: int, max_rows: int, fair: bool):
        self.eng, self.K, self.N, self.fair = eng, K, N, fair
        self.kA, self.kB = prefix + "lora_A.weight", prefix + "lora_B.weight"
        self.kS = prefix + "lora_S.weight" if fair else None
        self.prefix = prefix
        lo, dt = eng.cfg.lora, eng.dtype
        r = lo.rank
        f = lambda *s: torch.zeros(*s, device=eng.device, dtype=torch.float32)
        self.t, self.ts, self.u, self.us = f(max_rows, r), f(max_rows, r), f(max_rows, r), f(max_rows, r)
        ns = ops.lora_grad_splits(max_rows)
        self.pA, self.pB = f(ns * K * r), f(ns * N * r)
        # rank <= 16: the down projections ride inside the GEMMs (FFM_EPI_RANKOP); their operands are the LoRA
        # matrices re-packed to [16, K] in the compute dtype once per step (one launch for all sites)
        self.fused = 0 < r <= 16
        nb = ops.lora_down_blocks_max(max_rows, N, r, dt)
        if self.fused:
            nb = max(nb, self._tiles(max_rows))
            self.rkA = torch.zeros(16, K, device=eng.device, dtype=dt)
            self.rkB = torch.zeros(16, N, device=eng.device, dtype=dt)
            eng.pack_entries += [(eng.params.view(self.kA), False, self.rkA), (eng.params.view(self.kB), True, self.rkB)]
        self.pS = f(nb * lo.num_groups * r) if fair else None

    def _tiles(self, rows: int) -> int:
        """dS partial rows the dX GEMM (M = rows, N = in features, K = out features) writes."""
        from . import _lib as L
        return ops.gemm_tiles_m(rows, self.K, self.N, L.EPI_LORA | L.EPI_LORA_KR | L.EPI_RANKOP, self.eng.cfg.lora.rank,
                                self.eng.dtype, False)

    def _s(self) -> Tuple[Tensor, int]:
        e = self.eng
        return (e.sops.op(self.prefix), e.cfg.lora.num_groups) if self.fair else (e.ones_s, 1)

    def fwd(self, x: Tensor, W: Tensor, out: Tensor, attr: Optional[Tensor], rps: int, bias=None, res=None,
            colstats: Optional[Tensor] = None) -> None:
        """colstats: the GEMM also leaves the column sums of its row tiles there (the BatchNorm that follows)."""
        e, lo = self.eng, self.eng.cfg.lora
        rows = x.shape[0]
        S, G = self._s()
        attr = attr if self.fair else None
        if self.fused:
            ro = ops.RankOp(self.rkA, S, attr, rps, lo.scaling, lo.lambda_group, t_out=self.t[:rows], ts_out=self.ts[:rows])
            ops.gemm_nt(x, W, out, bias=bias, lw=e.params.view(self.kB), res=res, rankop=ro, colstats=colstats)
            return
        ops.lora_down(x, e.params.view(self.kA), False, S, attr, lo.rank, G, rps, lo.scaling,
                      lo.lambda_group, self.t[:rows], self.ts[:rows])
        ops.gemm_nt(x, W, out, bias=bias, ts=self.ts[:rows], lw=e.params.view(self.kB), res=res, colstats=colstats)

    def bwd(self, g: Tensor, Wt: Tensor, dx: Tensor, x: Tensor, attr: Optional[Tensor], rps: int, res=None,
            defer: bool = False, bnbwd=None, colstats: Optional[Tensor] = None) -> None:
        """g = dL/dy; writes dx = g W (+ LoRA term) (+ res) and the partial sums of dA, dB, dS (defer: the caller runs
        grads() itself, on the gradient stream).  bnbwd / colstats (fused sites only): dx is the gradient of a BatchNorm
        output, and the product's epilogue leaves that BatchNorm's backward column sums in colstats (ops.gemm_nt)."""
        e, lo = self.eng, self.eng.cfg.lora
        rows = g.shape[0]
        S, G = self._s()
        attr = attr if self.fair else None
        if self.fused:
            ro = ops.RankOp(self.rkB, S, attr, rps, lo.scaling, lo.lambda_group, ts_out=self.us[:rows],
                            t_fwd=self.t[:rows] if self.fair else None, ds_part=self.pS)
            ops.gemm_nt(g, Wt, dx, lw=e.params.view(self.kA), lw_is_kr=True, res=res, rankop=ro, bnbwd=bnbwd,
                        colstats=colstats)
        else:
            assert bnbwd is None
            ops.lora_down(g, e.params.view(self.kB), True, S, attr, lo.rank, G, rps, lo.scaling,
                          lo.lambda_group, self.u[:rows], self.us[:rows], self.t[:rows] if self.fair else None, self.pS)
            ops.gemm_nt(g, Wt, dx, ts=self.us[:rows], lw=e.params.view(self.kA), lw_is_kr=True, res=res)
        if not defer:
            self.grads(g, x)

    def grads(self, g: Tensor, x: Tensor) -> None:
        """The rank-r reductions dB = g^T ts, dA = x^T us (partials); nothing downstream of them in the dX chain."""
        rows, r = g.shape[0], self.eng.cfg.lora.rank
        ops.lora_grad_partial(g, self.ts[:rows], r, self.pB)
        ops.lora_grad_partial(x, self.us[:rows], r, self.pA)

    def reduce_entries(self, rows: int) -> list:
        e, lo = self.eng, self.eng.cfg.lora
        r, nsp = lo.rank, ops.lora_grad_splits(rows)
        gv = lambda k: e.params.view(k, "grad")
        ent = [(self.pB, nsp, self.N * r, gv(self.kB), self.N, r), (self.pA, nsp, self.K * r, gv(self.kA), 0, 0)]
        if self.fair:
            nb = self._tiles(rows) if self.fused else ops.lora_down_blocks(rows, self.N, r, e.dtype)
            ent.append((self.pS, nb, lo.num_groups * r, e.sops.grad(self.prefix), 0, 0))
        return ent


class _BN:
    """nn.BatchNorm2d over NHWC rows; weight / bias live in the flat trainable buffer."""

    def __init__(self, eng: "RN50Engine", prefix: str, C: int, max_rows: int, aux: bool = False):
        self.eng, self.prefix, self.C, self.aux = eng, prefix, C, aux    # aux: the downsample branch's own scratch
        f = lambda: torch.zeros(C, device=eng.device, dtype=torch.float32)
        self.run_mean, self.run_var, self.mean, self.rstd = f(), f(), f(), f()
        eng.bn_scratch = max(eng.bn_scratch, ops.bn_blocks(max_rows) * 2 * C)
        eng.bn_cmax = max(eng.bn_cmax, C)
        t = (max_rows + 127) // 128                                # row tiles of the GEMM that produces this layer's input
        if t <= 4096:
            eng.stat_scratch = max(eng.stat_scratch, 2 * t * C)
        # a convolution split over K leaves one partial row per 8 rows below 4096 rows and per 32 rows from there on
        # (csrc/gemm.hip: splitk_cs_rows) - NOT monotone in the row count, so a ragged last batch can need more partial
        # rows than the largest one: size for the worst M <= max_rows (at most 4096 partial rows either way)
        worst = max((min(max_rows, 4095) + 7) // 8, (max_rows + 31) // 32 if max_rows >= 4096 else 0)
        eng.stat_scratch = max(eng.stat_scratch, 2 * min(worst, 4096) * C)
        eng.bns.append(self)

    def fwd(self, x: Tensor, y: Tensor, relu: bool, res: Optional[Tensor] = None, part: Optional[Tensor] = None,
            part_rows: int = 0) -> None:
        """part / part_rows: the column sums the producing GEMM left behind (RN50Engine.stat_buf); else own pass over x."""
        e = self.eng
        ops.bn_fwd(x, e.params.view(self.prefix + "weight"), e.params.view(self.prefix + "bias"), self.run_mean,
                   self.run_var, self.mean, self.rstd, part if part_rows else (e.bn_part_d if self.aux else e.bn_part), y,
                   e.bn_training, relu, res, part_rows=part_rows)

    def bwd(self, dy: Tensor, relu_out: Optional[Tensor], x: Tensor, dx: Tensor, g_out: Optional[Tensor] = None,
            part: Optional[Tensor] = None, part_rows: int = 0) -> None:
        """part / part_rows: the backward column sums the GEMM that produced dy left behind (FFM_EPI_BNBWD); else own pass."""
        e = self.eng
        ops.bn_bwd(dy, relu_out, x, e.params.view(self.prefix + "weight"), self.mean, self.rstd,
                   part if part_rows else (e.bn_part_d if self.aux else e.bn_part), e.bn_k12_d if self.aux else e.bn_k12,
                   e.params.view(self.prefix + "weight", "grad"), e.params.view(self.prefix + "bias", "grad"), dx, g_out,
                   part_rows=part_rows)


class _Bneck:
    """clip/model.py:11-60 with every convolution at stride 1 and an average pool after conv2 when stride > 1."""

    def __init__(self, eng: "RN50Engine", p: str, inpl: int, planes: int, stride: int, H: int, max_images: int):
        self.eng, self.p, self.inpl, self.planes, self.stride = eng, p, inpl, planes, stride
        self.Hin, self.Hout = H, H // stride
        Ri, Ro = max_images * H * H, max_images * self.Hout * self.Hout
        e = lambda rows, C: torch.zeros(rows, C, device=eng.device, dtype=eng.dtype)
        out = 4 * planes
        self.has_down = stride > 1 or inpl != out
        self.c1 = _Lora(eng, p + "conv1.", inpl, planes, Ri, True)
        self.c3 = _Lora(eng, p + "conv3.", planes, out, Ro, True)
        self.bn1, self.bn2 = _BN(eng, p + "bn1.", planes, Ri), _BN(eng, p + "bn2.", planes, Ri)
        self.bn3 = _BN(eng, p + "bn3.", out, Ro)
        self.Kp = _round_up(9 * planes, eng.kq)
        # saved activations
        self.z1, self.a1, self.z2, self.a2 = e(Ri, planes), e(Ri, planes), e(Ri, planes), e(Ri, planes)
        self.a2p = e(Ro, planes) if stride > 1 else None
        self.z3, self.out = e(Ro, out), e(Ro, out)
        # gradients
        self.dz3, self.da2p, self.dz2, self.da1, self.dz1 = e(Ro, out), e(Ro, planes), e(Ri, planes), e(Ri, planes), e(Ri, planes)
        self.da2 = e(Ri, planes) if stride > 1 else None
        self.dx = e(Ri, inpl)
        if self.has_down:
            self.bnd = _BN(eng, p + "downsample.1.", out, Ro, aux=True)
            self.xp = e(Ro, inpl) if stride > 1 else None
            self.zd, self.idn, self.dzd, self.dxp = e(Ro, out), e(Ro, out), e(Ro, out), e(Ro, inpl)
            self.dxid = e(Ri, inpl) if stride > 1 else None
        else:
            self.gid = e(Ro, out)
        self.ev = torch.cuda.Event()
        self.rplans = {}                                          # images -> ReducePlan of this block's LoRA sites
        self.ev_d = [torch.cuda.Event() for _ in range(4)]        # downsample branch: fork / join, forward and backward

    def load(self, sd, putw) -> None:
        e, p = self.eng, self.p
        w1 = sd[p + "conv1.original_linear.weight"].reshape(self.planes, self.inpl)
        w3 = sd[p + "conv3.original_linear.weight"].reshape(4 * self.planes, self.planes)
        putw(p + "w1", e._w(w1)); putw(p + "w1t", e._wt(w1))
        putw(p + "w3", e._w(w3)); putw(p + "w3t", e._wt(w3))
        putw(p + "w2", e._w(e.conv3x3_rows(sd[p + "conv2.weight"], self.Kp)))
        putw(p + "w2b", e._w(e.conv3x3_rows_bwd(sd[p + "conv2.weight"], self.Kp)))
        if self.has_down:
            wd = sd[p + "downsample.0.weight"].reshape(4 * self.planes, self.inpl)
            putw(p + "wd", e._w(wd)); putw(p + "wdt", e._wt(wd))

    def forward(self, x: Tensor, images: int, attr: Optional[Tensor]) -> Tensor:
        e, p, W = self.eng, self.p, self.eng.rnw
        Hi, Ho = self.Hin, self.Hout
        ri, ro = images * Hi * Hi, images * Ho * Ho
        z1, a1, z2, a2, z3, out = self.z1[:ri], self.a1[:ri], self.z2[:ri], self.a2[:ri], self.z3[:ro], self.out[:ro]
        # BatchNorm statistics (training): the producing GEMM's epilogue leaves the column sums of its row tiles in
        # stat_buf[k], and the BatchNorm skips its own pass over the tensor (split-K convolutions excepted)
        sb, nt = e.stat_buf, e.stat_rows
        t_i, t_o = nt(ri), nt(ro)
        # The downsample branch (4 of the 16 blocks) shares nothing with conv1 .. conv3 until bn3 adds it: it runs on
        # the auxiliary stream beside them (its BatchNorm has scratch of its own), fork here, join in front of bn3.
        main = torch.cuda.current_stream(e.device)
        fork = self.has_down and getattr(e, "down_on_side", True) and e.aux_stream != main
        if fork:
            e._ev_record(self.ev_d[0], main)
            e._ev_wait(e.aux_stream, self.ev_d[0])
            with e._on(e.aux_stream):
                self._down_fwd(x, images, t_o)
            e._ev_record(self.ev_d[1], e.aux_stream)
        self.c1.fwd(x, W[p + "w1"], z1, attr, Hi * Hi, colstats=sb[0] if t_i else None)
        self.bn1.fwd(z1, a1, True, part=sb[0], part_rows=t_i)
        t2 = ops.conv3x3(a1, W[p + "w2"], z2, images, Hi, Hi, e.zero16, e.splitk, colstats=sb[0] if t_i else None)
        self.bn2.fwd(z2, a2, True, part=sb[0], part_rows=t2)
        a = a2
        if self.stride > 1:
            a = self.a2p[:ro]
            ops.avgpool2(a2, a, images, Hi, Hi)
        self.c3.fwd(a, W[p + "w3"], z3, attr, Ho * Ho, colstats=sb[0] if t_o else None)
        idn = x
        if self.has_down:
            idn = self.idn[:ro]
            if fork:
                e._ev_wait(main, self.ev_d[1])                    # join: the downsample branch's bn output
            else:
                self._down_fwd(x, images, t_o)
        self.bn3.fwd(z3, out, True, res=idn, part=sb[0], part_rows=t_o)   # relu(bn3(conv3) + identity)
        return out

    def _down_fwd(self, x: Tensor, images: int, t_o: int) -> None:
        """downsample: AvgPool2d(stride) -> 1x1 convolution -> BatchNorm (clip/model.py:29-36), into self.idn"""
        e, p, W = self.eng, self.p, self.eng.rnw
        ro = images * self.Hout * self.Hout
        xi = x
        if self.stride > 1:
            xi = self.xp[:ro]
            ops.avgpool2(x, xi, images, self.Hin, self.Hin)
        ops.gemm_nt(xi, W[p + "wd"], self.zd[:ro], colstats=e.stat_buf[1] if t_o else None)
        self.bnd.fwd(self.zd[:ro], self.idn[:ro], False, part=e.stat_buf[1], part_rows=t_o)

    def backward(self, g: Tensor, x: Tensor, images: int, attr: Optional[Tensor], bn3_rows: int = 0,
                 consumer: Optional["_Bneck"] = None):
        """g = dL/d(block output) -> (dL/d(block input), partial rows of the CONSUMER's bn3 sums left in stat_buf[0]).
        bn3_rows > 0: the dX product of the block behind this one (which produced g) has left this block's bn3 backward
        sums in stat_buf[0] and - identity-skip blocks - the ReLU-masked gradient in self.gid (FFM_EPI_BNBWD)."""
        e, p, W = self.eng, self.p, self.eng.rnw
        Hi, Ho = self.Hin, self.Hout
        ri, ro = images * Hi * Hi, images * Ho * Ho
        out = self.out[:ro]
        # (identity-skip blocks: the ReLU-masked gradient that goes on beside bn3 leaves the same pass)
        self.bn3.bwd(g, out, self.z3[:ro], self.dz3[:ro], g_out=None if (self.has_down or bn3_rows) else self.gid[:ro],
                     part=e.stat_buf[0], part_rows=bn3_rows)
        main = torch.cuda.current_stream(e.device)
        fork = self.has_down and getattr(e, "down_on_side", True) and e.aux_stream != main
        if self.has_down:
            gid = self.dxid[:ri] if self.stride > 1 else self.dxp[:ro]
            if fork:                                              # beside conv3 .. conv1's dX chain, joined at conv1's dX
                e._ev_record(self.ev_d[2], main)
                e._ev_wait(e.aux_stream, self.ev_d[2])
                with e._on(e.aux_stream):
                    self._down_bwd(g, images)
                e._ev_record(self.ev_d[3], e.aux_stream)
            else:
                self._down_bwd(g, images)
        else:
            gid = self.gid[:ro]
        a = self.a2p[:ro] if self.stride > 1 else self.a2[:ri]
        side = getattr(e, "grads_on_side", True)
        # stride 1: conv3's dX product writes the gradient of relu(bn2(.)) itself - its epilogue leaves bn2's backward column
        # sums of its row tiles behind (FFM_EPI_BNBWD), and bn2's backward skips its pass over dy / x / mask (13 of RN50's 16
        # blocks; FFM_BN_BWD_FUSED=0: A/B runs)
        t2 = e.stat_rows(ri) if (self.stride == 1 and self.c3.fused and getattr(e, "bn_bwd_fused", True)) else 0
        self.c3.bwd(self.dz3[:ro], W[p + "w3t"], self.da2p[:ro], a, attr, Ho * Ho, defer=side,
                    bnbwd=(self.z2[:ri], self.a2[:ri], self.bn2.mean, self.bn2.rstd) if t2 else None,
                    colstats=e.stat_buf[0] if t2 else None)
        da2 = self.da2p[:ro]
        if self.stride > 1:
            da2 = self.da2[:ri]
            ops.avgpool2(self.da2p[:ro], da2, images, Hi, Hi, backward=True)
        self.bn2.bwd(da2, self.a2[:ri], self.z2[:ri], self.dz2[:ri], part=e.stat_buf[0], part_rows=t2)
        # dX = conv3x3(dY; w') is the gradient of relu(bn1(.)): the same for bn1 where the convolution is not split over K
        # (conv3x3 returns the partial rows it wrote: one per 128-row tile, or - launches split over K, layer3 / layer4 -
        # one per 8 / 32 rows from the split-K reduction kernel; 0 only where that would exceed 4096 partial rows)
        fuse1 = e.stat_rows(ri) > 0 and getattr(e, "bn_bwd_fused", True)
        t1 = ops.conv3x3(self.dz2[:ri], W[p + "w2b"], self.da1[:ri], images, Hi, Hi, e.zero16, e.splitk,
                         colstats=e.stat_buf[0] if fuse1 else None,
                         bnbwd=(self.z1[:ri], self.a1[:ri], self.bn1.mean, self.bn1.rstd) if fuse1 else None)
        self.bn1.bwd(self.da1[:ri], self.a1[:ri], self.z1[:ri], self.dz1[:ri], part=e.stat_buf[0], part_rows=t1 if fuse1 else 0)
        if fork:
            e._ev_wait(main, self.ev_d[3])                        # join: the downsample branch's input gradient
        # this block's dX IS dL/d(output) of the block in front of it: its bn3's backward sums (and, for an identity-skip
        # block, the masked gradient it passes on) leave with this product's epilogue
        t3 = e.stat_rows(ri) if (consumer is not None and self.c1.fused and getattr(e, "bn_bwd_fused", True)) else 0
        self.c1.bwd(self.dz1[:ri], W[p + "w1t"], self.dx[:ri], x, attr, Hi * Hi, res=gid, defer=side,
                    bnbwd=(consumer.z3[:ri], consumer.out[:ri], consumer.bn3.mean, consumer.bn3.rstd,
                           None if consumer.has_down else consumer.gid[:ri]) if t3 else None,
                    colstats=e.stat_buf[0] if t3 else None)
        if side:
            # off the dX chain: this block's four rank-r reductions (every operand is a per-block buffer that stays put
            # until the next step's forward)
            main = torch.cuda.current_stream(e.device)
            e._ev_record(self.ev, main)
            e._ev_wait(e.grad_stream, self.ev)
            with e._on(e.grad_stream):
                self.c3.grads(self.dz3[:ro], a)
                self.c1.grads(self.dz1[:ri], x)
                # ... and the sum of this block's partials behind them (round 4: ONE launch for every site of the trunk at the
                # end of the backward pass sat in the step's tail for 92 us - layer1's sites alone have 784 partial rows)
                self.reduce_plan(images).run()
        return self.dx[:ri], t3

    def reduce_plan(self, images: int) -> "ops.ReducePlan":
        if images not in self.rplans:
            ent = []
            for site, H in self.loras():
                ent += site.reduce_entries(images * H * H)
            self.rplans[images] = ops.ReducePlan(ent, self.eng.device)
        return self.rplans[images]

    def _down_bwd(self, g: Tensor, images: int) -> None:
        e, p, W = self.eng, self.p, self.eng.rnw
        ri, ro = images * self.Hin * self.Hin, images * self.Hout * self.Hout
        self.bnd.bwd(g, self.out[:ro], self.zd[:ro], self.dzd[:ro])
        ops.gemm_nt(self.dzd[:ro], W[p + "wdt"], self.dxp[:ro])
        if self.stride > 1:
            ops.avgpool2(self.dxp[:ro], self.dxid[:ri], images, self.Hin, self.Hin, backward=True)

    def loras(self):
        return [(self.c1, self.Hin), (self.c3, self.Hout)]


class RN50Engine(FairLoRAEngine):
    """FairLoRAEngine with CLIP's ModifiedResNet as the image tower."""

    # ------------------------------------------------------------------ setup --
    def _init_vision(self, max_images: int) -> None:
        cfg, v, dt, dev = self.cfg, self.cfg.vision, self.dtype, self.device
        if cfg.dim_per_3d_slice:
            raise ValueError("the 3D OCT front end is built for the ViT tower only")
        if v.image_size % 32:
            raise ValueError("ModifiedResNet needs an image size divisible by 32")
        self.is3d = False
        self.vis = None
        self.kq = 64 if _is16(dt) else 32             # GEMM K granularity (128 bytes)
        self.rnw: Dict[str, Tensor] = {}
        self.bns: List[_BN] = []
        self.bn_scratch = self.bn_cmax = self.stat_scratch = 0
        self.bn_bwd_fused = os.environ.get("FFM_BN_BWD_FUSED", "1") != "0"      # _Bneck.backward: bn2's sums from conv3's dX product
        self.bn_training = False
        self.pack_entries: list = []
        e = lambda rows, C: torch.zeros(rows, C, device=dev, dtype=dt)
        w, H1 = v.width, v.image_size // 2
        R1 = max_images * H1 * H1
        self.H1, self.H2 = H1, H1 // 2
        self.ones_s = torch.ones(1, cfg.lora.rank, device=dev, dtype=torch.float32)
        # stem: conv1 (stride 2) .. conv3, three BatchNorms, 2x2 average pool
        self.Kp1, self.Kp2, self.Kp3 = _round_up(27, self.kq), _round_up(9 * (w // 2), self.kq), _round_up(9 * w, self.kq)
        self.cols1 = e(R1, self.Kp1)
        self.zero16 = torch.zeros(64, device=dev, dtype=dt)       # source of the zero padding of the implicit GEMMs
        # split-K partial tiles of the deep, small-map convolutions (layer3 / layer4): up to 8 x [M, N] fp32
        self.splitk = torch.zeros(8 * 160 * 128 * 128 // 4, device=dev, dtype=torch.float32)
        self.sbn = [_BN(self, f"image_encoder.bn{i}.", c, R1) for i, c in ((1, w // 2), (2, w // 2), (3, w))]
        self.sz = [e(R1, w // 2), e(R1, w // 2), e(R1, w)]
        self.sa = [e(R1, w // 2), e(R1, w // 2), e(R1, w)]
        self.dsa = [e(R1, w // 2), e(R1, w // 2), e(R1, w)]
        self.dsz = [e(R1, w // 2), e(R1, w // 2), e(R1, w)]
        self.p0 = e(max_images * self.H2 * self.H2, w)
        # residual stages
        self.blocks: List[_Bneck] = []
        inpl, H = w, self.H2
        for li, nblk in enumerate(v.layers):
            planes = w * (2 ** li)
            for j in range(nblk):
                stride = 2 if (li > 0 and j == 0) else 1
                blk = _Bneck(self, f"image_encoder.layer{li + 1}.{j}.", inpl, planes, stride, H, max_images)
                self.blocks.append(blk)
                inpl, H = planes * 4, H // stride
        assert inpl == v.embed_dim and H == v.spacial
        # attention pool
        E, L = v.embed_dim, v.tokens
        T = max_images * L
        ap = "image_encoder.attnpool."
        self.ap = {n: _Lora(self, f"{ap}{n}_proj.", E, v.out_dim if n == "c" else E, T, False) for n in "qkvc"}
        self.tok, self.qkv, self.att_o = e(T, E), e(T, 3 * E), e(T, E)
        self.lse = torch.zeros(max_images * v.heads * L, device=dev, dtype=torch.float32)
        self.delta = torch.zeros_like(self.lse)
        self.d_o, self.dqkv, self.dtok = e(T, E), e(T, 3 * E), [e(T, E), e(T, E)]
        self.dx4 = e(max_images * v.spacial * v.spacial, E)
        self.ev_ap = torch.cuda.Event()
        self.aux_stream = self._aux0 = torch.cuda.Stream(device=dev)      # the downsample branches (_Bneck.forward / backward)
        self.bn_part_d = torch.zeros(self.bn_scratch, device=dev, dtype=torch.float32)
        self.bn_k12_d = torch.zeros(2 * self.bn_cmax, device=dev, dtype=torch.float32)
        self.bn_part = torch.zeros(self.bn_scratch, device=dev, dtype=torch.float32)
        # column sums left behind by the GEMMs that produce a BatchNorm's input ([1]: the downsample branch, whose
        # product sits between conv3 and bn3)
        self.stat_buf = [torch.zeros(max(self.stat_scratch, 1), device=dev, dtype=torch.float32) for _ in range(2)]
        self.bn_k12 = torch.zeros(2 * self.bn_cmax, device=dev, dtype=torch.float32)
        self.rn_plans: Dict[int, ops.ReducePlan] = {}
        # num_batches_tracked of every BatchNorm2d, in self.bns order (one add per training step for all of them)
        self.nbt = torch.zeros(len(self.bns), device=dev, dtype=torch.int64)

    def _init_vision_late(self) -> None:
        from .engine import SOperands
        if getattr(self.cfg.lora, "lora_type", "FairLoRA") != "FairLoRA":
            raise NotImplementedError(self.cfg.lora.lora_type)    # trainers/GLP_OT_SVLoRA.py:561-567: ResNet knows FairLoRA only
        self.sops = SOperands(self.params, [site.prefix for blk in self.blocks for site, _ in blk.loras()], self.cfg,
                              self.device)
        self.fused_rank = 0 < self.cfg.lora.rank <= 16
        self.pack_plan = ops.PackPlan(self.pack_entries, self.dtype, self.device) if self.fused_rank else None

    def _n_layer_events(self) -> int:
        return 1

    def set_overlap(self, on: bool) -> None:
        super().set_overlap(on)
        self.aux_stream = self._aux0 if on else torch.cuda.current_stream(self.device)

    def conv3x3_rows(self, w: Tensor, Kp: int) -> Tensor:
        """[Cout, Cin, 3, 3] -> [Cout, Kp] with k = (ky*3 + kx)*Cin + c, zero padded (the order ffm_im2col3x3 writes)."""
        co = w.shape[0]
        rows = w.float().permute(0, 2, 3, 1).reshape(co, -1)
        out = torch.zeros(co, Kp, dtype=torch.float32)
        out[:, :rows.shape[1]] = rows
        return out

    def conv3x3_rows_bwd(self, w: Tensor, Kp: int) -> Tensor:
        """[Cout, Cin, 3, 3] -> [Cin, Kp] with k = (ky'*3 + kx')*Cout + co holding w[co, ci, 2-ky', 2-kx']: the input
        gradient of the convolution is the same implicit GEMM applied to dY with this matrix."""
        ci = w.shape[1]
        rows = w.float().flip(2, 3).permute(1, 2, 3, 0).reshape(ci, -1)
        out = torch.zeros(ci, Kp, dtype=torch.float32)
        out[:, :rows.shape[1]] = rows
        return out

    def _load_vision_frozen(self, sd: Dict[str, Tensor], put) -> None:
        v, ie = self.cfg.vision, "image_encoder."

        def putw(name: str, val: Tensor) -> None:                # stable addresses across reloads
            if name in self.rnw:
                self.rnw[name].copy_(val)
            else:
                self.rnw[name] = val

        w = v.width
        putw("s1", self._w(self.conv3x3_rows(sd[ie + "conv1.weight"], self.Kp1)))
        putw("s2", self._w(self.conv3x3_rows(sd[ie + "conv2.weight"], self.Kp2)))
        putw("s2b", self._w(self.conv3x3_rows_bwd(sd[ie + "conv2.weight"], self.Kp2)))
        putw("s3", self._w(self.conv3x3_rows(sd[ie + "conv3.weight"], self.Kp2)))
        putw("s3b", self._w(self.conv3x3_rows_bwd(sd[ie + "conv3.weight"], self.Kp3)))
        for blk in self.blocks:
            blk.load(sd, putw)
        ap = ie + "attnpool."
        putw("pos", self._w(sd[ap + "positional_embedding"]))
        for n in "qkvc":
            wt = sd[f"{ap}{n}_proj.original_linear.weight"]
            putw(f"ap_{n}", self._w(wt)); putw(f"ap_{n}t", self._wt(wt))
            putw(f"ap_{n}b", self._f(sd[f"{ap}{n}_proj.original_linear.bias"]))
        self.load_buffers(sd)

    # ------------------------------------------------------ BatchNorm buffers --
    def load_buffers(self, sd: Dict[str, Tensor]) -> None:
        """running_mean / running_var / num_batches_tracked of every BatchNorm2d (clip/model.py:17-36, 237-244)."""
        for bn in self.bns:
            bn.run_mean.copy_(sd[bn.prefix + "running_mean"].to(self.device, torch.float32))
            bn.run_var.copy_(sd[bn.prefix + "running_var"].to(self.device, torch.float32))
        self.nbt.copy_(torch.stack([sd[bn.prefix + "num_batches_tracked"].reshape(()).to(torch.int64).cpu()
                                    for bn in self.bns]))

    def buffer_views(self) -> Dict[str, Tensor]:
        """state_dict key -> the live device tensor (CustomCLIP registers these as its buffers)."""
        out = {}
        index = {bn.prefix: i for i, bn in enumerate(self.bns)}
        for k in buffer_keys(self.cfg):
            p, name = k.rsplit(".", 1)
            i = index[p + "."]
            out[k] = self.nbt[i] if name == "num_batches_tracked" else \
                (self.bns[i].run_mean if name == "running_mean" else self.bns[i].run_var)
        return out

    def buffers_flat(self) -> Tensor:
        """All BatchNorm buffers as one fp32 vector (means | variances | counters): the FedAvg exchange unit."""
        return torch.cat([bn.run_mean for bn in self.bns] + [bn.run_var for bn in self.bns] + [self.nbt.float()])

    def load_buffers_flat(self, t: Tensor) -> None:
        off = 0
        for which in ("run_mean", "run_var"):
            for bn in self.bns:
                getattr(bn, which).copy_(t[off:off + bn.C])
                off += bn.C
        self.nbt.copy_(t[off:off + len(self.bns)])                 # float -> int64 truncates, as load_state_dict does

    def buffer_state(self) -> Dict[str, Tensor]:
        return {k: v.clone() for k, v in self.buffer_views().items()}

    # ----------------------------------------------------------------- inputs --
    def _check_batch(self, image: Tensor) -> Tuple[int, int]:
        v = self.cfg.vision
        if not image.is_cuda or image.dtype != torch.float32:
            raise TypeError("image must be a float32 CUDA tensor of raw 0..255 values")
        b, c, h, w = image.shape
        if c != 3 or h != v.image_size or w != v.image_size:
            raise ValueError(f"expected [B,3,{v.image_size},{v.image_size}], got {tuple(image.shape)}")
        if b > self.max_images:
            raise ValueError(f"{b} images exceed the engine's max_images={self.max_images}")
        return b, 1

    def _load_inputs(self, image: Tensor, attr: Optional[Tensor], label: Optional[Tensor]):
        cfg = self.cfg
        image = self._as_f32(image)
        b, S = self._check_batch(image)
        ops.stem_im2col(image.contiguous(), self.cols1[:b * self.H1 * self.H1], 2, cfg.pixel_mean, cfg.pixel_std)
        if attr is not None:
            self.attr_i32[:b].copy_(attr)
        if label is not None:
            self.label_buf[:b].copy_(label)
        return b, S

    # ---------------------------------------------------------------- forward --
    def stat_rows(self, rows: int) -> int:
        """Row tiles of the 128x128 GEMM over `rows` rows when its epilogue should leave the BatchNorm column sums behind
        (training mode, at most 4096 partial rows: ffm_bn_fwd's limit); else 0."""
        t = (rows + 127) // 128
        return t if (self.bn_training and 0 < t <= 4096 and not getattr(self, "no_colstats", False)) else 0

    def _vision_forward(self, b: int, S: int, has_attr: bool, wait=None) -> None:
        cfg, v, W = self.cfg, self.cfg.vision, self.rnw
        H1, H2 = self.H1, self.H2
        r1 = b * H1 * H1
        a32 = self.attr_i32[:b] if has_attr else None
        sz, sa = [t[:r1] for t in self.sz], [t[:r1] for t in self.sa]
        if self.sops.glob:
            self._glue(self.sops.prepare)             # S_eff = S + S_global (GLOBAL_S)
        t1 = self.stat_rows(r1)
        ops.gemm_nt(self.cols1[:r1], W["s1"], sz[0], colstats=self.stat_buf[0] if t1 else None)
        self.sbn[0].fwd(sz[0], sa[0], True, part=self.stat_buf[0], part_rows=t1)
        for i, wn in ((1, "s2"), (2, "s3")):
            ti = ops.conv3x3(sa[i - 1], W[wn], sz[i], b, H1, H1, self.zero16, colstats=self.stat_buf[0] if t1 else None)
            self.sbn[i].fwd(sz[i], sa[i], True, part=self.stat_buf[0], part_rows=ti)
        x = self.p0[:b * H2 * H2]
        ops.avgpool2(sa[2], x, b, H1, H1)
        self._rank_operands_ready()                   # LoRA matrices -> GEMM rank operands (packed beside the stem)
        for blk in self.blocks:
            x = blk.forward(x, b, a32)
        # attention pool (clip/model.py:75-118): all HW + 1 tokens come back
        HW, L, E = v.spacial * v.spacial, v.tokens, v.embed_dim
        rows = b * L
        tok, qkv = self.tok[:rows], self.qkv[:rows]
        ops.attnpool_tokens(x, W["pos"], tok, b, HW)
        for i, n in enumerate("qkv"):
            self.ap[n].fwd(tok, W[f"ap_{n}"], qkv[:, i * E:(i + 1) * E], None, L, bias=W[f"ap_{n}b"])
        ops.attention_fwd(qkv, self.att_o[:rows], self.lse, b, L, v.heads, False)
        self.ap["c"].fwd(self.att_o[:rows], W["ap_c"], self.feat[:rows], None, L, bias=W["ap_cb"])
        if wait is not None:
            self._ev_wait(torch.cuda.current_stream(self.device), wait)     # text features ready
        self._head_forward(rows, b, L)

    # --------------------------------------------------------------- backward --
    def _vision_backward(self, b: int, S: int, has_attr: bool) -> None:
        cfg, v, W = self.cfg, self.cfg.vision, self.rnw
        HW, L, E = v.spacial * v.spacial, v.tokens, v.embed_dim
        rows = b * L
        a32 = self.attr_i32[:b] if has_attr else None
        tok, qkv, dqkv = self.tok[:rows], self.qkv[:rows], self.dqkv[:rows]
        side = getattr(self, "grads_on_side", True)
        self.ap["c"].bwd(self.dfeat[:rows], W["ap_ct"], self.d_o[:rows], self.att_o[:rows], None, L, defer=side)
        ops.attention_bwd(qkv, self.att_o[:rows], self.d_o[:rows], self.lse, self.delta, dqkv, b, L, v.heads, False)
        acc = None
        for i, n in enumerate("qkv"):
            dst = self.dtok[i & 1][:rows]
            self.ap[n].bwd(dqkv[:, i * E:(i + 1) * E], W[f"ap_{n}t"], dst, tok, None, L, res=acc, defer=side)
            acc = dst
        if side:
            # the attention pool's eight rank-r reductions beside the trunk's dX chain (dfeat, att_o, dqkv and tok stay
            # put until the next step)
            main = torch.cuda.current_stream(self.device)
            self._ev_record(self.ev_ap, main)
            self._ev_wait(self.grad_stream, self.ev_ap)
            with self._on(self.grad_stream):
                self.ap["c"].grads(self.dfeat[:rows], self.att_o[:rows])
                for i, n in enumerate("qkv"):
                    self.ap[n].grads(dqkv[:, i * E:(i + 1) * E], tok)
                self._reduce(b, "ap").run()
        g = self.dx4[:b * HW]
        ops.attnpool_tokens(acc, None, g, b, HW, backward=True)
        t3 = 0
        for i in range(len(self.blocks) - 1, -1, -1):
            x = self.blocks[i - 1].out[:b * self.blocks[i].Hin ** 2] if i > 0 else self.p0[:b * self.H2 * self.H2]
            g, t3 = self.blocks[i].backward(g, x, b, a32, bn3_rows=t3, consumer=self.blocks[i - 1] if i > 0 else None)
        # stem: only the BatchNorm weights / biases train, but their gradients need dX through conv3 and conv2
        H1 = self.H1
        r1 = b * H1 * H1
        sz, sa = [t[:r1] for t in self.sz], [t[:r1] for t in self.sa]
        dsa, dsz = [t[:r1] for t in self.dsa], [t[:r1] for t in self.dsz]
        ops.avgpool2(g, dsa[2], b, H1, H1, backward=True)
        self.sbn[2].bwd(dsa[2], sa[2], sz[2], dsz[2])
        # (the stem's two 3x3 dX products feed a BatchNorm backward each: its column sums leave with them, as in the blocks)
        fs = self.stat_rows(r1) > 0 and getattr(self, "bn_bwd_fused", True)
        for k, wn in ((1, "s3b"), (0, "s2b")):
            tk = ops.conv3x3(dsz[k + 1], W[wn], dsa[k], b, H1, H1, self.zero16, colstats=self.stat_buf[0] if fs else None,
                             bnbwd=(sz[k], sa[k], self.sbn[k].mean, self.sbn[k].rstd) if fs else None)
            self.sbn[k].bwd(dsa[k], sa[k], sz[k], dsz[k], part=self.stat_buf[0], part_rows=tk if fs else 0)
        if getattr(self, "grads_on_side", True):
            # the partial sums of every site -> params.grad, behind the reductions on the gradient stream
            main = torch.cuda.current_stream(self.device)
            self._ev_record(self.ev_layer[0], main)
            self._ev_wait(self.grad_stream, self.ev_layer[0])
            # (every block and the attention pool have summed their partials behind their own reductions by now)
            if self.sops.glob:
                self._glue(self.sops.finish, self.grad_stream)
            self._ev_record(self.ev_grads, self.grad_stream)
            self._ev_wait(main, self.ev_grads)
        else:
            self._reduce(b).run()
            if self.sops.glob:
                self._glue(self.sops.finish)

    def _reduce(self, b: int, what: str = "all") -> "ops.ReducePlan":
        """what: 'all' (every site, one launch: the arrangement without a gradient stream) or 'ap' (the attention pool's)."""
        if (b, what) not in self.rn_plans:
            v = self.cfg.vision
            ent = []
            if what == "all":
                for blk in self.blocks:
                    for site, H in blk.loras():
                        ent += site.reduce_entries(b * H * H)
            for n in "qkvc":
                ent += self.ap[n].reduce_entries(b * v.tokens)
            self.rn_plans[(b, what)] = ops.ReducePlan(ent, self.device)
        return self.rn_plans[(b, what)]

    # -------------------------------------------------------------------- API --
    def _step_body(self, b: int, S: int, has_attr: bool) -> None:
        self.bn_training = True                                   # model.train(): batch statistics, running stats move
        try:
            super()._step_body(b, S, has_attr)
        finally:
            self.bn_training = False

    def forward_backward(self, image: Tensor, attr: Optional[Tensor], label: Tensor) -> Dict[str, Tensor]:
        out = super().forward_backward(image, attr, label)
        self.nbt.add_(1)
        return out


def create_engine(cfg, state_dict, **kw) -> FairLoRAEngine:
    """The engine class for cfg's image tower (MODEL.BACKBONE.NAME: ViT-B/16 or RN50)."""
    cls = RN50Engine if hasattr(cfg.vision, "embed_dim") else FairLoRAEngine
    return cls(cfg, state_dict, **kw)
