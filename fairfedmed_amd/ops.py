"""Tensor-level wrappers over the C ABI (include/ffm_hip.h).

Each function takes CUDA (ROCm) tensors, checks device/dtype/contiguity and
enqueues the kernel on PyTorch's current stream.  There is no fallback: a CPU
tensor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

Tensor = torch.Tensor

# ---------------------------------------------------------------------------
# Launch recording.  Every buffer of the engine is static, so the ctypes arguments of a whole
# training step can be built once: while a recorder list is active, each C call is executed AND
# appended as a zero-argument callable; replaying the list costs ~1.5 us per launch on the host
# instead of ~16 us of Python (the eager loop is host-bound at ~7 ms/step for ~450 launches).
# ---------------------------------------------------------------------------
_REC = None


class record:
    def __init__(self, plan: list):
        self.plan = plan

    def __enter__(self):
        global _REC
        self.prev, _REC = _REC, self.plan
        return self.plan

    def __exit__(self, *a):
        global _REC
        _REC = self.prev


def recording() -> bool:
    return _REC is not None


def record_callable(fn) -> None:
    """Append a host-side step (event record/wait, PyTorch glue) to the active recording."""
    if _REC is not None:
        _REC.append(fn)


class _Launch:
    __slots__ = ("fn", "args", "name")

    def __init__(self, fn, args, name):
        self.fn, self.args, self.name = fn, args, name

    def __call__(self):
        rc = self.fn(*self.args)
        if rc:
            L.check(rc, self.name)


def _call(name: str, *args) -> None:
    fn = getattr(L.load(), name)
    if _REC is not None:
        _REC.append(_Launch(fn, args, name))
    rc = fn(*args)
    if rc:
        L.check(rc, name)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fairfedmed_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _f32(t: Optional[Tensor]):
    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
        raise TypeError("expected a contiguous fp32 tensor")
    return t


def _ld(t: Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D tensor"
    return t.stride(0)


class RankOp:
    """Arguments of FFM_EPI_RANKOP: the rank-r down projection computed inside the GEMM."""

    def __init__(self, rk: Tensor, S: Tensor, attr: Optional[Tensor], rows_per_sample: int, scaling: float,
                 lambda_group: float, t_out: Optional[Tensor] = None, ts_out: Optional[Tensor] = None,
                 t_fwd: Optional[Tensor] = None, ds_part: Optional[Tensor] = None, lw_wide: Optional[Tensor] = None,
                 lgrad: Optional[tuple] = None):
        self.rk, self.S, self.attr, self.rps = rk, S, attr, rows_per_sample
        self.scaling, self.lam = scaling, lambda_group
        self.t_out, self.ts_out, self.t_fwd, self.ds_part = t_out, ts_out, t_fwd, ds_part
        self.lw_wide = lw_wide                    # `lw` as [N, 32] rows in the activation dtype (PackPlan's wide output)
        # FFM_EPI_LGRAD (dX of c_proj): (lg_v [M, r], lg_part_c, lg_part_a) - the two large rank-r gradient partial
        # products leave with this launch's epilogue (gemm_lgrad_rows says whether its kernel can, and how many tiles)
        self.lgrad = lgrad


def gemm_tiles_m(M: int, N: int = 128, K: int = 128, flags: int = 0, rank: int = 0, dtype=torch.float32,
                 packed: bool = False) -> int:
    """Row tiles (= dS partial rows) of the kernel ffm_gemm_nt picks for this call."""
    return L.load().ffm_gemm_tiles_m(M, N, K, flags, rank, L.dtype_code(dtype), int(packed))


def gemm_lgrad_rows(M: int, N: int, K: int, rank: int, dtype, packed: bool = True) -> int:
    """Row tiles of the FFM_EPI_LGRAD partial products of the dX product of c_proj, or a negative code when the kernel
    ffm_gemm_nt picks for that call has no such epilogue."""
    flags = L.EPI_LORA | L.EPI_LORA_KR | L.EPI_DGELU | L.EPI_RANKOP
    return L.load().ffm_gemm_lgrad_rows(M, N, K, flags, rank, L.dtype_code(dtype), int(packed))


def gemm_tiles_n(M: int, N: int, K: int, flags: int = 0, rank: int = 0, dtype=torch.float32, packed: bool = False) -> int:
    """Column tiles (= rows of a ROWSTATS partial) of that kernel; negative when no kernel serves the flags."""
    return L.load().ffm_gemm_tiles_n(M, N, K, flags, rank, L.dtype_code(dtype), int(packed))


def gemm_tile_shape(M: int, N: int, K: int, flags: int = 0, rank: int = 0, dtype=torch.float32, packed: bool = False):
    """(configuration index or -1, tile rows, tile columns, waves per CU) of the kernel ffm_gemm_nt picks (diagnostics)."""
    import ctypes
    shp = (ctypes.c_int32 * 3)()
    cfg = L.load().ffm_gemm_tile_shape(M, N, K, flags, rank, L.dtype_code(dtype), int(packed), shp)
    return cfg, shp[0], shp[1], shp[2]


class LnIn:
    """FFM_EPI_LNIN: the LayerNorm in front of a product, folded into it.  part [np, M, 2]: partial row sums of the
    product's A rows (a producer's `rowstats`, or embed_lnpre's), c [N]: row sums of the gamma-scaled weight;
    mean / rstd [M]: optional outputs for the LayerNorm backward.  The caller passes the gamma-scaled weight as `b` /
    `b_packed` and d = W beta + bias as `bias`."""

    def __init__(self, part: Tensor, np_: int, c: Tensor, mean: Optional[Tensor] = None, rstd: Optional[Tensor] = None,
                 rk: Optional[Tensor] = None):
        self.part, self.np, self.c, self.mean, self.rstd = part, np_, c, mean, rstd
        self.rk = rk          # with a RankOp: [2, 16] corrections of its gamma-scaled rank operand (PackPlan's ln output)


class LnBwdStat:
    """FFM_EPI_LNB_STAT (the dX product of c_proj, with a RankOp that carries `lgrad`): also leave the two row sums of the
    LayerNorm backward that follows the NEXT dX product - part [tiles_n, M, 2] fp32 <- {0, sum_n c aux} per column tile.  The
    sums against the fixed vectors W gamma (LnIn's c) and d = W beta + b (the folded forward's bias) come out of the CONSUMER's
    rank operand: rows 14 / 15 of the rk it is packed with (PackPlan entry element 6)."""

    def __init__(self, part: Tensor):
        self.part = part


class LnBwdApply:
    """FFM_EPI_LNB_APPLY (the dX product of c_fc): store rstd (gamma g_h - c1/K - xhat c2/K) + res instead of g_h, with the
    row sums from a LnBwdStat producer's `part` (np column tiles), x the LayerNorm's input, mean / rstd its saved statistics,
    gamma its weight, rk [2, 16] the corrections {A^T gamma, A^T beta} (LnIn.rk of the forward), res the residual gradient."""

    def __init__(self, part: Tensor, np_: int, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, rk: Tensor, res: Tensor):
        self.part, self.np, self.x, self.gamma, self.mean, self.rstd, self.rk, self.res = part, np_, x, gamma, mean, rstd, rk, res


def pack_b(w: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """Frozen 16-bit weight [N, K] -> MFMA-fragment order for the panel GEMM (ffm_pack_b)."""
    _dev(w, out)
    assert L.is16(w.dtype) and w.dim() == 2 and w.stride(1) == 1
    N, K = w.shape
    if out is None:
        out = torch.empty(N * K, device=w.device, dtype=w.dtype)
    assert out.numel() == N * K and out.dtype == w.dtype and out.is_contiguous()
    _call("ffm_pack_b", L.ptr(w), L.ptr(out), N, K, w.stride(0), L.stream_ptr())
    return out


def gemm_nt(a: Tensor, b: Tensor, out: Tensor, *, bias=None, ts=None, lw=None, lw_is_kr=False, res=None,
            gelu_out=None, dgelu_aux=None, rankop: Optional[RankOp] = None, b_packed: Optional[Tensor] = None,
            x3: bool = False, rowstats: Optional[Tensor] = None, ln_in: Optional["LnIn"] = None,
            colstats: Optional[Tensor] = None, gelu_deriv: bool = False, bnbwd=None,
            sk_part: Optional[Tensor] = None, lnb_stat: Optional[LnBwdStat] = None,
            lnb_apply: Optional[LnBwdApply] = None) -> Tensor:
    """out = epilogue(a @ b.T);  a [M,K], b [N,K], out [M,N] (same dtype).  b_packed: pack_b(b), optional.
    x3 (float32 operands, at most 64 rows): FFM_F32_X3, the products as bf16 hi/lo pairs on the bf16 matrix cores; with
    `b` in float16: FFM_F32_X3_W16, the same on a weight rounded to IEEE half in memory (half the bytes).
    gelu_deriv (with gelu_out / dgelu_aux): `out` receives / `dgelu_aux` holds quick_gelu'(pre) instead of pre
    (ffm_gemm_args.gelu_deriv).
    bnbwd = (bn_x, bn_mask or None, mean, rstd[, gout]) with `colstats`: FFM_EPI_BNBWD - `out` is dL/dy of a train-mode BatchNorm
    (+ ReLU with output bn_mask) on bn_x, and `colstats` receives the backward's column sums {sum g, sum g xhat} per row tile
    (bn_bwd's part / part_rows) instead of the forward's {sum, sum of squares}."""
    _dev(a, b, out, bias, ts, lw, res, gelu_out, dgelu_aux, b_packed)
    w16 = x3 and b.dtype == torch.float16            # FFM_F32_X3_W16: float32 activations on a weight stored as IEEE half
    assert a.dtype == out.dtype and (b.dtype == a.dtype or w16) and (not x3 or a.dtype == torch.float32)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and tuple(out.shape) == (M, N)
    flags, rank = 0, 0
    if bias is not None:
        flags |= L.EPI_BIAS
        _f32(bias)
    ro = rankop
    if ro is not None:
        _dev(ro.rk, ro.S, ro.attr, ro.t_out, ro.ts_out, ro.t_fwd, ro.ds_part, ro.lw_wide)
        assert ro.lw_wide is None or (ro.lw_wide.dtype == a.dtype and tuple(ro.lw_wide.shape) == (N, 32)
                                      and ro.lw_wide.is_contiguous())
        flags |= L.EPI_LORA | L.EPI_RANKOP | (L.EPI_LORA_KR if lw_is_kr else 0)
        rank = ro.S.shape[1]
        assert ro.rk.dtype == a.dtype and tuple(ro.rk.shape) == (16, K) and ro.rk.is_contiguous()
        assert lw.numel() == rank * N and (ro.attr is None or ro.attr.dtype == torch.int32)
        _f32(lw), _f32(ro.S), _f32(ro.t_out), _f32(ro.ts_out), _f32(ro.t_fwd), _f32(ro.ds_part)
    elif ts is not None:
        flags |= L.EPI_LORA | (L.EPI_LORA_KR if lw_is_kr else 0)
        _f32(ts), _f32(lw)
        rank = ts.shape[1]
        assert ts.shape[0] >= M and lw.numel() == rank * N
    for extra in (res, gelu_out, dgelu_aux):
        if extra is not None:
            assert extra.dtype == out.dtype and _ld(extra) == _ld(out)
    if res is not None:
        flags |= L.EPI_RESIDUAL
    if gelu_out is not None:
        flags |= L.EPI_GELU
    if dgelu_aux is not None:
        flags |= L.EPI_DGELU
    if ro is not None:
        extra = (L.ptr(ro.rk), L.ptr(ro.S), L.ptr(ro.attr), L.ptr(ro.t_out), L.ptr(ro.ts_out), L.ptr(ro.t_fwd),
                 L.ptr(ro.ds_part), ro.S.shape[0], ro.rps, ro.scaling, ro.lam)
    else:
        extra = (None, None, None, None, None, None, None, 0, 0, 0.0, 0.0)
    lnx = (None, None, None, None, 0, int(gelu_deriv), None)
    if rowstats is not None:                       # [tiles_n, M, 2] fp32 partial row sums of the stored output
        _dev(rowstats)
        flags |= L.EPI_ROWSTATS
        assert _f32(rowstats).numel() >= 2 * M * max(1, gemm_tiles_n(M, N, K, flags, rank, a.dtype, b_packed is not None))
    if ln_in is not None:
        _dev(ln_in.part, ln_in.c, ln_in.mean, ln_in.rstd)
        flags |= L.EPI_LNIN
        assert _f32(ln_in.part).numel() >= 2 * M * ln_in.np and _f32(ln_in.c).numel() == N and bias is not None
        _dev(ln_in.rk)
        assert (ro is None) == (ln_in.rk is None) and (ln_in.rk is None or _f32(ln_in.rk).numel() == 32)
        lnx = (L.ptr(ln_in.part), L.ptr(ln_in.c), L.ptr(_f32(ln_in.mean)), L.ptr(_f32(ln_in.rstd)), ln_in.np, int(gelu_deriv),
               L.ptr(ln_in.rk))
    lgx = (None, None, None)
    if ro is not None and ro.lgrad is not None:
        lg_v, lg_c, lg_a = ro.lgrad
        _dev(lg_v, lg_c, lg_a)
        flags |= L.EPI_LGRAD
        rows = gemm_lgrad_rows(M, N, K, rank, a.dtype, b_packed is not None)
        assert rows > 0 and dgelu_aux is not None and not gelu_deriv, "FFM_EPI_LGRAD: ask gemm_lgrad_rows first"
        assert _f32(lg_v).shape[0] >= M and lg_v.shape[1] == rank
        assert min(_f32(lg_c).numel(), _f32(lg_a).numel()) >= rows * N * rank
        lgx = (L.ptr(lg_v), L.ptr(lg_c), L.ptr(lg_a))
    bnx = (None, None, None, None, None)
    if bnbwd is not None:
        bn_x, bn_mask, bn_mean, bn_rstd = bnbwd[:4]
        bn_gout = bnbwd[4] if len(bnbwd) > 4 else None       # optional: g = out * (bn_mask > 0), like bn_bwd's g_out
        _dev(bn_x, bn_mask, bn_mean, bn_rstd, bn_gout)
        assert bn_gout is None or (bn_gout.dtype == out.dtype and tuple(bn_gout.shape) == (M, N) and _ld(bn_gout) == _ld(out))
        assert colstats is not None and bn_x.dtype == out.dtype and tuple(bn_x.shape) == (M, N) and _ld(bn_x) == _ld(out)
        assert bn_mask is None or (bn_mask.dtype == out.dtype and tuple(bn_mask.shape) == (M, N) and _ld(bn_mask) == _ld(out))
        assert _f32(bn_mean).numel() == N and _f32(bn_rstd).numel() == N
        flags |= L.EPI_BNBWD
        bnx = (L.ptr(bn_x), L.ptr(bn_mask), L.ptr(bn_mean), L.ptr(bn_rstd), L.ptr(bn_gout))
    lnb = (None, None, None, 0, 0, None, None)
    res_ptr = L.ptr(res)
    if lnb_stat is not None:
        _dev(lnb_stat.part)
        flags |= L.EPI_LNB_STAT
        tn = gemm_tiles_n(M, N, K, flags, rank, a.dtype, b_packed is not None)
        assert tn > 0 and (flags & L.EPI_LGRAD), "FFM_EPI_LNB_STAT rides on the FFM_EPI_LGRAD epilogue: ask gemm_tiles_n first"
        assert _f32(lnb_stat.part).numel() >= 2 * M * tn
        lnb = (None, None, L.ptr(lnb_stat.part), 0, 0, None, None)
    if lnb_apply is not None:
        la = lnb_apply
        _dev(la.part, la.x, la.gamma, la.mean, la.rstd, la.rk, la.res)
        assert res is None and ln_in is None and bias is None
        flags |= L.EPI_LNB_APPLY
        assert 0 < la.np <= (8 if ro is not None else 24) and _f32(la.part).numel() >= 2 * M * la.np and _f32(la.gamma).numel() == N
        assert la.x.dtype == out.dtype and tuple(la.x.shape) == (M, N) and _ld(la.x) == _ld(out)
        assert la.res.dtype == out.dtype and tuple(la.res.shape) == (M, N) and _ld(la.res) == _ld(out)
        assert _f32(la.mean).numel() >= M and _f32(la.rstd).numel() >= M and (ro is None or _f32(la.rk).numel() == 32)
        lnb = (None, None, L.ptr(la.part), la.np, 0, L.ptr(la.x), L.ptr(la.gamma))
        lnx = (None, None, L.ptr(la.mean), L.ptr(la.rstd), 0, int(gelu_deriv), L.ptr(la.rk))
        res_ptr = L.ptr(la.res)
    args = L.GemmArgs(L.ptr(a), L.ptr(b), L.ptr(out), M, N, K, _ld(a), _ld(b), _ld(out), flags, rank,
                      L.ptr(bias), L.ptr(ts), L.ptr(lw), res_ptr, L.ptr(gelu_out), L.ptr(dgelu_aux), *extra,
                      L.ptr(b_packed), L.ptr(ro.lw_wide) if ro is not None else None, L.ptr(rowstats), *lnx,
                      L.ptr(_f32(colstats)), *lgx, *bnx, L.ptr(_f32(sk_part)), *lnb)
    if sk_part is not None:                        # x3 products split over K: scratch for the partial tiles (gemm_splitk_floats)
        _dev(sk_part)
        assert x3 and sk_part.numel() >= gemm_splitk_floats(M, N, K, w16)
    if colstats is not None:                       # [tiles_m, 2, N] fp32 column sums of the stored output (128x128 kernel)
        _dev(colstats)
        assert b_packed is None and colstats.numel() >= 2 * N * ((M + 127) // 128)
    assert b_packed is None or (b_packed.numel() == N * K and b_packed.dtype == b.dtype)
    _call("ffm_gemm_nt", C.byref(args), (L.F32_X3_W16 if w16 else L.F32_X3) if x3 else L.dtype_code(a.dtype), L.stream_ptr())
    return out


def gemm_splitk_floats(M: int, N: int, K: int, w16: bool = False) -> int:
    """floats of gemm_nt(x3=True, sk_part=...)'s scratch for this product; 0: it runs as one launch."""
    return int(L.load().ffm_gemm_splitk_floats(M, N, K, L.F32_X3_W16 if w16 else L.F32_X3))


def layernorm_fwd(x: Tensor, y: Tensor, gamma: Tensor, beta: Tensor, mean: Optional[Tensor] = None,
                  rstd: Optional[Tensor] = None) -> Tensor:
    _dev(x, y, gamma, beta, mean, rstd)
    rows, width = x.shape
    assert x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype
    _call("ffm_layernorm_fwd", L.ptr(x), L.ptr(y), L.ptr(_f32(gamma)), L.ptr(_f32(beta)), L.ptr(_f32(mean)),
                                       L.ptr(_f32(rstd)), rows, width, L.dtype_code(x.dtype), L.stream_ptr())
    return y


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, res: Optional[Tensor],
                  out: Tensor) -> Tensor:
    _dev(dy, x, gamma, mean, rstd, res, out)
    rows, width = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and out.is_contiguous()
    assert dy.dtype == x.dtype == out.dtype and (res is None or (res.dtype == x.dtype and res.is_contiguous()))
    _call("ffm_layernorm_bwd", L.ptr(dy), L.ptr(x), L.ptr(_f32(gamma)), L.ptr(_f32(mean)), L.ptr(_f32(rstd)),
                                       L.ptr(res), L.ptr(out), rows, width, L.dtype_code(x.dtype), L.stream_ptr())
    return out


def patchify(img: Tensor, cols: Tensor, patch: int, mean3, std3, prenormalised: bool = False) -> Tensor:
    _dev(img, cols)
    B, Cc, H, W = img.shape
    assert Cc == 3 and img.dtype == torch.float32 and img.is_contiguous() and cols.is_contiguous()
    m = (C.c_float * 3)(*[float(v) for v in mean3])
    s = (C.c_float * 3)(*[float(v) for v in std3])
    _call("ffm_patchify", L.ptr(img), L.ptr(cols), B, H, W, patch, m, s, int(prenormalised),
                                  L.dtype_code(cols.dtype), L.stream_ptr())
    return cols


def embed_lnpre(patch: Tensor, cls: Tensor, pos: Tensor, gamma: Tensor, beta: Tensor, x: Tensor, B: int,
                Ltok: int, rowstat: Optional[Tensor] = None) -> Tensor:
    """rowstat: optional fp32 [>= B*Ltok, 2], receives {sum, sum of squares} of every output row (LnIn.part, np = 1)."""
    _dev(patch, cls, pos, gamma, beta, x, rowstat)
    width = x.shape[1]
    assert patch.dtype == cls.dtype == pos.dtype == x.dtype
    assert rowstat is None or rowstat.numel() >= 2 * B * Ltok
    _call("ffm_embed_lnpre", L.ptr(patch), L.ptr(cls), L.ptr(pos), L.ptr(_f32(gamma)), L.ptr(_f32(beta)),
                                     L.ptr(x), L.ptr(_f32(rowstat)), B, Ltok, width, L.dtype_code(x.dtype), L.stream_ptr())
    return x


def slice_blocks(H: int, W: int) -> int:
    return L.load().ffm_slice_blocks(H, W)


def slice_wgrad_blocks(H: int, W: int) -> int:
    """Partial rows per ViT image of slice_bwd's wpart [N * blocks, 3 D 25 + 3]."""
    return L.load().ffm_slice_wgrad_blocks(H, W)


def slice_bwd_ab_blocks() -> int:
    return L.load().ffm_slice_bwd_ab_blocks()


def slice_conv_fwd(img: Tensor, w: Tensor, bias: Tensor, conv: Tensor, mm_part: Tensor, mnmx: Tensor, cnt: Tensor,
                   D: int) -> None:
    """img: fp32 [B, S*D, H, W] raw 0..255, read as [B*S, D, H, W]."""
    _dev(img, w, bias, conv, mm_part, mnmx, cnt)
    B, Cc, H, W = img.shape
    assert img.dtype == torch.float32 and img.is_contiguous() and Cc % D == 0 and cnt.dtype == torch.int32
    _call("ffm_slice_conv_fwd", L.ptr(img), L.ptr(_f32(w)), L.ptr(_f32(bias)), L.ptr(_f32(conv)), L.ptr(_f32(mm_part)),
          L.ptr(_f32(mnmx)), L.ptr(cnt), B * Cc // D, D, H, W, L.stream_ptr())


def patchify_minmax(conv: Tensor, mnmx: Tensor, cnt: Tensor, cols: Tensor, patch: int, mean3, std3) -> None:
    _dev(conv, mnmx, cnt, cols)
    N, _, H, W = conv.shape
    m = (C.c_float * 3)(*[float(v) for v in mean3])
    s = (C.c_float * 3)(*[float(v) for v in std3])
    _call("ffm_patchify_minmax", L.ptr(_f32(conv)), L.ptr(_f32(mnmx)), L.ptr(cnt), L.ptr(cols), N, H, W, patch, m, s,
          L.dtype_code(cols.dtype), L.stream_ptr())


def embed_lnpre_bwd(dx: Tensor, patch: Tensor, pos: Tensor, gamma: Tensor, dpatch: Tensor, B: int, Ltok: int) -> None:
    _dev(dx, patch, pos, gamma, dpatch)
    assert dx.dtype == patch.dtype == pos.dtype == dpatch.dtype
    _call("ffm_embed_lnpre_bwd", L.ptr(dx), L.ptr(patch), L.ptr(pos), L.ptr(_f32(gamma)), L.ptr(dpatch), B, Ltok,
          dx.shape[1], L.dtype_code(dx.dtype), L.stream_ptr())


def slice_bwd(dcols: Tensor, img: Tensor, conv: Tensor, mnmx: Tensor, cnt: Tensor, dconv: Tensor, ab_part: Tensor,
              gmm: Tensor, wpart: Tensor, D: int, patch: int, std3) -> None:
    _dev(dcols, img, conv, mnmx, cnt, dconv, ab_part, gmm, wpart)
    N, _, H, W = conv.shape
    s = (C.c_float * 3)(*[float(v) for v in std3])
    _call("ffm_slice_bwd", L.ptr(dcols), L.ptr(img), L.ptr(_f32(conv)), L.ptr(_f32(mnmx)), L.ptr(cnt), L.ptr(_f32(dconv)),
          L.ptr(_f32(ab_part)), L.ptr(_f32(gmm)), L.ptr(_f32(wpart)), N, D, H, W, patch, s, L.dtype_code(dcols.dtype),
          L.stream_ptr())


# ---------------------------------------------------------------- RN50 trunk (NHWC rows) ----
def conv_out(h: int, stride: int) -> int:
    return (h - 1) // stride + 1


def stem_im2col(img: Tensor, cols: Tensor, stride: int, mean3, std3) -> None:
    _dev(img, cols)
    B, Cc, H, W = img.shape
    assert Cc == 3 and img.dtype == torch.float32 and img.is_contiguous() and cols.is_contiguous()
    m = (C.c_float * 3)(*[float(v) for v in mean3])
    s = (C.c_float * 3)(*[float(v) for v in std3])
    _call("ffm_stem_im2col", L.ptr(img), L.ptr(cols), B, H, W, stride, cols.shape[1], m, s, L.dtype_code(cols.dtype),
          L.stream_ptr())


def im2col3x3(x: Tensor, cols: Tensor, B: int, H: int, W: int, stride: int) -> None:
    _dev(x, cols)
    assert x.is_contiguous() and cols.is_contiguous() and x.dtype == cols.dtype and x.shape[0] == B * H * W
    _call("ffm_im2col3x3", L.ptr(x), L.ptr(cols), B, H, W, x.shape[1], stride, cols.shape[1], L.dtype_code(x.dtype),
          L.stream_ptr())


def col2im3x3(dcols: Tensor, dx: Tensor, B: int, H: int, W: int, stride: int) -> None:
    _dev(dcols, dx)
    assert dx.is_contiguous() and dcols.is_contiguous() and dx.dtype == dcols.dtype and dx.shape[0] == B * H * W
    _call("ffm_col2im3x3", L.ptr(dcols), L.ptr(dx), B, H, W, dx.shape[1], stride, dcols.shape[1], L.dtype_code(dx.dtype),
          L.stream_ptr())


def conv3x3(x: Tensor, w: Tensor, out: Tensor, B: int, H: int, W: int, zeros: Tensor,
            scratch: Optional[Tensor] = None, colstats: Optional[Tensor] = None, bnbwd=None) -> int:
    """3x3 / pad 1 / stride 1 convolution on NHWC rows as an implicit GEMM (ffm_conv3x3_nhwc): x [B*H*W, C],
    w [N, Kp] with k = (ky*3 + kx)*C + c, out [B*H*W, N].  scratch: fp32 buffer for split-K partial tiles (optional).
    colstats: optional fp32 buffer for the column sums of the output's row tiles; returns the number of partial rows
    written into it (0 when the launch is split over K or colstats is None: the BatchNorm forms its own sums).
    bnbwd = (bn_x, bn_mask or None, mean, rstd): `out` is dL/dy of a train-mode BatchNorm (+ ReLU) on bn_x and the partial
    rows are that BatchNorm's BACKWARD sums {sum g, sum g xhat} (bn_bwd's part / part_rows); a launch split over K returns
    0 and writes none (the caller's bn_bwd then runs its own pass)."""
    _dev(x, w, out, zeros, scratch, colstats)
    assert x.is_contiguous() and w.is_contiguous() and out.is_contiguous() and x.dtype == w.dtype == out.dtype
    assert x.shape[0] == B * H * W and tuple(out.shape) == (B * H * W, w.shape[0]) and zeros.numel() * zeros.element_size() >= 16
    nsc = 0 if scratch is None else scratch.numel()
    prow = 0
    if colstats is not None:
        prow = L.load().ffm_conv3x3_colstat_rows(B, H, W, x.shape[1], w.shape[0], w.shape[1], nsc, L.dtype_code(x.dtype))
        assert prow >= 0 and _f32(colstats).numel() >= 2 * w.shape[0] * prow
    if bnbwd is not None and prow > 0:
        bn_x, bn_mask, bn_mean, bn_rstd = bnbwd
        _dev(bn_x, bn_mask, bn_mean, bn_rstd)
        assert bn_x.dtype == out.dtype and bn_x.shape == out.shape and bn_x.is_contiguous()
        assert bn_mask is None or (bn_mask.dtype == out.dtype and bn_mask.shape == out.shape and bn_mask.is_contiguous())
        assert _f32(bn_mean).numel() == w.shape[0] and _f32(bn_rstd).numel() == w.shape[0]
        _call("ffm_conv3x3_nhwc_bnbwd", L.ptr(x), L.ptr(w), L.ptr(out), B, H, W, x.shape[1], w.shape[0], w.shape[1], L.ptr(zeros),
              L.ptr(_f32(scratch)), nsc, L.ptr(colstats), L.ptr(bn_x), L.ptr(bn_mask), L.ptr(bn_mean), L.ptr(bn_rstd),
              L.dtype_code(x.dtype), L.stream_ptr())
        return prow
    _call("ffm_conv3x3_nhwc", L.ptr(x), L.ptr(w), L.ptr(out), B, H, W, x.shape[1], w.shape[0], w.shape[1], L.ptr(zeros),
          L.ptr(_f32(scratch)), nsc, L.ptr(colstats) if (prow > 0 and bnbwd is None) else None, L.dtype_code(x.dtype), L.stream_ptr())
    return prow if bnbwd is None else 0


def bn_blocks(rows: int) -> int:
    return L.load().ffm_bn_blocks(rows)


def bn_fwd(x: Tensor, gamma: Tensor, beta: Tensor, run_mean: Tensor, run_var: Tensor, mean: Tensor, rstd: Tensor,
           part: Optional[Tensor], y: Tensor, training: bool, relu: bool, res: Optional[Tensor] = None,
           part_rows: int = 0) -> None:
    """part_rows > 0: `part` already holds that many rows of column sums from the producer of x (gemm_nt / conv3x3
    colstats): the column-sum pass over x is skipped."""
    _dev(x, gamma, beta, run_mean, run_var, mean, rstd, part, y, res)
    rows, Cc = x.shape
    assert x.is_contiguous() and y.is_contiguous() and x.dtype == y.dtype and (res is None or res.dtype == x.dtype)
    _call("ffm_bn_fwd", L.ptr(x), L.ptr(_f32(gamma)), L.ptr(_f32(beta)), L.ptr(_f32(run_mean)), L.ptr(_f32(run_var)),
          L.ptr(_f32(mean)), L.ptr(_f32(rstd)), L.ptr(_f32(part)), part_rows if training else 0, L.ptr(res), L.ptr(y), rows, Cc,
          int(training), int(relu), L.dtype_code(x.dtype), L.stream_ptr())


def bn_bwd(dy: Tensor, relu_out: Optional[Tensor], x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, part: Tensor,
           k12: Tensor, dgamma: Tensor, dbeta: Tensor, dx: Tensor, g_out: Optional[Tensor] = None, part_rows: int = 0) -> None:
    """g_out (optional, like dy): also receives g = dy * (relu_out > 0), the identity path's gradient.
    part_rows > 0: `part` already holds that many rows of {sum g, sum g xhat} from the producer of dy (gemm_nt's `bnbwd`):
    the column-sum pass over dy / x / relu_out is skipped."""
    _dev(dy, relu_out, x, gamma, mean, rstd, part, k12, dgamma, dbeta, dx, g_out)
    rows, Cc = x.shape
    assert dy.is_contiguous() and x.is_contiguous() and dx.is_contiguous() and dy.dtype == x.dtype == dx.dtype
    assert g_out is None or (g_out.is_contiguous() and g_out.dtype == dy.dtype and g_out.shape == dy.shape)
    assert part_rows == 0 or (g_out is None and part.numel() >= part_rows * 2 * Cc)
    _call("ffm_bn_bwd", L.ptr(dy), L.ptr(relu_out), L.ptr(x), L.ptr(_f32(gamma)), L.ptr(_f32(mean)), L.ptr(_f32(rstd)),
          L.ptr(_f32(part)), part_rows, L.ptr(_f32(k12)), L.ptr(_f32(dgamma)), L.ptr(_f32(dbeta)), L.ptr(dx), L.ptr(g_out),
          rows, Cc, L.dtype_code(x.dtype), L.stream_ptr())


def avgpool2(inp: Tensor, out: Tensor, B: int, H: int, W: int, backward: bool = False) -> None:
    """forward: inp [B*H*W, C] -> out [B*(H/2)*(W/2), C]; backward: inp = d(pooled), out = d(x) [B*H*W, C]."""
    _dev(inp, out)
    assert inp.is_contiguous() and out.is_contiguous() and inp.dtype == out.dtype
    _call("ffm_avgpool2", L.ptr(inp), L.ptr(out), B, H, W, inp.shape[1], int(backward), L.dtype_code(inp.dtype),
          L.stream_ptr())


def add(a: Tensor, b: Tensor, out: Tensor) -> None:
    _dev(a, b, out)
    assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous() and a.dtype == b.dtype == out.dtype
    _call("ffm_add", L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), L.dtype_code(a.dtype), L.stream_ptr())


def relu_bwd(g: Tensor, y: Tensor, out: Tensor) -> None:
    _dev(g, y, out)
    assert g.is_contiguous() and y.is_contiguous() and out.is_contiguous() and g.dtype == y.dtype == out.dtype
    _call("ffm_relu_bwd", L.ptr(g), L.ptr(y), L.ptr(out), g.numel(), L.dtype_code(g.dtype), L.stream_ptr())


def attnpool_tokens(inp: Tensor, pos: Optional[Tensor], out: Tensor, B: int, HW: int, backward: bool = False) -> None:
    _dev(inp, pos, out)
    assert inp.is_contiguous() and out.is_contiguous() and inp.dtype == out.dtype
    _call("ffm_attnpool_tokens", L.ptr(inp), L.ptr(pos), L.ptr(out), B, HW, inp.shape[1], int(backward),
          L.dtype_code(inp.dtype), L.stream_ptr())


OT_MODES = {"Sinkhorn": 1, "COT": 2}


def ot_head_fwd(f: Tensor, tn: Tensor, logit_scale: Tensor, rnorm: Tensor, sim: Tensor, T: Tensor, errs: Tensor,
                istop: Tensor, tsum: Tensor, logits_img: Tensor, B: int, Ltok: int, n_cls: int, N: int, mode: str,
                eps: float, thresh: float, max_iter: int, top_percent: float) -> None:
    _dev(f, tn, logit_scale, rnorm, sim, T, errs, istop, tsum, logits_img)
    assert f.is_contiguous() and tn.is_contiguous() and istop.dtype == torch.int32
    _call("ffm_ot_head_fwd", L.ptr(f), L.ptr(_f32(tn)), L.ptr(_f32(logit_scale)), L.ptr(_f32(rnorm)), L.ptr(_f32(sim)),
          L.ptr(_f32(T)), L.ptr(_f32(errs)), L.ptr(istop), L.ptr(_f32(tsum)), L.ptr(_f32(logits_img)), B, Ltok, f.shape[1],
          n_cls, N, OT_MODES[mode], eps, thresh, max_iter, top_percent, L.dtype_code(f.dtype), L.stream_ptr())


def ot_head_bwd(f: Tensor, tn: Tensor, logit_scale: Tensor, rnorm: Tensor, T: Tensor, dlogits_img: Tensor, df: Tensor,
                dtn_part: Tensor, B: int, Ltok: int, n_cls: int, N: int) -> None:
    _dev(f, tn, logit_scale, rnorm, T, dlogits_img, df, dtn_part)
    assert f.is_contiguous() and df.is_contiguous() and f.dtype == df.dtype
    _call("ffm_ot_head_bwd", L.ptr(f), L.ptr(_f32(tn)), L.ptr(_f32(logit_scale)), L.ptr(_f32(rnorm)), L.ptr(_f32(T)),
          L.ptr(_f32(dlogits_img)), L.ptr(df), L.ptr(_f32(dtn_part)), B, Ltok, f.shape[1], n_cls, N, L.dtype_code(f.dtype),
          L.stream_ptr())


def expand_u8(src: Tensor, dst: Tensor, rep: int) -> Tensor:
    """uint8 [B, C1, H, W] -> fp32 [B, C1*rep, H, W] with each channel repeated rep times (ffm_expand_u8)."""
    _dev(src, dst)
    B, C1, H, W = src.shape
    assert src.dtype == torch.uint8 and src.is_contiguous() and dst.dtype == torch.float32 and dst.is_contiguous()
    assert tuple(dst.shape) == (B, C1 * rep, H, W)
    _call("ffm_expand_u8", L.ptr(src), L.ptr(dst), B, C1, H * W, rep, L.stream_ptr())
    return dst


EVAL_SLOTS = 10


EVAL_SORT_FROM = 32768      # samples from which the O(N log N) evaluator replaces the all-pairs kernel


def eval_counts(prob: Tensor, label: Tensor, attr: Optional[Tensor], num_groups: int, method: str = "auto",
                out: Optional[Tensor] = None) -> Tensor:
    """Integer counts behind every score of the binary-task evaluator: int64 [(G + 2), 10] on the device (rows: groups
    0..G-1, unknown, all).  method: 'pairs' (ffm_eval_counts, all pairs), 'sort' (ffm_eval_counts_sorted, for large test
    sets), 'auto' (by N); the two give identical integers."""
    _dev(prob, label, attr)
    N = prob.shape[0]
    assert prob.dtype == torch.float32 and prob.dim() == 2 and prob.shape[1] == 2 and prob.is_contiguous()
    assert label.dtype == torch.int64 and label.is_contiguous() and label.numel() == N
    assert attr is None or (attr.dtype == torch.int64 and attr.is_contiguous() and attr.numel() == N)
    assert method in ("auto", "pairs", "sort")
    if out is None:
        out = torch.empty(num_groups + 2, EVAL_SLOTS, device=prob.device, dtype=torch.int64)
    assert out.dtype == torch.int64 and out.is_contiguous() and tuple(out.shape) == (num_groups + 2, EVAL_SLOTS) and out.is_cuda
    if method == "sort" or (method == "auto" and N >= EVAL_SORT_FROM):
        if recording():
            # refused BEFORE anything is allocated or launched: a recorded launch would keep the raw pointer of the
            # temporary workspace and replay into freed memory
            raise RuntimeError("eval_counts(method='sort') holds a temporary workspace and cannot be recorded")
        nbytes = int(L.load().ffm_eval_counts_ws_bytes(N))
        if nbytes <= 0:
            raise RuntimeError("ffm_eval_counts_ws_bytes failed")
        ws = torch.empty(nbytes + 256, device=prob.device, dtype=torch.uint8)      # device memory is PyTorch's job
        off = (-ws.data_ptr()) % 256
        _call("ffm_eval_counts_sorted", L.ptr(prob), L.ptr(label), L.ptr(attr), N, num_groups, L.ptr(out),
              ws.data_ptr() + off, nbytes, L.stream_ptr())
        return out
    _call("ffm_eval_counts", L.ptr(prob), L.ptr(label), L.ptr(attr), N, num_groups, L.ptr(out), L.stream_ptr())
    return out


def attention_fwd(qkv: Tensor, out: Tensor, lse: Optional[Tensor], B: int, Ltok: int, heads: int,
                  causal: bool = False) -> Tensor:
    _dev(qkv, out, lse)
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.dtype == out.dtype
    assert qkv.shape[1] == 3 * heads * 64 and out.shape[1] == heads * 64
    _call("ffm_attention_fwd", L.ptr(qkv), L.ptr(out), L.ptr(_f32(lse)), B, Ltok, heads, int(causal),
                                       L.dtype_code(qkv.dtype), L.stream_ptr())
    return out


def attention_bwd_lnstat_ok(Ltok: int, causal: bool, dtype: torch.dtype) -> bool:
    """Whether attention_bwd(ln_stat=...) is served for this shape (ffm_attention_bwd_lnstat_ok)."""
    return L.is16(dtype) and bool(L.load().ffm_attention_bwd_lnstat_ok(Ltok, int(causal), L.dtype_code(dtype)))


def attention_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, delta: Tensor, dqkv: Tensor, B: int,
                  Ltok: int, heads: int, causal: bool = False, ln_stat: Optional[tuple] = None) -> Tensor:
    """ln_stat = (wg [3E], d [3E], part [2 heads, B Ltok, 2]): also leave ln_1's backward row sums (ffm_attention_bwd_lnstat)."""
    _dev(qkv, out, dout, lse, delta, dqkv)
    for t in (qkv, out, dout, dqkv):
        assert t.is_contiguous() and t.dtype == qkv.dtype
    if ln_stat is not None:
        wg, d, part = ln_stat
        _dev(wg, d, part)
        E3 = 3 * heads * 64
        assert _f32(wg).numel() == E3 and _f32(d).numel() == E3 and _f32(part).numel() >= 2 * heads * B * Ltok * 2
        _call("ffm_attention_bwd_lnstat", L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(_f32(lse)), L.ptr(_f32(delta)), L.ptr(dqkv),
              L.ptr(wg), L.ptr(d), L.ptr(part), B, Ltok, heads, int(causal), L.dtype_code(qkv.dtype), L.stream_ptr())
        return dqkv
    _call("ffm_attention_bwd", L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(_f32(lse)), L.ptr(_f32(delta)),
                                       L.ptr(dqkv), B, Ltok, heads, int(causal), L.dtype_code(qkv.dtype),
                                       L.stream_ptr())
    return dqkv


def lora_down_blocks(M: int, K: int, r: int, dtype: torch.dtype) -> int:
    """dS partial rows ffm_lora_down writes for exactly this (M, K, r, dtype): the count to reduce over."""
    return L.load().ffm_lora_down_blocks(M, K, r, L.dtype_code(dtype))


def lora_down_blocks_max(M_max: int, K: int, r: int, dtype: torch.dtype) -> int:
    """Upper bound of lora_down_blocks over every M <= M_max: for sizing ds_part buffers only."""
    return L.load().ffm_lora_down_blocks_max(M_max, K, r, L.dtype_code(dtype))


class ReducePlan:
    """Device-resident descriptor table for ffm_reduce_partials_multi."""

    def __init__(self, entries, device):
        # entries: (part tensor, nsplit, n, out tensor, transpose_K, transpose_r)
        import ctypes as C
        arr = (L.ReduceDesc * len(entries))()
        self.keep = entries
        self.max_n = 0
        for i, (part, nsplit, n, out, tK, tr) in enumerate(entries):
            _dev(part, out)
            arr[i] = L.ReduceDesc(part.data_ptr(), out.data_ptr(), nsplit, n, tK, tr)
            self.max_n = max(self.max_n, n * (8 if nsplit > 64 else 1))   # 32 lanes per output beyond 64 partial rows
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.n = len(entries)

    def run(self) -> None:
        _call("ffm_reduce_partials_multi", self.table.data_ptr(), self.n, self.max_n, L.stream_ptr())


class PackPlan:
    """Device-resident descriptor table for ffm_lora_pack_multi (all adapters in one launch)."""

    def __init__(self, entries, dtype, device):
        # entries: (src fp32 tensor [K,r] or [r,K], layout_rk, dst [16,K] dtype[, wide [K,32] dtype[, ln]]);
        # ln = (gamma [K], beta [K], ln_rk [2,16]): a LayerNorm folded into the product dst rides in (layout_rk False):
        # dst is gamma-scaled and ln_rk receives its corrections (ffm_lora_pack_ln); optional 6th element (row14 [K], row15 [K]):
        # two caller vectors as rows 14 / 15 of dst (r <= 14; ffm_pack_desc.row14 / row15)
        arr = (L.PackDesc * len(entries))()
        self.keep, self.max_K, self.dtype = entries, 0, dtype
        self.any_ln = False
        for i, ent in enumerate(entries):
            src, layout_rk, dst = ent[:3]
            wide = ent[3] if len(ent) > 3 else None
            ln = ent[4] if len(ent) > 4 else None
            _dev(src, dst, wide)
            K = dst.shape[1]
            r = src.shape[0] if layout_rk else src.shape[1]
            assert dst.dtype == dtype and dst.shape[0] == 16 and dst.is_contiguous() and src.is_contiguous()
            assert wide is None or (wide.dtype == dtype and tuple(wide.shape) == (K, 32) and wide.is_contiguous())
            lnp = (None, None, None)
            if ln is not None:
                assert not layout_rk and _f32(ln[0]).numel() == K and _f32(ln[1]).numel() == K and _f32(ln[2]).numel() == 32
                _dev(*ln)
                lnp = tuple(t.data_ptr() for t in ln)
                self.any_ln = True
            rows = ent[5] if len(ent) > 5 and ent[5] is not None else (None, None)
            if rows[0] is not None:
                _dev(*rows)
                assert r <= 14 and _f32(rows[0]).numel() == K and _f32(rows[1]).numel() == K
            arr[i] = L.PackDesc(src.data_ptr(), dst.data_ptr(), K, r, int(layout_rk), 0, L.ptr(wide), *lnp, L.ptr(rows[0]), L.ptr(rows[1]))
            self.max_K = max(self.max_K, K)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        self.n = len(entries)

    def run(self) -> None:
        _call("ffm_lora_pack_multi", self.table.data_ptr(), self.n, self.max_K, L.dtype_code(self.dtype),
                                             L.stream_ptr())
        if self.any_ln:
            _call("ffm_lora_pack_ln", self.table.data_ptr(), self.n, L.dtype_code(self.dtype), L.stream_ptr())


def lora_grad_partial_ln(x: Tensor, v: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, beta: Tensor, r: int,
                         part: Tensor) -> None:
    """lora_grad_partial for x = LayerNorm(x_raw) that was folded into its consumer: x holds the RAW rows."""
    _dev(x, v, mean, rstd, gamma, beta, part)
    M, K = x.shape
    assert L.is16(x.dtype) and K % 128 == 0 and r <= 16
    _call("ffm_lora_grad_partial_ln", L.ptr(x), _ld(x), L.ptr(_f32(v)), L.ptr(_f32(mean)), L.ptr(_f32(rstd)),
          L.ptr(_f32(gamma)), L.ptr(_f32(beta)), M, K, r, L.ptr(_f32(part)), L.dtype_code(x.dtype), L.stream_ptr())


def lora_grad_splits(M: int) -> int:
    return L.load().ffm_lora_grad_splits(M)


def lora_down(x: Tensor, P: Tensor, layout_rk: bool, S: Tensor, attr: Optional[Tensor], r: int, G: int,
              rows_per_sample: int, scaling: float, lambda_group: float, t: Optional[Tensor], ts: Optional[Tensor],
              t_fwd: Optional[Tensor] = None, ds_part: Optional[Tensor] = None) -> None:
    _dev(x, P, S, attr, t, ts, t_fwd, ds_part)
    M, K = x.shape
    if attr is not None:
        assert attr.dtype == torch.int32
    _call("ffm_lora_down", L.ptr(x), _ld(x), L.ptr(_f32(P)), int(layout_rk), L.ptr(_f32(S)), L.ptr(attr), M,
                                   K, r, G, rows_per_sample, scaling, lambda_group, L.ptr(_f32(t)), L.ptr(_f32(ts)),
                                   L.ptr(_f32(t_fwd)), L.ptr(_f32(ds_part)), L.dtype_code(x.dtype), L.stream_ptr())


def lora_grad_partial(x: Tensor, v: Tensor, r: int, part: Tensor) -> None:
    _dev(x, v, part)
    M, K = x.shape
    _call("ffm_lora_grad_partial", L.ptr(x), _ld(x), L.ptr(_f32(v)), M, K, r, L.ptr(_f32(part)),
                                           L.dtype_code(x.dtype), L.stream_ptr())


def reduce_partials(part: Tensor, nsplit: int, n: int, out: Tensor, transpose_K: int = 0, transpose_r: int = 0,
                    accumulate: bool = False) -> None:
    _dev(part, out)
    _call("ffm_reduce_partials", L.ptr(_f32(part)), nsplit, n, L.ptr(out), transpose_K, transpose_r,
                                         int(accumulate), L.stream_ptr())


def head_fwd(f: Tensor, tbar: Tensor, logit_scale: Tensor, fbar: Tensor, rnorm: Tensor, logits_img: Tensor, B: int,
             Ltok: int, n_cls: int) -> None:
    _dev(f, tbar, logit_scale, fbar, rnorm, logits_img)
    D = f.shape[1]
    assert f.is_contiguous()
    _call("ffm_head_fwd", L.ptr(f), L.ptr(_f32(tbar)), L.ptr(_f32(logit_scale)), L.ptr(_f32(fbar)),
                                  L.ptr(_f32(rnorm)), L.ptr(_f32(logits_img)), B, Ltok, D, n_cls,
                                  L.dtype_code(f.dtype), L.stream_ptr())


def text_embed(prefix: Tensor, ctx: Tensor, suffix: Tensor, pos: Tensor, x: Tensor, n_cls: int, TL: int) -> None:
    """prompts = [prefix, ctx, suffix] + positional embedding -> text tower input x [n_text * TL, w] (ffm_text_embed)."""
    _dev(prefix, ctx, suffix, pos, x)
    n_prompts, n_ctx, w = ctx.shape
    for t in (prefix, ctx, suffix, pos, x):
        assert t.dtype == torch.float32 and t.is_contiguous()
    assert prefix.shape[0] == suffix.shape[0] == n_prompts * n_cls and x.shape[0] >= n_prompts * n_cls * TL and pos.shape[0] >= TL
    _call("ffm_text_embed", L.ptr(prefix), L.ptr(ctx), L.ptr(suffix), suffix.shape[1], L.ptr(pos), L.ptr(x), n_prompts, n_cls,
          n_ctx, TL, w, L.stream_ptr())


def text_tail_fwd(x: Tensor, eot_row: Tensor, lnw: Tensor, lnb: Tensor, proj: Tensor, tf: Tensor, tn: Tensor, rnorm: Tensor,
                  stats: Tensor, tbar: Optional[Tensor], n_prompts: int, n_cls: int) -> None:
    """EOT gather -> ln_final -> text_projection -> normalise (-> mean over the prompts into tbar) (ffm_text_tail_fwd)."""
    _dev(x, eot_row, lnw, lnb, proj, tf, tn, rnorm, stats, tbar)
    assert eot_row.dtype == torch.int32 and x.dtype == torch.float32 and proj.dtype == torch.float32 and proj.is_contiguous()
    w, D = proj.shape
    assert x.shape[1] == w and tf.shape == tn.shape == (n_prompts * n_cls, D) and tf.is_contiguous() and tn.is_contiguous()
    _call("ffm_text_tail_fwd", L.ptr(x), L.ptr(eot_row), L.ptr(lnw), L.ptr(lnb), L.ptr(proj), L.ptr(tf), L.ptr(tn), L.ptr(rnorm),
          L.ptr(stats), L.ptr(tbar), n_prompts, n_cls, w, D, L.stream_ptr())


def text_tail_bwd(x: Tensor, eot_row: Tensor, lnw: Tensor, proj: Tensor, tn: Tensor, rnorm: Tensor, stats: Tensor,
                  dtbar: Optional[Tensor], dtn: Optional[Tensor], dy: Tensor, g: Tensor, n_prompts: int, n_cls: int, TL: int) -> None:
    """Gradient of the head's text operand -> gradient of the tower's output rows g [n_text * TL, w] (ffm_text_tail_bwd)."""
    _dev(x, eot_row, lnw, proj, tn, rnorm, stats, dtbar, dtn, dy, g)
    w, D = proj.shape
    assert g.dtype == torch.float32 and g.is_contiguous() and g.shape[0] >= n_prompts * n_cls * TL and g.shape[1] == w
    _call("ffm_text_tail_bwd", L.ptr(x), L.ptr(eot_row), L.ptr(lnw), L.ptr(proj), L.ptr(tn), L.ptr(rnorm), L.ptr(stats),
          L.ptr(dtbar), L.ptr(dtn), L.ptr(dy), L.ptr(g), n_prompts, n_cls, TL, w, D, L.stream_ptr())


def text_ctx_grad(g: Tensor, dctx: Tensor, n_cls: int, TL: int) -> None:
    """d ctx [n_prompts, n_ctx, w] = the tower input gradient's ctx rows summed over the classes (ffm_text_ctx_grad)."""
    _dev(g, dctx)
    n_prompts, n_ctx, w = dctx.shape
    assert g.dtype == dctx.dtype == torch.float32 and g.is_contiguous() and dctx.is_contiguous()
    _call("ffm_text_ctx_grad", L.ptr(g), L.ptr(dctx), n_prompts, n_cls, n_ctx, TL, w, L.stream_ptr())


def ce_loss(logits_img: Tensor, label: Tensor, logits: Tensor, prob: Tensor, loss: Tensor, dlogits_img: Tensor,
            finite: Optional[Tensor], nb: int, S: int, n_cls: int) -> None:
    _dev(logits_img, label, logits, prob, loss, dlogits_img, finite)
    assert label.dtype == torch.int64 and (finite is None or finite.dtype == torch.int32)
    _call("ffm_ce_loss", L.ptr(_f32(logits_img)), L.ptr(label), L.ptr(_f32(logits)), L.ptr(_f32(prob)),
                                 L.ptr(_f32(loss)), L.ptr(_f32(dlogits_img)), L.ptr(finite), nb, S, n_cls,
                                 L.stream_ptr())


def head_bwd(f: Tensor, tbar: Tensor, logit_scale: Tensor, fbar: Tensor, rnorm: Tensor, dlogits_img: Tensor,
             df: Tensor, dtbar: Tensor, B: int, Ltok: int, n_cls: int) -> None:
    _dev(f, tbar, logit_scale, fbar, rnorm, dlogits_img, df, dtbar)
    D = f.shape[1]
    assert f.dtype == df.dtype and df.is_contiguous()
    _call("ffm_head_bwd", L.ptr(f), L.ptr(_f32(tbar)), L.ptr(_f32(logit_scale)), L.ptr(_f32(fbar)),
                                  L.ptr(_f32(rnorm)), L.ptr(_f32(dlogits_img)), L.ptr(df), L.ptr(_f32(dtbar)), B,
                                  Ltok, D, n_cls, L.dtype_code(f.dtype), L.stream_ptr())


def sgd_momentum(p: Tensor, g: Tensor, buf: Tensor, lr: float, momentum: float, weight_decay: float,
                 first_step: bool, repeats: int = 1) -> None:
    """torch.optim.SGD.step(), `repeats` times on the same gradient (the reference's shared optimizer is stepped once
    per registered model name: Dassl/dassl/engine/trainer.py:333-337)."""
    _dev(p, g, buf)
    if repeats == 1:
        _call("ffm_sgd_momentum", L.ptr(_f32(p)), L.ptr(_f32(g)), L.ptr(_f32(buf)), p.numel(), lr, momentum,
              weight_decay, int(first_step), L.stream_ptr())
    else:
        _call("ffm_sgd_momentum_n", L.ptr(_f32(p)), L.ptr(_f32(g)), L.ptr(_f32(buf)), p.numel(), lr, momentum,
              weight_decay, int(first_step), int(repeats), L.stream_ptr())


SCALE_STATE = 8     # floats of the fp16 gradient-scale state (include/ffm_hip.h: ffm_loss_scale)


def loss_scale(p: Tensor, state: Tensor) -> None:
    """p *= state[0] (the device-resident gradient scale); marks the step as good (ffm_loss_scale)."""
    _dev(p, state)
    assert p.is_contiguous() and state.numel() == SCALE_STATE
    _call("ffm_loss_scale", L.ptr(_f32(p)), p.numel(), L.ptr(_f32(state)), L.stream_ptr())


def unscale_check(g: Tensor, state: Tensor) -> None:
    """g *= 1 / scale; a non-finite gradient marks the step as overflowed (ffm_unscale_check)."""
    _dev(g, state)
    assert g.is_contiguous() and state.numel() == SCALE_STATE
    _call("ffm_unscale_check", L.ptr(_f32(g)), g.numel(), L.ptr(_f32(state)), L.stream_ptr())


def sgd_momentum_gated(p: Tensor, g: Tensor, buf: Tensor, lr: float, momentum: float, weight_decay: float,
                       first_step: bool, repeats: int, state: Tensor) -> None:
    """sgd_momentum unless the step overflowed (then nothing moves); afterwards the scale backs off / grows."""
    _dev(p, g, buf, state)
    _call("ffm_sgd_momentum_gated", L.ptr(_f32(p)), L.ptr(_f32(g)), L.ptr(_f32(buf)), p.numel(), lr, momentum,
          weight_decay, int(first_step), int(repeats), L.ptr(_f32(state)), L.stream_ptr())


def sgd_momentum_dev(p: Tensor, g: Tensor, buf: Tensor, hp: Tensor) -> None:
    _dev(p, g, buf, hp)
    _call("ffm_sgd_momentum_dev", L.ptr(_f32(p)), L.ptr(_f32(g)), L.ptr(_f32(buf)), p.numel(), L.ptr(_f32(hp)),
                                          L.stream_ptr())


def scale_by(p: Tensor, w: Tensor, out: Tensor) -> None:
    _dev(p, w, out)
    _call("ffm_scale_by", L.ptr(_f32(p)), L.ptr(_f32(w)), L.ptr(_f32(out)), p.numel(), L.stream_ptr())


def scale_acc(p: Tensor, w: Tensor, acc: Tensor) -> None:
    """acc += p * w, product and sum rounded separately (ffm_scale_acc)."""
    _dev(p, w, acc)
    _call("ffm_scale_acc", L.ptr(_f32(p)), L.ptr(_f32(w)), L.ptr(_f32(acc)), p.numel(), L.stream_ptr())


def scale_check(p: Tensor, scale: float, finite: Optional[Tensor] = None) -> None:
    """p *= scale in place; clears the int32 flag when a product is not finite (ffm_scale_check)."""
    _dev(p, finite)
    assert p.is_contiguous() and (finite is None or finite.dtype == torch.int32)
    _call("ffm_scale_check", L.ptr(_f32(p)), float(scale), p.numel(), L.ptr(finite), L.stream_ptr())


def fedavg_finish(avg: Tensor, prev: Tensor, out: Tensor, s_offsets: Optional[Tensor], G: int, r: int,
                  shared_half_s: bool, beta: float) -> None:
    _dev(avg, prev, out, s_offsets)
    n_s = 0 if s_offsets is None else s_offsets.numel()
    if s_offsets is not None:
        assert s_offsets.dtype == torch.int64
    _call("ffm_fedavg_finish", L.ptr(_f32(avg)), L.ptr(_f32(prev)), L.ptr(_f32(out)), avg.numel(),
                                       L.ptr(s_offsets), n_s, G, r, int(shared_half_s), beta, L.stream_ptr())


def cast_from_f32(src: Tensor, dtype: torch.dtype) -> Tensor:
    _dev(src)
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    _call("ffm_cast_f32_to", L.ptr(_f32(src.contiguous())), L.ptr(dst), src.numel(), L.dtype_code(dtype),
                                     L.stream_ptr())
    return dst


def cast_to_f32(src: Tensor) -> Tensor:
    _dev(src)
    dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    _call("ffm_cast_to_f32", L.ptr(src.contiguous()), L.ptr(dst), src.numel(), L.dtype_code(src.dtype),
                                     L.stream_ptr())
    return dst


def transpose_cast(src: Tensor, dtype: torch.dtype) -> Tensor:
    """[rows, cols] fp32 -> [cols, rows] dtype."""
    _dev(src)
    rows, cols = src.shape
    dst = torch.empty((cols, rows), dtype=dtype, device=src.device)
    _call("ffm_transpose_cast", L.ptr(_f32(src.contiguous())), L.ptr(dst), rows, cols, L.dtype_code(dtype),
                                        L.stream_ptr())
    return dst
