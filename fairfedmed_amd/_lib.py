"""ctypes binding of libffm_hip.so (include/ffm_hip.h).

The product path has NO fallback: if the library is missing, or a call returns
a non-zero status, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FFM_LIB_PATH: a diagnostic build of the same library (tools/panel_stamps.py); there is still no fallback
LIB_PATH = os.environ.get("FFM_LIB_PATH") or os.path.join(_HERE, "csrc", "libffm_hip.so")

F32, BF16, F32_X3, F16 = 0, 1, 2, 3      # FFM_F16: IEEE half storage (the reference's PREC="fp16")
F32_X3_W16 = 4                           # FFM_F32_X3 with the weight operand stored as IEEE half
EPI_BIAS, EPI_LORA, EPI_LORA_KR, EPI_RESIDUAL, EPI_GELU, EPI_DGELU, EPI_RANKOP = 1, 2, 4, 8, 16, 32, 64
EPI_ROWSTATS, EPI_LNIN, EPI_LGRAD, EPI_BNBWD = 128, 256, 512, 1024
EPI_LNB_STAT, EPI_LNB_APPLY = 2048, 4096
ABI_VERSION = 12

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", _vp), ("b", _vp), ("c", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("lda", _i32), ("ldb", _i32), ("ldc", _i32),
        ("flags", _i32), ("rank", _i32),
        ("bias", _vp), ("ts", _vp), ("lw", _vp), ("res", _vp), ("c2", _vp), ("aux", _vp),
        ("rk", _vp), ("S", _vp), ("attr", _vp), ("t_out", _vp), ("ts_out", _vp), ("t_fwd", _vp), ("ds_part", _vp),
        ("G", _i32), ("rows_per_sample", _i32), ("scaling", _f32), ("lambda_group", _f32),
        ("b_packed", _vp), ("lw_wide", _vp),
        ("rowstat_part", _vp), ("ln_part", _vp), ("ln_c", _vp), ("ln_mean", _vp), ("ln_rstd", _vp), ("ln_np", _i32), ("gelu_deriv", _i32),
        ("ln_rk", _vp), ("colstat_part", _vp),
        ("lg_v", _vp), ("lg_part_c", _vp), ("lg_part_a", _vp),
        ("bn_x", _vp), ("bn_mask", _vp), ("bn_mean", _vp), ("bn_rstd", _vp), ("bn_gout", _vp),
        ("sk_part", _vp),
        ("lnb_wg", _vp), ("lnb_d", _vp), ("lnb_part", _vp), ("lnb_np", _i32), ("lnb_pad_", _i32), ("lnb_x", _vp), ("lnb_gamma", _vp),
    ]


class PackDesc(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("K", _i32), ("r", _i32), ("layout_rk", _i32), ("pad_", _i32),
                ("dst_wide", _vp), ("gamma", _vp), ("beta", _vp), ("ln_rk", _vp), ("row14", _vp), ("row15", _vp)]


class ReduceDesc(C.Structure):
    _fields_ = [("part", _vp), ("out", _vp), ("nsplit", _i32), ("n", _i32), ("transpose_K", _i32),
                ("transpose_r", _i32)]


# name -> argtypes (restype is always int, except the two helpers)
SIGNATURES = {
    "ffm_abi_version": [],
    "ffm_gemm_nt": [C.POINTER(GemmArgs), _i32, _vp],
    "ffm_gemm_splitk_floats": [_i32, _i32, _i32, _i32],
    "ffm_gemm_tiles_m": [_i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "ffm_gemm_tiles_n": [_i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "ffm_gemm_lgrad_rows": [_i32, _i32, _i32, _i32, _i32, _i32, _i32],
    "ffm_gemm_tile_shape": [_i32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)],
    "ffm_pack_b": [_vp, _vp, _i32, _i32, _i32, _vp],
    "ffm_lora_pack_multi": [_vp, _i32, _i32, _i32, _vp],
    "ffm_lora_pack_ln": [_vp, _i32, _i32, _vp],
    "ffm_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "ffm_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "ffm_patchify": [_vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _i32, _i32, _vp],
    "ffm_embed_lnpre": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "ffm_slice_blocks": [_i32, _i32],
    "ffm_slice_wgrad_blocks": [_i32, _i32],
    "ffm_slice_bwd_ab_blocks": [],
    "ffm_slice_conv_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "ffm_patchify_minmax": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _i32, _vp],
    "ffm_embed_lnpre_bwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "ffm_slice_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_f32),
                      _i32, _vp],
    "ffm_stem_im2col": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _i32, _vp],
    "ffm_im2col3x3": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_col2im3x3": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_bn_blocks": [_i32],
    "ffm_bn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_bn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "ffm_avgpool2": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_add": [_vp, _vp, _vp, C.c_int64, _i32, _vp],
    "ffm_relu_bwd": [_vp, _vp, _vp, C.c_int64, _i32, _vp],
    "ffm_attnpool_tokens": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_conv3x3_nhwc": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _i32, _vp],
    "ffm_conv3x3_nhwc_bnbwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp],
    "ffm_conv3x3_colstat_rows": [_i32, _i32, _i32, _i32, _i32, _i32, _i64, _i32],
    "ffm_eval_counts": [_vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "ffm_eval_counts_ws_bytes": [_i32],
    "ffm_eval_counts_sorted": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp],
    "ffm_ot_head_fwd": [_vp] * 10 + [_i32] * 6 + [_f32, _f32, _i32, _f32, _i32, _vp],
    "ffm_ot_head_bwd": [_vp] * 8 + [_i32] * 6 + [_vp],
    "ffm_expand_u8": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "ffm_scale_check": [_vp, _f32, _i64, _vp, _vp],
    "ffm_attention_fwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_attention_bwd_lnstat_ok": [_i32, _i32, _i32],
    "ffm_attention_bwd_lnstat": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_lora_down": [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp,
                      _i32, _vp],
    "ffm_lora_down_blocks": [_i32, _i32, _i32, _i32],
    "ffm_lora_down_blocks_max": [_i32, _i32, _i32, _i32],
    "ffm_lora_grad_partial": [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "ffm_lora_grad_partial_ln": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "ffm_lora_grad_splits": [_i32],
    "ffm_reduce_partials": [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp],
    "ffm_reduce_partials_multi": [_vp, _i32, _i32, _vp],
    "ffm_head_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_ce_loss": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "ffm_head_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_text_embed": [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_text_tail_fwd": [_vp] * 10 + [_i32] * 4 + [_vp],
    "ffm_text_tail_bwd": [_vp] * 11 + [_i32] * 5 + [_vp],
    "ffm_text_ctx_grad": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "ffm_sgd_momentum": [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _vp],
    "ffm_sgd_momentum_n": [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _i32, _vp],
    "ffm_sgd_momentum_dev": [_vp, _vp, _vp, _i64, _vp, _vp],
    "ffm_loss_scale": [_vp, _i64, _vp, _vp],
    "ffm_unscale_check": [_vp, _i64, _vp, _vp],
    "ffm_sgd_momentum_gated": [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _i32, _vp, _vp],
    "ffm_scale_by": [_vp, _vp, _vp, _i64, _vp],
    "ffm_scale_acc": [_vp, _vp, _vp, _i64, _vp],
    "ffm_fedavg_finish": [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _vp],
    "ffm_cast_f32_to": [_vp, _vp, _i64, _i32, _vp],
    "ffm_cast_to_f32": [_vp, _vp, _i64, _i32, _vp],
    "ffm_transpose_cast": [_vp, _vp, _i32, _i32, _i32, _vp],
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library or fail loudly (no CPU / PyTorch fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m fairfedmed_amd.build` "
            "(the FairLoRA engine has no fallback path)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name.endswith(("_ws_bytes", "_splitk_floats")) else C.c_int
    v = lib.ffm_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libffm_hip.so ABI {v} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status}"
                           + (" (invalid argument)" if status == -1 else " (unsupported size)" if status == -2
                              else " (HIP error)"))


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise TypeError(f"unsupported activation dtype {dt}")


def is16(dt: torch.dtype) -> bool:
    """The two 16-bit storage modes (bfloat16: the throughput mode BASELINE.json names; float16: the reference's own
    PREC="fp16") run the same kernels - every `dtype == torch.bfloat16` decision of the engines is really `is16(dtype)`."""
    return dt in (torch.bfloat16, torch.float16)


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream
