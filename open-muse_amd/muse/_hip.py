"""ctypes binding of libmuse_hip.so (C-ABI declared in include/muse_hip.h).

There is NO fallback: every compute entry point of this package goes through this library, and `lib()` raises if
the shared object is missing or a symbol cannot be resolved.  PyTorch is used only for device memory, streams and
torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

F32, BF16, F16 = 0, 1, 2     # F16: GEMM operands only (muse_gemm / muse_gemm_group)
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MUSE_HIP_LIB") or os.path.join(_HERE, "libmuse_hip.so")   # (override: kernel experiments)
_lib = None

c_void_p, c_int, c_i64, c_float = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("bias", c_void_p), ("rowvec", c_void_p), ("residual", c_void_p),
        ("dtype", c_int), ("out_dtype", c_int), ("layout_a", c_int), ("layout_b", c_int),
        ("M", c_int), ("N", c_int), ("K", c_int), ("batch", c_int), ("zdiv", c_int),
        ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64), ("ldr", c_i64),
        ("sA0", c_i64), ("sA1", c_i64), ("sB0", c_i64), ("sB1", c_i64), ("sC0", c_i64), ("sC1", c_i64),
        ("alpha", c_float), ("accumulate", c_int), ("act", c_int), ("split_k", c_int), ("split_stride", c_i64),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("o", c_void_p),
        ("ldq", c_i64), ("ldk", c_i64), ("ldv", c_i64), ("ldo", c_i64),
        ("bsq", c_i64), ("bsk", c_i64), ("bsv", c_i64), ("bso", c_i64),
        ("batch", c_int), ("heads", c_int), ("head_dim", c_int), ("seq_q", c_int), ("seq_kv", c_int),
        ("alpha", c_float),
    ]


# name -> argtypes (every function returns int unless listed in _RESTYPES)
SIGNATURES = {
    "muse_version": [],
    "muse_gemm": [C.POINTER(GemmDesc), c_void_p],
    "muse_gemm_x3": [C.POINTER(GemmDesc), c_i64, c_i64, c_void_p],
    "muse_gemm_tile": [C.POINTER(GemmDesc)],
    "muse_gemm_path": [C.POINTER(GemmDesc)],
    "muse_transpose": [c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_i64, c_int, c_i64, c_i64, c_void_p],
    "muse_layernorm_fwd": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                           c_float, c_void_p],
    "muse_layernorm_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                           c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "muse_layernorm_bwd_nblk": [c_int],
    "muse_layernorm_pair_fwd": [c_void_p] * 10 + [c_int, c_int, c_float, c_void_p],
    "muse_layernorm_pair_bwd": [c_void_p] * 14 + [c_int, c_int, c_int, c_void_p],
    "muse_colsum": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "muse_add_rowvec": [c_void_p, c_void_p, c_i64, c_int, c_void_p],
    "muse_bias_grad_rows_per_block": [],
    "muse_bias_grad_partial": [c_void_p, c_int, c_void_p, c_i64, c_int, c_i64, c_void_p],
    "muse_softmax_fwd": [c_void_p, c_void_p, c_int, c_i64, c_int, c_i64, c_void_p],
    "muse_softmax_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_i64, c_void_p],
    "muse_attention_seq_pad": [c_int],
    "muse_attention_fwd_ex": [C.POINTER(AttnDesc), c_void_p, c_void_p],
    "muse_attention_bwd_ex": [C.POINTER(AttnDesc), c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p,
                              c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p],
    "muse_attention_x3_fwd": [C.POINTER(AttnDesc), c_void_p, c_void_p, c_i64, c_void_p],
    "muse_attention_x3_bwd": [C.POINTER(AttnDesc), c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p,
                              c_i64, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p],
    "muse_attention_x3_fwd_stream": [C.POINTER(AttnDesc), c_void_p, c_void_p, c_i64, c_void_p],
    "muse_attention_x3_bwd_stream": [C.POINTER(AttnDesc), c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64,
                                     c_void_p, c_i64, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p],
    "muse_attention_x3_merge": [c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p],
    "muse_sum_parts_strided": [c_void_p, c_i64, c_int, c_i64, c_int, c_void_p, c_i64, c_int, c_void_p],
    "muse_attention_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "muse_attention_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                           c_void_p],
    "muse_glu_fwd": [c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_glu_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_glu_fwd_x3": [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p],
    "muse_glu_bwd_x3": [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p],
    "muse_ffn_mid_rows_per_block": [],
    "muse_ffn_mid_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p],
    "muse_ffn_mid_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                         c_void_p],
    "muse_gelu_fwd": [c_void_p, c_void_p, c_int, c_i64, c_void_p],
    "muse_gelu_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p],
    "muse_gelu_bwd_f32_bf16": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p],
    "muse_embed_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "muse_embed_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_embed_bwd_scratch_floats": [c_int, c_int],
    "muse_embed_bwd2": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_embed_bwd2_scratch_bytes": [c_int, c_int, c_int, c_int],
    "muse_cross_entropy_fwd": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_float,
                               c_void_p],
    "muse_cross_entropy_bwd": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int,
                               c_i64, c_float, c_void_p],
    "muse_adamw_flat": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_float, c_float, c_float, c_float,
                        c_float, c_int, c_float, c_void_p],
    "muse_adamw_multi": [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p],
    "muse_ema_multi": [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "muse_adamw_flat_groups": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int, c_void_p,
                               c_int, c_int, c_float, c_void_p],
    "muse_adamw_multi_groups": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float, c_void_p],
    "muse_gemm_group_ok": [c_void_p, c_int, c_int],
    "muse_gemm_group": [c_void_p, c_int, c_int, c_void_p],
    "muse_sum_multi": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p],
    "muse_conv_out_direct": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_upsample2x_split_nhwc": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "muse_split_f32_to_bf16x2": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p],
    "muse_upsample2x_nhwc": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_sum_slices_epilogue": [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_int, c_i64, c_i64, c_int, c_void_p],
    "muse_split_f32_to_bf16_cat3": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_i64, c_int, c_int, c_void_p],
    "muse_sum_slices": [c_void_p, c_void_p, c_int, c_i64, c_i64, c_int, c_void_p],
    "muse_cast_f32_to_bf16": [c_void_p, c_void_p, c_i64, c_void_p],
    "muse_cast_bf16_to_f32": [c_void_p, c_void_p, c_i64, c_void_p],
    "muse_cast_f32_to_f16": [c_void_p, c_void_p, c_i64, c_float, c_void_p, c_void_p],
    "muse_operand_images": [c_int, c_float, c_void_p],
    "muse_adamw_skip_flag": [c_void_p],
    "muse_mask_sample": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_i64,
                         c_i64, c_float, c_void_p],
    "muse_sample_step": [c_void_p, c_void_p, c_float, c_i64, c_i64, c_int, c_void_p, c_i64, c_void_p, c_void_p, C.c_uint64, C.c_uint32,
                         c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "muse_mask_tokens": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                         c_i64, c_float, c_int, c_float, c_void_p],
    "muse_dropout": [c_void_p, c_void_p, c_int, c_i64, c_float, C.c_uint64, C.c_uint64, c_void_p],
    "muse_cond_dropout": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_float, c_void_p],
    "muse_conv2d_nhwc": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_int, c_int, c_void_p],
    "muse_conv2d_nhwc_split": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_int, c_void_p],
    "muse_conv2d_nhwc_split2": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_int, c_int, c_int, c_int, c_void_p],
    "muse_conv_in_direct": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_conv2d_nhwc_gn_split2_ok": [c_int, c_int, c_int, c_int, c_int, c_int],
    "muse_conv_persistent": [c_int],
    "muse_conv2d_nhwc_gn_split2": [c_void_p] * 9 + [c_int] * 7 + [c_void_p],
    "muse_groupnorm_scale_shift": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "muse_groupnorm_silu_nhwc_split": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_int, c_float, c_int, c_void_p],
    "muse_groupnorm_silu_nhwc": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_float, c_int, c_void_p],
    "muse_groupnorm_nchunk": [c_int],
    "muse_avgpool2x2_nhwc": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_avgpool2x2_nhwc_stats": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_nchw_to_nhwc": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_nhwc_to_nchw": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "muse_argmin_rows": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_void_p],
    "muse_row_sumsq": [c_void_p, c_void_p, c_i64, c_int, c_i64, c_void_p],
    "muse_gather_rows": [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_norm_res_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_int, c_void_p],
    "muse_adaln_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_adaln_fwd_ex": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_silu_fwd": [c_void_p, c_void_p, c_i64, c_void_p],
    "muse_dwconv3x3_nhwc": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "muse_space_to_depth2_nhwc": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "muse_grn_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_grn_fwd_ex": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_sinusoidal_encode": [c_void_p, c_void_p, c_i64, c_int, c_float, c_void_p],
    "muse_weighted_mean": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p],
    "muse_norm_res_bwd_nblk": [c_i64],
    "muse_norm_res_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_int, c_void_p],
    "muse_norm_res_bwd_ex": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_int, c_void_p],
    "muse_adaln_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_norm_adaln_fwd": [c_void_p] * 7 + [c_int, c_i64, c_int, c_float, c_int, c_void_p],
    "muse_norm_adaln_bwd": [c_void_p] * 9 + [c_int, c_i64, c_int, c_float, c_int, c_void_p],
    "muse_norm_adaln_fwd_x3": [c_void_p] * 7 + [c_int, c_i64, c_int, c_float, c_int, c_void_p],
    "muse_norm_adaln_bwd_x3": [c_void_p] * 9 + [c_int, c_i64, c_int, c_float, c_int, c_void_p],
    "muse_colsum_segments": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "muse_silu_bwd": [c_void_p, c_void_p, c_void_p, c_i64, c_void_p],
    "muse_dwconv3x3_bwd_nchunk": [c_i64],
    "muse_dwconv3x3_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "muse_grn_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p],
    "muse_scale_rows": [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_i64, c_void_p],
    "muse_probe_tr16": [c_void_p, c_void_p, c_void_p],
}
_RESTYPES = {"muse_embed_bwd_scratch_floats": c_i64, "muse_embed_bwd2_scratch_bytes": c_i64}


class MuseHipError(RuntimeError):
    pass


def lib():
    """Load libmuse_hip.so once; raise loudly if the native library is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MuseHipError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (or `make -C open-muse_amd/csrc`). "
                "This package has no CPU / eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise MuseHipError(f"{what} failed with code {code} "
                           f"({'hipError' if code > 0 else {-1: 'bad argument', -2: 'alignment', -3: 'unsupported'}.get(code, '?')})")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise MuseHipError(f"unsupported dtype {t.dtype}")


def ptr(t):
    return None if t is None else t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MuseHipError("muse (MI355X build): tensors must live on the GPU; there is no CPU compute path")
