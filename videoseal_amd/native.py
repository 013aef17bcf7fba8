"""ctypes binding of libvideoseal_hip.so (the C-ABI declared in include/videoseal_hip.h).

This is the stub a maintainer of the reference would add to call the MI355X kernels
(INTEGRATION.md).  There is NO fallback: if the shared library is missing or an entry
point fails, the call raises -- the product path never silently runs on ATen / CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIDEOSEAL_LIB: another build of the same ABI (same-box A/B of two source trees, tools/ab_libs.sh); the default is the in-tree library
LIB_PATH = os.environ.get("VIDEOSEAL_LIB") or os.path.join(_HERE, "csrc", "libvideoseal_hip.so")

ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH, ACT_SILU = 0, 1, 2, 3, 4
PAD_ZERO, PAD_REFLECT = 0, 1
CONV_FORCE_F32, CONV_FORCE_SPLIT, CONV_TILE_HI, CONV_PRE = 0x10, 0x20, 0x40, 0x80
VIDEO_MODES = {"repeat": 0, "alternate": 1, "interpolate": 2}

EXPORTS = [
    "vs_version", "vs_arch", "vs_error_string", "vs_sizeof_conv_desc", "vs_sizeof_tail_desc", "vs_debug_set",
    "vs_model_create", "vs_model_destroy", "vs_model_workspace_bytes", "vs_model_embed", "vs_model_detect", "vs_conv_gemm", "vs_to_planes", "vs_to_planes_affine", "vs_layernorm_act", "vs_layernorm_patch2x2", "vs_stem_conv_ln", "vs_rmsnorm_act", "vs_vit_attention", "vs_dwconv7_ln", "vs_dwconv7_ln_planes", "vs_grn_scale", "vs_grn_scale_from_partials", "vs_grn_scale_from_straddle_partials", "vs_grn_apply",
    "vs_upcat2x", "vs_upconv_supported", "vs_upconv_gather_ln", "vs_cat2_scale", "vs_msg_pre", "vs_upconv_fused_supported", "vs_upconv_fused_preferred", "vs_upconv_fused", "vs_im2col3x3", "vs_msg_latent", "vs_broadcast_channels", "vs_outc_tanh", "vs_pool_linear", "vs_resize_pre", "vs_resize_pre_u8",
    "vs_jnd_heatmap", "vs_embed_tail", "vs_aug_color_scratch_floats", "vs_aug_color", "vs_aug_color_chain", "vs_aug_crop_resize_color", "vs_aug_crop_flip", "vs_aug_warp", "vs_resize_nchw",
    "vs_gaussian_blur", "vs_median_filter", "vs_jpeg_workspace_bytes", "vs_jpeg_roundtrip", "vs_h264_proxy_workspace_bytes", "vs_h264_proxy_roundtrip",
    "vs_bn_partial_doubles", "vs_bn_batch_stats", "vs_bn_partial_sums", "vs_bn_finish_sums", "vs_scale_shift_act", "vs_aug_mask_blend", "vs_aug_add_scaled", "vs_aug_gather_frames", "vs_aug_window_average",
    "vs_gemm_wgrad_partial_floats", "vs_gemm_wgrad", "vs_conv3x3_wgrad", "vs_conv3x3_wgrad_supported", "vs_conv3x3_wgrad_partial_floats", "vs_pad_embed1", "vs_reflect_fold1", "vs_pack_conv", "vs_gaussian_blur_bwd", "vs_aug_warp_bwd", "vs_aug_gather_frames_bwd", "vs_aug_window_average_bwd", "vs_gelu_bwd", "vs_act_bwd", "vs_rmsnorm_act_bwd", "vs_vit_attention_bwd_scratch_floats", "vs_vit_attention_bwd", "vs_dwconv7", "vs_dwconv7_wgrad_partial_floats", "vs_dwconv7_wgrad", "vs_colreduce_partial_floats",
    "vs_layernorm_bwd", "vs_gelu_grn_bwd", "vs_patchify", "vs_unpatch", "vs_patchify_s", "vs_unpatch_s", "vs_col2im3x3_reflect", "vs_colmean", "vs_pool_gelu_bwd", "vs_matmul_small",
    "vs_bce_logits",
    "vs_bn_mean_rstd", "vs_bn_bwd_partial_floats", "vs_bn_relu_bwd_sums", "vs_bn_relu_bwd_apply", "vs_dilate2", "vs_im2col3x3_strided", "vs_upcat2x_bwd",
    "vs_msg_table_grad", "vs_outc_tanh_bwd", "vs_relu_bwd",
    "vs_resize_nchw_bwd", "vs_embed_tail_bwd", "vs_tail_key_reduce", "vs_aug_crop_flip_bwd", "vs_mask_mul", "vs_aug_color_bwd_scratch_floats",
    "vs_aug_color_bwd", "vs_clamp01_bwd", "vs_nhwc_to_nchw_scaled", "vs_percep_partial_doubles", "vs_percep_mse", "vs_percep_mse_grad",
    "vs_split_block", "vs_check_finite", "vs_absmax", "vs_resblock_thin", "vs_resblock_thin_supported", "vs_cnx_block_supported", "vs_cnx_block_image_bytes", "vs_cnx_block",
]


class NativeError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """mirror of vs_conv_desc_t"""
    _fields_ = [
        ("inp", C.c_void_p), ("in_sb", C.c_int64), ("in_sy", C.c_int64), ("in_sx", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("SH", C.c_int32), ("SW", C.c_int32), ("PH", C.c_int32), ("PW", C.c_int32),
        ("pad_mode", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("wt", C.c_void_p), ("CinP", C.c_int32), ("N", C.c_int32),
        ("a_scale", C.c_void_p), ("a_scale_ld", C.c_int64), ("a_shift", C.c_void_p),
        ("bias", C.c_void_p), ("act", C.c_int32), ("n_store", C.c_int32),
        ("res", C.c_void_p), ("res_ld", C.c_int64),
        ("in2", C.c_void_p), ("in2_ld", C.c_int64), ("Cin2", C.c_int32), ("Cin2P", C.c_int32),
        ("wt2", C.c_void_p), ("bias2", C.c_void_p),
        ("out", C.c_void_p), ("out_ld", C.c_int64), ("out_coff", C.c_int32), ("tile_hint", C.c_int32),
        ("wt_split", C.c_void_p), ("wt2_split", C.c_void_p), ("wt_blk", C.c_void_p), ("wt2_blk", C.c_void_p),
        ("splitk_ws", C.c_void_p), ("splitk_ld", C.c_int64), ("split_k", C.c_int32), ("arith", C.c_int32),
        ("sumsq_part", C.c_void_p), ("in_pl", C.c_void_p), ("in2_pl", C.c_void_p), ("out_pl", C.c_void_p), ("a_mul", C.c_float), ("acc_mul", C.c_float), ("acc_mul2", C.c_float), ("grn_nchunk", C.c_int32), ("grn_part", C.c_void_p), ("grn_gamma", C.c_void_p), ("sumsq_hw", C.c_int32), ("reserved_", C.c_int32),
    ]


class TailDesc(C.Structure):
    """mirror of vs_tail_desc_t"""
    _fields_ = [
        ("imgs", C.c_void_p), ("out", C.c_void_p), ("preds_w", C.c_void_p),
        ("delta", C.c_void_p), ("hmap_lowres", C.c_void_p), ("taps43", C.c_void_p),
        ("F", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("S_h", C.c_int32), ("S_w", C.c_int32), ("Cd", C.c_int32),
        ("step", C.c_int32), ("video_mode", C.c_int32), ("total_key", C.c_int32),
        ("attenuate", C.c_int32), ("clamp", C.c_int32), ("antialias", C.c_int32),
        ("scaling_i", C.c_float), ("scaling_w", C.c_float), ("io_u8", C.c_int32), ("variant", C.c_int32),
    ]


class ResblockThinDesc(C.Structure):
    """mirror of vs_resblock_thin_desc_t"""
    _fields_ = [
        ("x", C.c_void_p), ("x_ld", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("w0_split", C.c_void_p), ("w1_split", C.c_void_p), ("wr_split", C.c_void_p),
        ("b0", C.c_void_p), ("b1", C.c_void_p), ("br", C.c_void_p),
        ("arith", C.c_int32), ("Cout", C.c_int32),
        ("a_mul", C.c_float), ("acc_mul0", C.c_float), ("acc_mul1", C.c_float), ("acc_mulr", C.c_float),
        ("out", C.c_void_p), ("out_ld", C.c_int64),
    ]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(or `make -C videoseal_amd/csrc`). There is no CPU/ATen fallback for the VideoSeal hot path.")
    L = C.CDLL(LIB_PATH)
    L.vs_version.restype = C.c_int
    L.vs_arch.restype = C.c_char_p
    L.vs_error_string.restype = C.c_char_p
    L.vs_error_string.argtypes = [C.c_int]
    P, I, I64, F = C.c_void_p, C.c_int, C.c_int64, C.c_float
    sig = {
        "vs_conv_gemm": [C.POINTER(ConvDesc), P],
        "vs_to_planes": [P, I64, I, I64, F, P, P],
        "vs_to_planes_affine": [P, I64, I, I64, F, P, I64, P, I, P, P],
        "vs_layernorm_act": [P, I64, I, I64, P, P, F, I, P, I64, P],
        "vs_layernorm_patch2x2": [P, I, I, I, I, I64, P, P, F, P, P],
        "vs_stem_conv_ln": [P, I, I, I, I, P, P, P, P, F, I, P, I64, P],
        "vs_rmsnorm_act": [P, I64, I, I64, P, I, P, I64, P, I64, P],
        "vs_vit_attention": [P, I, I, I, I, I, I, P, P, P, P],
        "vs_dwconv7_ln": [P, I, I, I, I, I64, P, P, P, P, F, P, I64, P],
        "vs_dwconv7_ln_planes": [P, I, I, I, I, I64, P, P, P, P, F, F, I, P, P],
        "vs_grn_scale": [P, I, I, I, I64, P, P, P, P],
        "vs_grn_scale_from_partials": [P, I, I, I, P, P, I64, P],
        "vs_grn_scale_from_straddle_partials": [P, I, I, I, P, P, I64, P],
        "vs_grn_apply": [P, I, I, I, I64, P, I64, P, P],
        "vs_upcat2x": [P, I, I64, P, I, I64, F, I, I, I, P, I64, P],
        "vs_upconv_supported": [I],
        "vs_upconv_gather_ln": [P, I64, I, I, I, I, P, P, F, I, P, I64, P],
        "vs_cat2_scale": [P, I, I64, P, I, I64, F, I64, P, I64, P],
        "vs_msg_pre": [P, I, I, P, P],
        "vs_upconv_fused_supported": [I, I, I],
        "vs_upconv_fused_preferred": [I, I, I],
        "vs_upconv_fused": [P, I, I64, P, I, I64, F, P, I, I, I, I, P, P, F, I, P, I64, I, F, F, P],
        "vs_im2col3x3": [P, I, I, I, I64, I, P, P],
        "vs_msg_latent": [P, P, I, I, I, P, P],
        "vs_broadcast_channels": [P, I, I, P, I, I, I64, I, P],
        "vs_outc_tanh": [P, I64, I, I, I64, P, P, I, I, P, P],
        "vs_pool_linear": [P, I, I, I, I64, P, P, I, P, P],
        "vs_resize_pre": [P, I, I, I, I, I, I, I, P, F, F, P, I, P, P],
        "vs_resize_pre_u8": [P, I, I, I, I, I, I, P, F, F, P, I, P, P],
        "vs_jnd_heatmap": [P, I, I, I, I64, I64, I64, I64, P, P, P],
        "vs_embed_tail": [C.POINTER(TailDesc), P],
        "vs_aug_color": [P, P, I, I, I, I, F, P, P],
        "vs_aug_crop_flip": [P, P, I, I, I, I, I, I, I, I, P],
        "vs_aug_color_chain": [P, P, I, I, I, I, C.POINTER(C.c_int), C.POINTER(C.c_float), P, P],
        "vs_aug_crop_resize_color": [P, P, I, I, I, I, I, I, I, I, I, I, I, C.POINTER(C.c_int), C.POINTER(C.c_float), P],
        "vs_resize_nchw": [P, P, I, I, I, I, I, I, P],
        "vs_aug_warp": [P, P, I, I, I, I, I, I, P, I, P],
        "vs_gaussian_blur": [P, P, P, I, I, I, I, F, P],
        "vs_median_filter": [P, P, I, I, I, I, P],
        "vs_jpeg_roundtrip": [P, P, I, I, I, I, P, P],
        "vs_h264_proxy_roundtrip": [P, P, I, I, I, I, I, P, P],
        "vs_bn_batch_stats": [P, I64, I, I64, P, P, F, F, P, P, P, P, P, P],
        "vs_bn_partial_sums": [P, I64, I, I64, P, P, P],
        "vs_bn_finish_sums": [P, I, I64, P, P, F, F, P, P, P, P, P],
        "vs_scale_shift_act": [P, I64, I, I64, P, P, I, P, I64, P, I64, P],
        "vs_aug_mask_blend": [P, P, P, P, I, I, I, I, P],
        "vs_aug_add_scaled": [P, P, F, P, I64, P],
        "vs_aug_gather_frames": [P, P, P, I, I64, P],
        "vs_aug_window_average": [P, P, I, I64, I, F, P],
        "vs_gemm_wgrad": [P, I64, I, P, I64, I, I64, P, P, P],
        "vs_conv3x3_wgrad": [P, I64, I, P, I64, I, I, I, I, I, P, P, P],
        "vs_conv3x3_wgrad_supported": [I, I64, I],
        "vs_pad_embed1": [P, I, I, I, I64, P, P],
        "vs_reflect_fold1": [P, I, I, I, I64, P, P],
        "vs_pack_conv": [P, I, I, I, I, I, I, P, P],
        "vs_gaussian_blur_bwd": [P, P, P, I, I, I, I, F, P],
        "vs_aug_warp_bwd": [P, P, I, I, I, I, I, I, P, I, P, P],
        "vs_aug_gather_frames_bwd": [P, P, P, P, I, I64, P],
        "vs_aug_window_average_bwd": [P, P, I, I64, I, F, P],
        "vs_gelu_bwd": [P, I64, P, I64, I64, I, P, I64, P],
        "vs_act_bwd": [P, I64, P, I64, I64, I, I, P, I64, P],
        "vs_rmsnorm_act_bwd": [P, I64, I, I64, P, I, P, I64, P, I64, P, I64, P],
        "vs_vit_attention_bwd": [P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, P],
        "vs_dwconv7": [P, I, I, I, I, I64, P, P, I, P, I64, P, I64, P],
        "vs_dwconv7_wgrad": [P, I64, P, I64, I, I, I, I, P, P, P],
        "vs_layernorm_bwd": [P, I64, P, I64, P, I64, I, F, P, I64, P, P, P, P, P],
        "vs_gelu_grn_bwd": [P, I64, P, I64, P, I, I, I, P, P, P, I64, P, P, P],
        "vs_patchify": [P, I, I, I, I64, I, P, P],
        "vs_unpatch": [P, I, I, I, I64, I, P, P],
        "vs_patchify_s": [P, I, I, I, I64, I, I, P, P],
        "vs_unpatch_s": [P, I, I, I, I64, I, I, P, P],
        "vs_col2im3x3_reflect": [P, I, I, I, I64, P, P],
        "vs_colmean": [P, I, I, I64, P, P],
        "vs_pool_gelu_bwd": [P, I64, P, I64, I, I, I, P, I64, P],
        "vs_matmul_small": [P, I64, P, I64, I, I, I, P, I64, P],
        "vs_bce_logits": [P, P, I, I, I, F, F, P, P, P],
        "vs_bn_mean_rstd": [P, I, I64, F, P, P, P],
        "vs_bn_relu_bwd_sums": [P, I64, P, I64, P, P, P, P, I, I64, I, P, P, P, P, P],
        "vs_bn_relu_bwd_apply": [P, I64, P, I64, P, P, P, P, I, P, I64, I, P, I64, P],
        "vs_dilate2": [P, I, I, I, I64, I, I, P, P],
        "vs_im2col3x3_strided": [P, I, I, I, I64, I, P, P],
        "vs_upcat2x_bwd": [P, I64, I, I, I, I, I, F, P, I64, P, I64, P],
        "vs_msg_table_grad": [P, P, I, I, I, P, P],
        "vs_outc_tanh_bwd": [P, P, I64, I, I, P, I, I, P, I64, P, P],
        "vs_relu_bwd": [P, I64, P, I64, I64, I, P, I64, P],
        "vs_resize_nchw_bwd": [P, P, I, I, I, I, I, I, P, P],
        "vs_embed_tail_bwd": [P, P, P, P, P, I, I, I, I, I, F, F, P, P],
        "vs_tail_key_reduce": [P, P, I, I, I, I, I, I, I, P, P],
        "vs_aug_crop_flip_bwd": [P, P, I, I, I, I, I, I, I, I, P],
        "vs_mask_mul": [P, P, P, I, I, I, I, I, P],
        "vs_aug_color_bwd": [P, P, P, I, I, I, I, F, P, P, P],
        "vs_clamp01_bwd": [P, P, P, I64, P],
        "vs_nhwc_to_nchw_scaled": [P, I, I, I, I, I64, F, P, P],
        "vs_percep_mse": [P, P, I, I, I, I, P, P, P],
        "vs_percep_mse_grad": [P, P, I, I, I, I, F, P, P],
        "vs_split_block": [P, I, I64, I, I, F, P, P, P],
        "vs_check_finite": [P, I64, P, P],
        "vs_absmax": [P, I64, P, P],
        "vs_resblock_thin": [P, P],
        "vs_resblock_thin_supported": [I, I, I],
        "vs_cnx_block": [P, P, I, I64, I, I, F, F, P, I64, P, P, I64, P, I64, P, P],
        "vs_cnx_block_supported": [I, I64, I],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.vs_aug_color_scratch_floats.restype = C.c_int64
    L.vs_aug_color_scratch_floats.argtypes = [I, I, I]
    L.vs_jpeg_workspace_bytes.restype = C.c_int64
    L.vs_jpeg_workspace_bytes.argtypes = [I, I, I]
    L.vs_h264_proxy_workspace_bytes.restype = C.c_int64
    L.vs_h264_proxy_workspace_bytes.argtypes = [I, I, I]
    L.vs_vit_attention_bwd_scratch_floats.restype = C.c_int64
    L.vs_vit_attention_bwd_scratch_floats.argtypes = [I, I, I, I, I]
    for name in ("vs_aug_color_bwd_scratch_floats", "vs_percep_partial_doubles"):
        getattr(L, name).restype = C.c_int64
        getattr(L, name).argtypes = [I, I, I]
    L.vs_cnx_block_image_bytes.restype = C.c_int64
    L.vs_cnx_block_image_bytes.argtypes = [I]
    L.vs_bn_partial_doubles.restype = C.c_int64
    L.vs_bn_partial_doubles.argtypes = [I64, I64]
    for name, args in (("vs_gemm_wgrad_partial_floats", [I64, I, I]), ("vs_conv3x3_wgrad_partial_floats", [I, I64, I, I, I, I]), ("vs_dwconv7_wgrad_partial_floats", [I, I, I64]),
                       ("vs_colreduce_partial_floats", [I, I64, I64]), ("vs_bn_bwd_partial_floats", [I64, I64])):
        getattr(L, name).restype = C.c_int64
        getattr(L, name).argtypes = args
    L.vs_sizeof_conv_desc.restype = C.c_int
    L.vs_sizeof_tail_desc.restype = C.c_int
    if L.vs_version() != 3:
        raise NativeError(f"{LIB_PATH} has ABI version {L.vs_version()}, this binding is written for 3: rebuild (make -C videoseal_amd/csrc)")
    if L.vs_sizeof_conv_desc() != C.sizeof(ConvDesc) or L.vs_sizeof_tail_desc() != C.sizeof(TailDesc):
        raise NativeError("ctypes mirrors of vs_conv_desc_t / vs_tail_desc_t are out of date with the shared library")
    _lib = L
    return L


ERR_UNSUPPORTED = -2      # VS_ERR_UNSUPPORTED: configuration outside what a kernel implements (callers with another form fall back to it)


def check(code: int, what: str) -> None:
    if code != 0:
        raise NativeError(f"{what} failed: {lib().vs_error_string(code).decode()} (code {code})")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeError("HIP kernels need device tensors (got a CPU tensor); move the model/inputs to cuda first")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def f32c(t: torch.Tensor) -> torch.Tensor:
    """contiguous float32 view/copy (plumbing)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
