"""ctypes binding of libdvt_hip.so (the C ABI declared in include/dvt_hip.h).

The product path has no fallback: if the shared library is missing or a call
fails, a RuntimeError carrying ``dvt_last_error()`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DVT_LIB_PATH: development A/B of another build of the same ABI -- tools/dev/ab_libs.sh; the product loads the in-tree library)
LIB_PATH = os.environ.get("DVT_LIB_PATH") or os.path.join(_HERE, "libdvt_hip.so")

F32, BF16, F16 = 0, 1, 2
ABI_VERSION = 5            # == DVT_ABI_VERSION of include/dvt_hip.h (bumped with every descriptor layout change)
EPI_NONE, EPI_GELU, EPI_RELU, EPI_RESIDUAL, EPI_DGELU, EPI_DRELU = range(6)

c_i64 = C.c_int64
c_f = C.c_float
c_p = C.c_void_p
c_int = C.c_int


class SplitKPending(C.Structure):
    _fields_ = [
        ("slab", c_p), ("splits", C.c_int32), ("valid", C.c_int32),
        ("M", c_i64), ("N", c_i64), ("C", c_p), ("ldc", c_i64),
        ("accumulate", C.c_int32), ("cs_accumulate", C.c_int32),
        ("cs_slab", c_p), ("cs_out", c_p),
        ("conv_cin", C.c_int32), ("conv_taps", C.c_int32), ("conv_cin_l", C.c_int32), ("conv_cout_l", C.c_int32),
    ]


class LnPending(C.Structure):
    _fields_ = [("partial", c_p), ("nparts", C.c_int32), ("d", C.c_int32), ("dgamma", c_p), ("dbeta", c_p),
                ("accumulate", C.c_int32), ("valid", C.c_int32)]


class LnBwdDesc(C.Structure):
    _fields_ = [("dy", c_p), ("dy_dtype", C.c_int32), ("x", c_p), ("x_dtype", C.c_int32),
                ("gamma", c_p), ("mean", c_p), ("rstd", c_p), ("dx_add", c_p),
                ("dx", c_p), ("dx_dtype", C.c_int32), ("dx_lp", c_p), ("dx_lp_dtype", C.c_int32),
                ("dgamma", c_p), ("dbeta", c_p), ("workspace", c_p),
                ("n0", c_i64), ("n1", c_i64), ("d", c_i64), ("xs0", c_i64), ("xs1", c_i64), ("ys0", c_i64), ("ys1", c_i64),
                ("dy_first", c_p), ("dy_first_stride", c_i64), ("dx_first", c_p), ("dx_first_stride", c_i64),
                ("accumulate_gamma", C.c_int32), ("accumulate_beta", C.c_int32),
                ("defer_reduce", C.c_int32), ("pending", C.POINTER(LnPending))]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_p), ("B", c_p), ("C", c_p),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64),
        ("a_kmajor", C.c_int32), ("b_kmajor", C.c_int32),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("epilogue", C.c_int32), ("accumulate", C.c_int32),
        ("bias", c_p), ("residual", c_p), ("ldr", c_i64),
        ("aux", c_p), ("ldaux", c_i64),
        ("alpha", c_f), ("split_k", C.c_int32),
        ("workspace", c_p),
        ("colsum_out", c_p), ("colsum_accumulate", C.c_int32),
        ("defer_reduce", C.c_int32), ("pending", C.POINTER(SplitKPending)), ("carry", C.POINTER(SplitKPending)),
        ("residual_f32", C.c_int32),
    ]


class ConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("N", C.c_int64)] + \
               [(n, C.c_int32) for n in ("H", "W", "C", "Cout", "kh", "kw", "sh", "sw", "ph", "pw", "dtype")] + \
               [("workspace", C.c_void_p), ("stats_partial", C.c_void_p), ("trim_w", C.c_int32),
                ("defer_reduce", C.c_int32), ("pending", C.c_void_p), ("carry", C.c_void_p),
                ("residual", C.c_void_p), ("wgrad_master_layout", C.c_int32), ("wgrad_accumulate", C.c_int32),
                ("wgrad_cout_l", C.c_int32), ("wgrad_cin_l", C.c_int32),
                ("out_h", C.c_int32), ("out_w", C.c_int32), ("out_rows", C.c_void_p), ("residual_compact", C.c_int32)]


class BnAffine(C.Structure):
    _fields_ = [("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("c_valid", C.c_int32), ("relu", C.c_int32)]


class PackEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("cout_l", "cin_l", "kh", "kw", "cout_p", "cin_p", "ld", "kind", "dtype",
                                         "cls_sh", "cls_sw", "cls_rh", "cls_rw")]


class HeadBceDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "g1", "b1", "g2", "b2", "w", "c", "target", "logits", "loss", "grads")] + \
               [(n, C.c_int32) for n in ("rows", "d", "classes", "x_dtype")] + [("eps1", C.c_float), ("eps2", C.c_float)]


class EmitEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dst_lp", C.c_void_p), ("n", C.c_int64),
                ("accumulate", C.c_int32), ("lp_dtype", C.c_int32)]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", c_p), ("k", c_p), ("v", c_p), ("o", c_p), ("lse", c_p),
        ("d_o", c_p), ("dq", c_p), ("dk", c_p), ("dv", c_p),
        ("B", c_i64), ("H", c_i64), ("Lq", c_i64), ("Lk", c_i64), ("dh", c_i64),
        ("q_sb", c_i64), ("q_sh", c_i64), ("q_sl", c_i64),
        ("k_sb", c_i64), ("k_sh", c_i64), ("k_sl", c_i64),
        ("v_sb", c_i64), ("v_sh", c_i64), ("v_sl", c_i64),
        ("o_sb", c_i64), ("o_sh", c_i64), ("o_sl", c_i64),
        ("scale", c_f), ("dtype", C.c_int32),
        ("workspace", c_p),
        ("dropout_p", c_f), ("rng_state", c_p), ("rng_offset", C.c_uint64),
        ("bwd_two_pass", C.c_int32),
    ]


class AttnClsDesc(C.Structure):
    _fields_ = [
        ("x", c_p), ("xs0", c_i64), ("xs1", c_i64), ("gamma", c_p), ("beta", c_p), ("eps", c_f),
        ("S", c_i64), ("N", c_i64), ("d", c_i64), ("H", c_i64), ("dtype", C.c_int32),
        ("R", c_p), ("A", c_p), ("lse", c_p), ("P", c_p), ("mean", c_p), ("rstd", c_p),
        ("dM", c_p), ("dx", c_p), ("G", c_p), ("dgamma", c_p), ("dbeta", c_p),
        ("accumulate_gamma", C.c_int32), ("accumulate_beta", C.c_int32), ("workspace", c_p),
    ]


# name -> (restype, argtypes); must list every symbol of include/dvt_hip.h
SIGNATURES = {
    "dvt_version": (c_int, []),
    "dvt_last_error": (C.c_char_p, []),
    "dvt_device_info": (c_int, [C.POINTER(c_int), C.POINTER(c_int), C.c_char_p, c_int]),
    "dvt_cast": (c_int, [c_p, c_int, c_p, c_int, c_i64, c_p]),
    "dvt_add": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_p]),
    "dvt_add_rowtable": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_int, c_p]),
    "dvt_copy2d": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    "dvt_rows_sum": (c_int, [c_p, c_i64, c_i64, c_i64, c_p, c_int, c_int, c_p]),
    "dvt_permute_021": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_int, c_p]),
    "dvt_axpby_f32": (c_int, [c_p, c_int, c_f, c_p, c_f, c_i64, c_p]),
    "dvt_act_fwd": (c_int, [c_p, c_p, c_i64, c_int, c_int, c_p]),
    "dvt_act_bwd": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_int, c_p]),
    "dvt_patchify": (c_int, [c_p, c_int, c_p, c_int, c_i64, c_int, c_int, c_int, c_int, c_p]),
    "dvt_patchify_bwd": (c_int, [c_p, c_int, c_p, c_int, c_i64, c_int, c_int, c_int, c_int, c_p]),
    "dvt_tokens_assemble_fwd": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    "dvt_tokens_assemble_bwd": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p]),
    "dvt_rows_gather_fwd": (c_int, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_int, c_p]),
    "dvt_rows_gather_bwd": (c_int, [c_p, c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_int, c_int, c_p]),
    "dvt_mean_rows_fwd": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
    "dvt_mean_rows_bwd": (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
    "dvt_layernorm_fwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                  c_i64, c_f, c_int, c_p]),
    "dvt_layernorm_bwd_workspace_bytes": (C.c_size_t, [c_i64]),
    "dvt_layernorm_fwd_mixed": (c_int, [c_p, c_int, c_p, c_p, c_p, c_int, c_p, c_p] + [c_i64] * 7 + [c_f, c_p]),
    "dvt_layernorm_bwd_partial_bytes": (C.c_size_t, [c_i64, c_i64]),
    "dvt_layernorm_bwd_ex": (c_int, [C.POINTER(LnBwdDesc), c_p]),
    "dvt_layernorm_reduce_group": (c_int, [c_p, c_int, c_p]),
    "dvt_conv_weight_pack_group": (c_int, [c_p, c_int, c_p]),
    "dvt_layernorm_bwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64,
                                  c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p]),
    "dvt_layernorm_bwd_first": (c_int, [c_p] * 10 + [c_i64] * 7 + [c_p, c_i64, c_p, c_i64, c_int, c_int, c_int, c_p]),
    "dvt_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "dvt_gemm": (c_int, [C.POINTER(GemmDesc), c_p]),
    "dvt_gemm_route": (c_int, [C.POINTER(GemmDesc)]),
    "dvt_splitk_reduce_pending": (c_int, [C.POINTER(SplitKPending), c_p]),
    "dvt_gemm_pair_fused": (c_int, [C.POINTER(GemmDesc), C.POINTER(GemmDesc)]),
    "dvt_gemm_pair": (c_int, [C.POINTER(GemmDesc), C.POINTER(GemmDesc), c_p]),
    "dvt_colsum_workspace_bytes": (C.c_size_t, [c_i64, c_i64]),
    "dvt_colsum": (c_int, [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_int, c_int, c_p]),
    "dvt_attention_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(AttnDesc)]),
    "dvt_attention_fwd": (c_int, [C.POINTER(AttnDesc), c_p]),
    "dvt_attention_bwd": (c_int, [C.POINTER(AttnDesc), c_p]),
    "dvt_attn_cls_supported": (c_int, [C.POINTER(AttnClsDesc)]),
    "dvt_attn_cls_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(AttnClsDesc)]),
    "dvt_attn_cls_fwd": (c_int, [C.POINTER(AttnClsDesc), c_p]),
    "dvt_attn_cls_bwd": (c_int, [C.POINTER(AttnClsDesc), c_p]),
    "dvt_heads_expand": (c_int, [c_p, c_i64, c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
    "dvt_heads_contract": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_int, c_p]),
    "dvt_heads_outer": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_f, c_int, c_int, c_p]),
    "dvt_heads_expand_outer": (c_int, [c_p, c_i64, c_p, c_i64, c_p, c_f, c_p, c_p, c_p, c_p, c_i64, c_f, c_int,
                                       c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    "dvt_heads_contract_outer": (c_int, [c_p, c_p, c_p, c_i64, c_p, c_i64, c_f, c_p, c_i64, c_p, c_i64, c_f, c_int,
                                         c_i64, c_i64, c_i64, c_i64, c_int, c_p]),
    "dvt_im2col": (c_int, [c_p, c_int, c_int, c_p, c_int, c_i64] + [c_int] * 9 + [c_i64, c_p]),
    "dvt_col2im": (c_int, [c_p, c_p, c_i64] + [c_int] * 9 + [c_i64, c_p, c_int, c_int, c_p]),
    "dvt_col2im_nchw": (c_int, [c_p, c_int, c_p, c_int, c_i64] + [c_int] * 9 + [c_i64, c_p]),
    "dvt_conv_weight_pack": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_i64, c_p]),
    "dvt_conv_weight_unpack_grad": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_i64, c_int, c_p]),
    "dvt_bn_workspace_bytes": (C.c_size_t, [c_i64, c_int]),
    "dvt_bn_stats": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_int, c_f, c_f, c_int, c_p]),
    "dvt_bn_eval_invstd": (c_int, [c_p, c_p, c_int, c_f, c_p]),
    "dvt_bn_apply_fwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_int, c_int, c_int, c_p]),
    "dvt_bn_bwd": (c_int, [c_p] * 13 + [c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_maxpool_fwd": (c_int, [c_p, c_p, c_p, c_i64] + [c_int] * 7 + [c_p]),
    "dvt_bn_relu_maxpool_fwd": (c_int, [c_p] * 7 + [c_i64, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_bn_bwd_pooled": (c_int, [c_p] * 11 + [c_i64] + [c_int] * 7 + [c_p]),
    "dvt_maxpool_bwd": (c_int, [c_p, c_p, c_p, c_i64] + [c_int] * 7 + [c_p]),
    "dvt_transpose_last2": (c_int, [c_p, c_p, c_i64, c_int, c_int, c_int, c_p]),
    "dvt_head_bce_supported": (c_int, [c_int, c_int, c_int]),
    "dvt_head_bce_grads_elems": (c_i64, [c_int, c_int, c_int]),
    "dvt_head_bce_fwd": (c_int, [c_p, c_p]),
    "dvt_scaled_emit_group": (c_int, [c_p, c_p, c_int, c_p]),
    "dvt_bce_logits_fwd": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_p]),
    "dvt_bce_logits_bwd": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_int, c_p]),
    "dvt_ce_argmax_fwd": (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_int, c_p]),
    "dvt_ce_argmax_bwd": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_int, c_p]),
    "dvt_adamw_step": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_p]),
    "dvt_frames_preprocess_workspace_bytes": (C.c_size_t, [c_i64, c_int, c_int, c_int, c_int]),
    "dvt_frames_preprocess": (c_int, [c_p, c_p, c_int, c_i64, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p]),
    "dvt_f1_samples_workspace_bytes": (C.c_size_t, [c_i64, c_int]),
    "dvt_f1_samples": (c_int, [c_p, c_p, c_i64, c_int, c_p, c_int, c_p, c_p, c_p]),
    "dvt_average_precision_workspace_bytes": (C.c_size_t, [c_i64, c_int]),
    "dvt_average_precision": (c_int, [c_p, c_p, c_i64, c_int, c_p, c_p, c_p, c_p, c_p]),
    "dvt_l2norm_rows_fwd": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_f, c_int, c_p]),
    "dvt_l2norm_rows_bwd": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_int, c_f, c_int, c_p]),
    "dvt_cosine_rows": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_f, c_int, c_p]),
    "dvt_gate_fwd": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_p]),
    "dvt_gate_bwd": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_p]),
    "dvt_contrastive_fwd": (c_int, [c_p, c_int, c_f, c_p, c_p, c_p, c_p]),
    "dvt_contrastive_bwd": (c_int, [c_p, c_p, c_int, c_f, c_p, c_p, c_p]),
    "dvt_adamw_step_scaled": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_p, c_int, c_f,
                                      c_f, c_p, c_f, c_p, c_p]),
    "dvt_device_delay": (c_int, [C.c_uint64, c_p]),
    "dvt_zero": (c_int, [c_p, C.c_size_t, c_p]),
    "dvt_dropout": (c_int, [c_p, c_p, c_i64, c_f, c_p, C.c_uint64, c_int, c_p]),
    "dvt_dropout_fused": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_p, C.c_uint64, c_int, c_int, c_p]),
    "dvt_rng_advance": (c_int, [c_p, C.c_uint64, c_p]),
    "dvt_conv_weight_pack_dgrad": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_pad3_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_unpad3_f32": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_conv2d_implicit_supported": (c_int, [C.POINTER(ConvDesc)]),
    "dvt_conv2d_implicit_k": (c_i64, [C.POINTER(ConvDesc)]),
    "dvt_nchw_to_nhwc_pad": (c_int, [c_p, c_int, c_p, c_int, c_i64, c_int, c_int, c_int, c_int, c_p]),
    "dvt_conv_stem7_supported": (c_int, [c_i64, c_int, c_int, c_int]),
    "dvt_conv_stem7_stats_parts": (c_i64, [c_i64, c_int, c_int]),
    "dvt_conv_stem7": (c_int, [c_p, c_p, c_i64, c_p, c_p, c_i64, c_int, c_int, c_int, c_p]),
    "dvt_conv3x3_c64_supported": (c_int, [c_i64, c_int, c_int, c_int]),
    "dvt_conv3x3_c64_wgrad_supported": (c_int, [c_i64, c_int, c_int, c_int]),
    "dvt_conv3x1_wgrad_supported": (c_int, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    "dvt_conv3x1_wgrad_workspace_bytes": (C.c_size_t, [c_i64, c_int, c_int]),
    "dvt_conv3x1_wgrad": (c_int, [c_p, C.POINTER(BnAffine), c_p, c_p, c_p, c_i64, c_int, c_int, c_int, c_int, C.POINTER(SplitKPending), c_int, c_p]),
    "dvt_conv3x1_fwd_supported": (c_int, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    "dvt_conv3x1_fwd_stats_parts": (c_i64, [c_i64, c_int, c_int, c_int]),
    "dvt_conv3x1_fwd": (c_int, [c_p, C.POINTER(BnAffine), c_p, c_i64, c_p, c_p, c_i64, c_int, c_int, c_int, c_int, c_p]),
    "dvt_conv3x3_c64_wgrad_workspace_bytes": (C.c_size_t, [c_i64, c_int, c_int]),
    "dvt_conv3x3_c64_wgrad": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_int, c_int, c_int, c_int, C.POINTER(SplitKPending), c_int, c_p]),
    "dvt_conv3x3_c64_stats_parts": (c_i64, [c_i64, c_int, c_int]),
    "dvt_conv3x3_c64_wgrad_wide": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, C.POINTER(SplitKPending), c_int, c_p]),
    "dvt_conv3x3_c64": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_int, c_int, c_p]),
    "dvt_conv3x3_stream_supported": (c_int, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    "dvt_conv3x3_stream_stats_parts": (c_i64, [c_i64, c_int, c_int, c_int, c_int]),
    "dvt_conv3x3_stream": (c_int, [c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_conv3x1_stream_supported": (c_int, [c_i64, c_int, c_int, c_int, c_int, c_int]),
    "dvt_conv3x1_stream": (c_int, [c_p, c_p, c_p, c_i64, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_conv3x1_stream_bn_bwd_workspace_bytes": (C.c_size_t, [c_i64, c_int, c_int]),
    "dvt_conv3x1_stream_bn_bwd": (c_int, [c_p, c_p, c_p, C.POINTER(BnAffine), c_p, c_p, c_p, c_p, c_i64, c_int, c_int, c_int,
                                          c_int, c_int, c_p]),
    "dvt_conv_weight_pairs": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_conv_weight_pairs_bwd": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_conv2d_implicit": (c_int, [C.POINTER(ConvDesc), c_p]),
    "dvt_conv2d_implicit_stats_parts": (c_i64, [C.POINTER(ConvDesc)]),
    "dvt_conv2d_implicit_stats_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "dvt_conv2d_implicit_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "dvt_bn_stats_from_partials": (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_i64, c_int, c_int, c_f, c_f, c_p]),
    "dvt_conv2d_implicit_wgrad_supported": (c_int, [C.POINTER(ConvDesc)]),
    "dvt_conv2d_implicit_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "dvt_conv2d_implicit_wgrad": (c_int, [C.POINTER(ConvDesc), c_p]),
    "dvt_conv_weight_unpack_grad_t": (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
    "dvt_sgd_step": (c_int, [c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_p, c_p]),
    "dvt_adagrad_step": (c_int, [c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_i64, c_p, c_p]),
    "dvt_comm_unique_id": (c_int, [c_p]),
    "dvt_comm_init": (c_int, [C.POINTER(c_p), c_p, c_int, c_int]),
    "dvt_comm_allreduce": (c_int, [c_p, c_p, c_i64, c_int, c_p]),
    "dvt_comm_broadcast": (c_int, [c_p, c_p, c_i64, c_int, c_int, c_p]),
    "dvt_comm_destroy": (c_int, [c_p]),
    "dvt_adamw_step_dev": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p]),
    "dvt_adamw_step_fused": (c_int, [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_p, c_int, c_p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python data-efficient-video-transformers_amd/build.py` (needs hipcc); there is no "
            "fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.dvt_version() != ABI_VERSION:
        raise RuntimeError(f"libdvt_hip.so ABI version {lib.dvt_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().dvt_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libdvt_hip {what} failed (status {rc}): {msg}")
