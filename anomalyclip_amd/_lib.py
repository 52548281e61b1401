"""ctypes binding of libacx (include/acx.h).  The library is the product's only compute path: if it
cannot be loaded the package raises, it never falls back to PyTorch ops or to the oracle."""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import _build

c_void_p, c_int32, c_int64, c_float, c_size_t = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

ACX_F32, ACX_BF16, BF16X3, BF16X3P = 0, 1, 2, 3      # BF16X3 / BF16X3P: output only (three bf16 planes hi | mid | lo; P: K-panel layout)
BF16X2P, ACX_F16, F16X2P = 4, 5, 6                     # hi + mid planes only; fp16 planes of the two-plane split (pairs = 3); their K-panel output
PREC_F32, PREC_BF16, PREC_F32X6, PREC_F32X3, PREC_F16X3 = 0, 1, 2, 3, 4
ACT_NONE, ACT_QUICKGELU, ACT_LEAKYRELU = 0, 1, 2
AMAP_IDENTITY, AMAP_CONV3X3, AMAP_TESTTILE, AMAP_TILETABLE = 0, 1, 2, 3
NORM_LAYER, NORM_CHAN = 0, 1
OPT_RING_MIN_TILES, OPT_SK_MAX_M, OPT_TN_P256_MIN_ROWS, OPT_X6_CUS, OPT_X6_TAIL_SPLIT = 1, 2, 3, 4, 5
OPT_X6_STRIP_TAIL, OPT_X6_MIN_TILES = 6, 7
ACX_F64, ACX_I64 = 16, 17                              # element types of the collectives only (acx_allreduce / acx_allgather)
COMM_SUM, COMM_MAX, COMM_MIN = 0, 1, 2
COMM_ID_BYTES = 128


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldw", c_int32), ("ldc", c_int32),
        ("a_dtype", c_int32), ("c_dtype", c_int32), ("prec", c_int32),
        ("bias", c_void_p), ("act", c_int32),
        ("residual", c_void_p), ("ldr", c_int32),
        ("a_sub", c_void_p),
        ("amap", c_int32), ("gn", c_int32), ("gl", c_int32), ("cin", c_int32), ("seg", c_int32),
        ("pos0", c_void_p), ("pos1", c_void_p),
        ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("zero_page", c_void_p), ("a_act", c_int32), ("gelu_grad_of", c_void_p), ("ldg", c_int32),
        ("a_norm_w", c_void_p), ("a_norm_b", c_void_p), ("a_norm_eps", c_float),
        ("counters", c_void_p), ("n_counters", c_int32),
        ("tile_table", c_void_p),
        ("pairs", c_int32), ("panels", c_int32), ("a_plane_stride", c_int64), ("w_plane_stride", c_int64),
        ("c_plane_rows", c_int64), ("out_scale", c_float),
    ]


class BlockWeights(C.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "ln1_w", "ln1_b", "ln2_w", "ln2_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
        "fc_w", "fc_b", "proj_w", "proj_b",
        "in_proj_w_bf16", "out_proj_w_bf16", "fc_w_bf16", "proj_w_bf16")]


class VitWeights(C.Structure):
    _fields_ = [
        ("conv1_w", c_void_p), ("conv1_w_bf16", c_void_p),
        ("class_embedding", c_void_p), ("positional_embedding", c_void_p),
        ("ln_pre_w", c_void_p), ("ln_pre_b", c_void_p), ("ln_post_w", c_void_p), ("ln_post_b", c_void_p),
        ("proj_t", c_void_p), ("proj_t_bf16", c_void_p),
        ("blocks", C.POINTER(BlockWeights)),
    ]


class CurveResult(C.Structure):          # == struct acx_curve_result (56 bytes, device memory)
    _fields_ = [("auroc", C.c_double), ("ap", C.c_double), ("n_pos", c_int64), ("n_neg", c_int64),
                ("n_distinct", c_int64), ("opt_index", c_int64), ("opt_threshold", c_float), ("pad", c_float)]


class TnProblem(C.Structure):          # == struct acx_tn_problem
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("b_sub", c_void_p), ("M", c_int32), ("N1", c_int32),
                ("N2", c_int32), ("lda", c_int32), ("ldb", c_int32), ("reserved", c_int32)]


class PrepSeg(C.Structure):            # == struct acx_prep_seg
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", c_int32), ("cols", c_int32), ("src_ld", c_int32),
                ("dst_ld", c_int32), ("transpose", c_int32), ("reserved", c_int32)]


class VitDesc(C.Structure):
    _fields_ = [(n, c_int32) for n in ("resolution", "patch", "width", "layers", "heads", "embed_dim", "prec")]


_SIGS = {
    "acx_version": (C.c_int, []),
    "acx_create": (C.c_int, [C.POINTER(c_void_p), C.c_int]),
    "acx_destroy": (None, [c_void_p]),
    "acx_last_error": (C.c_char_p, [c_void_p]),
    "acx_set_option": (C.c_int, [c_void_p, c_int32, c_int64]),
    "acx_gemm": (C.c_int, [c_void_p, C.POINTER(GemmDesc), c_void_p]),
    "acx_layernorm": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                c_int64, c_int32, c_float, c_int32, c_void_p]),
    "acx_attention": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                c_int32, c_void_p]),
    "acx_attention_bf16": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "acx_attention_x3": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "acx_attention_p3": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "acx_attention_p3n": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acx_attention_x3_panel": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "acx_attention_cls": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "acx_vit_patches": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acx_vit_embed": (C.c_int, [c_void_p] * 7 + [c_int32, c_int32, c_int32, c_void_p]),
    "acx_vit_workspace_bytes": (c_size_t, [C.POINTER(VitDesc), c_int32]),
    "acx_vit_encode": (C.c_int, [c_void_p, C.POINTER(VitDesc), C.POINTER(VitWeights), c_void_p, c_int32, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "acx_transformer_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "acx_transformer_workspace_bytes_prec": (c_size_t, [c_int32, c_int32, c_int32]),
    "acx_transformer_forward": (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_int32, C.POINTER(BlockWeights), c_void_p, c_size_t, c_void_p]),
    "acx_text_directions": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "acx_selector_project": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                       c_void_p]),
    "acx_probe_mfma": (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, C.POINTER(C.c_double), c_void_p]),
    "acx_probe_copy": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "acx_split_bf16x3_multi": (C.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "acx_probe_read": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "acx_comm_unique_id": (C.c_int, [c_void_p, c_size_t]),
    "acx_comm_init": (C.c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "acx_comm_destroy": (C.c_int, [c_void_p]),
    "acx_comm_info": (C.c_int, [c_void_p, C.POINTER(c_int32), C.POINTER(c_int32)]),
    "acx_allreduce": (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "acx_allgather": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "acx_bn_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "acx_bn_combine": (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "acx_bn_stats": (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "acx_selector_project_stats": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "acx_selector_bn": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32,
                                  c_float, c_void_p]),
    "acx_axial_attention": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                      c_int32, c_void_p]),
    "acx_cls_head": (C.c_int, [c_void_p] * 8 + [c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acx_cls_head_tiles": (C.c_int, [c_void_p] * 8 + [c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "acx_class_probs": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "acx_prompt_embed": (C.c_int, [c_void_p] * 6 + [c_int32] * 6 + [c_void_p]),
    "acx_gather_rows": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "acx_add_bcast": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "acx_concat_features": (C.c_int, [c_void_p] * 5 + [c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "acx_prof_enable": (C.c_int, [c_void_p, C.c_int]),
    "acx_prof_gemm_flops": (C.c_int, [c_void_p, C.POINTER(C.c_double)]),
    "acx_prof_collect": (C.c_int, [c_void_p, C.POINTER(c_int32), C.POINTER(C.c_double)]),
    "acx_prof_gemm_tn": (C.c_int, [c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_int32)]),
    "acx_gemm_tn_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "acx_gemm_tn": (C.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                              c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "acx_reduce_rows": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "acx_layernorm_bwd": (C.c_int, [c_void_p] * 6 + [c_int64, c_int32, c_float, c_int32, c_float, c_void_p, c_void_p]),
    "acx_cls_head_bwd": (C.c_int, [c_void_p] * 10 + [c_int64, c_int32, c_void_p]),
    "acx_act": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "acx_leaky_grad_planes": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "acx_add": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "acx_transpose": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "acx_conv_weight_dx": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "acx_seq_attention_bwd": (C.c_int, [c_void_p] * 4 + [c_int32] * 7 + [c_void_p, c_void_p]),
    "acx_pos_grad": (C.c_int, [c_void_p] * 5 + [c_int32] * 4 + [c_void_p]),
    "acx_bn_bwd_stats": (C.c_int, [c_void_p] * 4 + [c_int64, c_int32, c_void_p, c_size_t, c_void_p]),
    "acx_bn_bwd_apply": (C.c_int, [c_void_p] * 6 + [c_int32, c_int64, c_int64, c_int32, c_float, c_void_p, c_void_p]),
    "acx_axpby": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float, c_void_p]),
    "acx_colsum_partials": (C.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "acx_text_directions_bwd": (C.c_int, [c_void_p] * 5 + [c_int32] * 3 + [c_void_p]),
    "acx_select_idx": (C.c_int, [c_void_p] * 7 + [c_int32] * 7 + [c_void_p]),
    "acx_gather_segments": (C.c_int, [c_void_p] * 4 + [c_int32] * 5 + [c_void_p]),
    "acx_scatter_segments": (C.c_int, [c_void_p] * 4 + [c_int32] * 5 + [c_void_p]),
    "acx_mil_loss": (C.c_int, [c_void_p] * 13 + [c_size_t] + [c_int32] * 6 + [c_void_p, c_void_p, c_void_p]),
    "acx_mil_loss_one": (C.c_int, [c_void_p] * 13 + [c_size_t] + [c_int32] * 6 + [c_void_p, c_void_p, c_void_p, c_void_p]),
    "acx_mil_loss_bn": (C.c_int, [c_void_p] * 14 + [c_size_t, c_void_p, c_size_t] + [c_int32] * 6 + [c_void_p, c_void_p, c_void_p, c_void_p]),
    "acx_selector_tail": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32] + [c_void_p] * 7 + [c_float, c_float, c_void_p, c_int64] +
                          [c_void_p] * 6 + [c_int32] * 7 + [c_float, c_void_p]),
    "acx_text_directions_bwd_parts": (C.c_int, [c_void_p] * 4 + [c_int32, c_int64, c_void_p] + [c_int32] * 3 + [c_void_p]),
    "acx_adamw": (C.c_int, [c_void_p] * 5 + [c_int64] + [c_float] * 5 + [c_int32, c_void_p]),
    "acx_multi_axpy": (C.c_int, [c_void_p, c_int32, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_int64), c_float, c_void_p]),
    "acx_adamw_multi": (C.c_int, [c_void_p, c_int32] + [C.POINTER(c_void_p)] * 4 + [C.POINTER(c_int64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, c_int32, c_void_p]),
    "acx_row_parts": (c_int64, [c_int64]),
    "acx_prep_multi": (C.c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "acx_multi_copy": (C.c_int, [c_void_p, c_int32, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_int64), c_void_p]),
    "acx_adamw_hyper": (C.c_int, [c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, c_int32,
                                  C.POINTER(c_float)]),
    "acx_adamw_multi_dev": (C.c_int, [c_void_p, c_int32] + [C.POINTER(c_void_p)] * 4 + [C.POINTER(c_int64), c_void_p, c_float,
                                      C.c_double, C.c_double, C.c_double, c_void_p]),
    "acx_bn_pack": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "acx_bn_running_update": (C.c_int, [c_void_p] * 6 + [c_int32, c_float, c_float, c_void_p]),
    "acx_fill_f32": (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "acx_gemm_tn_group_workspace_bytes": (c_size_t, [c_int32, c_void_p]),
    "acx_gemm_tn_group": (C.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "acx_colsum_fused_part_bytes": (c_size_t, [c_int64, c_int32]),
    "acx_colsum_fused": (C.c_int, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p, c_size_t, c_void_p, c_float, c_void_p]),
    "acx_colsum_fused_group": (C.c_int, [c_void_p, c_int32, C.POINTER(c_void_p), C.POINTER(c_int32), C.POINTER(c_int64),
                                         C.POINTER(c_int32), C.POINTER(c_void_p), C.POINTER(c_void_p), c_void_p, c_int32, c_void_p]),
    "acx_reduce_rows_group": (C.c_int, [c_void_p, c_int32, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_int32),
                                        C.POINTER(c_int32), c_void_p]),
    "acx_gemm_tn_zp": (C.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                 c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_void_p]),
    "acx_split_f16x2": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_float, c_int32, c_void_p]),
    "acx_gemm_tn_parts": (C.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                    c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p, C.POINTER(c_int32), c_void_p]),
    "acx_gemm_tn_x6_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "acx_gemm_tn_x6": (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_int32, c_int32, c_int32,
                                 c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_void_p]),
    "acx_ctx_grad": (C.c_int, [c_void_p] * 3 + [c_int32] * 5 + [c_void_p]),
    "acx_scatter_rows": (C.c_int, [c_void_p] * 4 + [c_int64, c_int32, c_void_p]),
    "acx_preprocess_frames": (C.c_int, [c_void_p] * 6 + [c_int32, c_void_p, c_void_p, c_int32] + [c_int32] * 5 +
                              [C.POINTER(c_float), C.POINTER(c_float), c_void_p]),
    "acx_cast_bf16": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "acx_split_bf16x3": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "acx_split_bf16x3_panel": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "acx_colsum": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "acx_sort_workspace_bytes": (c_int64, [c_int64]),
    "acx_sort_pairs_batched": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                         c_void_p, c_int64, c_void_p]),
    "acx_sort_pairs": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64,
                                 c_void_p]),
    "acx_clf_curve_workspace_bytes": (c_int64, [c_int64]),
    "acx_clf_curve_batched": (C.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, C.POINTER(c_int32),
                                        C.POINTER(c_int32), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "acx_clf_curve": (C.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_int64, c_void_p]),
    "acx_test_counts": (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
}

# entry points added by later translation units register themselves here (name -> signature)
EXTRA_SIGS: dict = {}

_lock = threading.Lock()
_lib = None
_ctxs: dict = {}


class AcxError(RuntimeError):
    pass


def declared_symbols():
    return sorted(set(_SIGS) | set(EXTRA_SIGS))


def lib() -> C.CDLL:
    """Loads (building if the in-tree binary is missing or stale and hipcc is present) libacx.so."""
    global _lib
    with _lock:
        if _lib is None:
            path = os.environ.get("ACX_LIB_PATH") or _build.LIB
            if os.environ.get("ACX_LIB_PATH"):
                pass
            elif not os.path.exists(path) or (os.environ.get("ACX_AUTOBUILD") == "1" and _build._stale()):
                # The prebuilt in-tree binary is used as it is (several ranks may import concurrently and file
                # times do not survive a snapshot copy); rebuilding is explicit: __graft_entry__.build(),
                # `python -m anomalyclip_amd._build`, or ACX_AUTOBUILD=1 for development.
                if not os.path.exists(_build.HIPCC):
                    raise AcxError(f"libacx.so not found at {path} and hipcc is not available to build it; "
                                   f"there is no fallback compute path")
                path = _build.build(verbose=False)
            # torch first: its bundled HIP runtime must be the one libacx.so's libamdhip64 dependency resolves to.  Loaded
            # the other way round (libacx.so, then torch) the process ends up with two HIP runtimes and the one libacx sees
            # reports "0 devices" while torch.cuda.is_available() is true (__graft_entry__.build() followed by smoke()).
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
            try:
                L = C.CDLL(path)
            except OSError as e:
                raise AcxError(f"libacx.so could not be loaded from {path}: {e}. There is no fallback path: "
                               f"build it with `python -m anomalyclip_amd._build`.") from e
            for name, (res, args) in {**_SIGS, **EXTRA_SIGS}.items():
                try:
                    fn = getattr(L, name)
                except AttributeError as e:
                    raise AcxError(f"libacx.so does not export {name}") from e
                fn.restype = res
                fn.argtypes = args
            _lib = L
        return _lib


def ctx(device: int = 0) -> int:
    """One acx context per device."""
    L = lib()
    with _lock:
        if device not in _ctxs:
            h = c_void_p()
            rc = L.acx_create(C.byref(h), device)
            if rc != 0:
                raise AcxError(f"acx_create({device}) failed [{rc}]: {L.acx_last_error(None).decode()}")
            _ctxs[device] = h.value
        return _ctxs[device]


def check(rc: int, handle) -> None:
    if rc != 0:
        raise AcxError(f"libacx error {rc}: {lib().acx_last_error(handle).decode()}")
