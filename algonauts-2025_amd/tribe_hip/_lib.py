"""ctypes binding of libtribe_hip.so (C ABI declared in include/tribe_hip.h).

The shared object is built in-tree by `__graft_entry__.build()` / `make -C csrc`.
There is NO fallback: if the library is missing or a call fails, this module
raises -- the product path never silently drops to PyTorch or to the oracle.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as _np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libtribe_hip.so"

F32, BF16, F64 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_SWIGLU, ACT_SILU, ACT_GLU, ACT_GELU_BWD, ACT_EXP2, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5, 6, 7
BIAS_NONE, BIAS_COL, BIAS_ROW = 0, 1, 2
ROLES = ["generic", "projector", "qkv", "attn_scores", "attn_pv", "out_proj", "ff1", "ff2", "voxel_head", "attention"]

i64, i32, f32, vp, sz = C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_size_t


# tribe_feature_piece as a numpy record (tables of pieces are built vectorised on the host and uploaded as bytes)
ADAM_TENSOR_DTYPE = _np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("p_bf16", "<u8")], align=True)   # tribe_adam_tensor
FEATURE_PIECE_DTYPE = _np.dtype([("src", "<u8"), ("ld", "<i8"), ("src_first", "<i4"), ("src_count", "<i4"), ("dst_first", "<i4"),
                                 ("dst_count", "<i4")], align=True)


class GemmDesc(C.Structure):
    """struct tribe_gemm_desc"""

    _fields_ = [
        ("M", i64), ("N", i64), ("K", i64),
        ("batch1", i64), ("batch0", i64),
        ("A", vp), ("lda", i64), ("sA1", i64), ("sA0", i64),
        ("B", vp), ("ldb", i64), ("sB1", i64), ("sB0", i64),
        ("C", vp), ("ldc", i64), ("sC1", i64), ("sC0", i64),
        ("c_dtype", i32), ("alpha", f32),
        ("gather1", vp), ("gather_a", i32), ("gather_bias", i32),
        ("bias", vp), ("bias_mode", i32), ("sBias1", i64),
        ("act", i32),
        ("res", vp), ("ldres", i64), ("sRes1", i64), ("sRes0", i64),
        ("res_scale", vp),
        ("rowadd", vp), ("ld_rowadd", i64), ("rowadd_period", i64),
        ("gadd", vp), ("gadd_index", vp), ("gadd_div", i64), ("ld_gadd", i64),
        ("aux", vp), ("ld_aux", i64), ("gather_b", i32),
        ("role", i32), ("tile_hint", i32), ("trans_ab", i32),
        ("c_bf16", vp), ("ld_c_bf16", i64), ("row_sumsq", vp), ("ld_row_sumsq", i64), ("row_scale", vp),
        ("sBias0", i64),
        ("stream_k", i32), ("reserved0", i32), ("stream_k_ws", vp), ("stream_k_ws_bytes", i64),
    ]


class AttentionDesc(C.Structure):
    """struct tribe_attention_desc"""

    _fields_ = [
        ("q", vp), ("k", vp), ("v", vp), ("ld_q", i64), ("ld_k", i64), ("ld_v", i64),
        ("out", vp), ("ld_out", i64), ("B", i64), ("T", i64),
        ("heads_q", i32), ("heads_kv", i32), ("dim_head", i32), ("causal", i32), ("scale", f32),
        ("rel_qe", vp), ("ld_rel_qe", i64), ("rel_stride_h", i32), ("rel_left", i32), ("rel_right", i32),
        ("lse", vp),
    ]


class LlamaLayer(C.Structure):
    """struct tribe_llama_layer"""

    _fields_ = [("input_norm_w", vp), ("w_qkv", vp), ("w_o", vp), ("post_norm_w", vp), ("w_gate_up", vp), ("w_down", vp)]


class LlamaFp8Layer(C.Structure):
    """struct tribe_llama_fp8_layer"""

    _fields_ = [("w_qkv", vp), ("w_o", vp), ("w_gate_up", vp), ("w_down", vp), ("w_scale", f32 * 4), ("in_scale", f32 * 4)]


class LlamaDesc(C.Structure):
    """struct tribe_llama_desc"""

    _fields_ = [
        ("B", i64), ("T", i64),
        ("dim", i32), ("depth", i32), ("heads_q", i32), ("heads_kv", i32), ("dim_head", i32), ("inter", i32),
        ("rms_eps", f32),
        ("embed", vp), ("embed_dtype", i32), ("vocab", i64),
        ("layers_host", C.POINTER(LlamaLayer)),
        ("final_norm_w", vp), ("cos_tab", vp), ("sin_tab", vp),
        ("ids", vp), ("pool_start", vp), ("pool_len", vp),
        ("fp8_host", C.POINTER(LlamaFp8Layer)), ("amax_out", vp),
    ]


class VitLayer(C.Structure):
    """struct tribe_vit_layer"""

    _fields_ = [(n, vp) for n in ("norm1_w", "norm1_b", "w_qkv", "b_qkv", "w_proj", "b_proj", "norm2_w", "norm2_b", "w_fc1", "b_fc1",
                                  "w_fc2", "b_fc2")]


class VitFp8Layer(C.Structure):
    """struct tribe_vit_fp8_layer"""

    _fields_ = [("w_qkv", vp), ("w_proj", vp), ("w_fc1", vp), ("w_fc2", vp), ("w_scale", f32 * 4), ("in_scale", f32 * 4)]


class Vjepa2Desc(C.Structure):
    """struct tribe_vjepa2_desc"""

    _fields_ = [
        ("B", i64),
        ("frames", i32), ("chans", i32), ("height", i32), ("width", i32), ("tubelet", i32), ("patch", i32),
        ("dim", i32), ("depth", i32), ("heads", i32), ("dim_head", i32), ("mlp", i32),
        ("ln_eps", f32),
        ("w_patch", vp), ("b_patch", vp), ("K_pad", i64),
        ("layers_host", C.POINTER(VitLayer)),
        ("cos_tab", vp), ("sin_tab", vp), ("pixels", vp),
        ("fp8_host", C.POINTER(VitFp8Layer)), ("amax_out", vp),
    ]


class ConformerLayer(C.Structure):
    """struct tribe_conformer_layer"""

    _fields_ = [(n, vp) for n in (
        "ffn1_ln_w", "ffn1_ln_b", "w_ffn1_in", "b_ffn1_in", "w_ffn1_out", "b_ffn1_out_half",
        "attn_ln_w", "attn_ln_b", "w_qkv", "b_qkv", "dist_emb", "w_attn_out", "b_attn_out",
        "conv_ln_w", "conv_ln_b", "w_pw1", "w_dw_kc", "dw_ln_w", "dw_ln_b", "w_pw2",
        "ffn2_ln_w", "ffn2_ln_b", "w_ffn2_in", "b_ffn2_in", "w_ffn2_out", "b_ffn2_out_half", "final_ln_w", "final_ln_b")]


class ConformerFp8Layer(C.Structure):
    """struct tribe_conformer_fp8_layer"""

    _fields_ = [("w_ffn1_in", vp), ("w_ffn1_out", vp), ("w_ffn2_in", vp), ("w_ffn2_out", vp), ("w_scale", f32 * 4), ("in_scale", f32 * 4)]


class W2vBertDesc(C.Structure):
    """struct tribe_w2vbert_desc"""

    _fields_ = [
        ("B", i64), ("T", i64), ("feat_dim", i32), ("feat_pad", i32),
        ("dim", i32), ("depth", i32), ("heads", i32), ("dim_head", i32), ("inter", i32), ("conv_kernel", i32),
        ("rel_left", i32), ("rel_right", i32), ("ln_eps", f32),
        ("fp_ln_w", vp), ("fp_ln_b", vp), ("w_fp", vp), ("b_fp", vp),
        ("layers_host", C.POINTER(ConformerLayer)),
        ("features", vp), ("out_index", vp), ("n_out", i64),
        ("fp8_host", C.POINTER(ConformerFp8Layer)), ("amax_out", vp),
    ]


class EncoderLayer(C.Structure):
    """struct tribe_encoder_layer"""

    _fields_ = [
        ("attn_norm_g", vp), ("w_qkv", vp), ("w_out", vp), ("attn_res_scale", vp),
        ("ff_norm_g", vp), ("w_ff1", vp), ("b_ff1", vp), ("w_ff2", vp), ("b_ff2", vp), ("ff_res_scale", vp),
    ]


class EncoderDesc(C.Structure):
    """struct tribe_encoder_desc"""

    _fields_ = [
        ("B", i64), ("T", i64),
        ("dim", i32), ("depth", i32), ("heads", i32), ("dim_head", i32), ("ff_inner", i32), ("rot_dim", i32),
        ("rotary_interleaved", i32),
        ("norm_gain_scale", f32), ("norm_eps", f32),
        ("layers_host", C.POINTER(EncoderLayer)),
        ("final_norm_g", vp),
        ("cos_tab", vp), ("sin_tab", vp),
    ]


ABI_VERSION = 5
# (struct name, mirror) in the order of tribe_abi_struct_sizes(): compared with the library's sizeof() at load time
STRUCT_MIRRORS = [
    ("tribe_gemm_desc", GemmDesc), ("tribe_attention_desc", AttentionDesc), ("tribe_encoder_layer", EncoderLayer), ("tribe_encoder_desc", EncoderDesc),
    ("tribe_vit_layer", VitLayer), ("tribe_vit_fp8_layer", VitFp8Layer), ("tribe_vjepa2_desc", Vjepa2Desc), ("tribe_conformer_layer", ConformerLayer),
    ("tribe_conformer_fp8_layer", ConformerFp8Layer), ("tribe_w2vbert_desc", W2vBertDesc), ("tribe_llama_layer", LlamaLayer),
    ("tribe_llama_fp8_layer", LlamaFp8Layer), ("tribe_llama_desc", LlamaDesc), ("tribe_feature_piece", FEATURE_PIECE_DTYPE),
    ("tribe_adam_tensor", ADAM_TENSOR_DTYPE),
]

# name -> (restype, argtypes); must cover every symbol include/tribe_hip.h declares
SIGNATURES = {
    "tribe_version": (C.c_int, []),
    "tribe_abi_struct_sizes": (C.c_int, [C.POINTER(i64), i32]),
    "tribe_last_error": (C.c_char_p, []),
    "tribe_gemm_bf16": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "tribe_gemm_sumsq_slots": (C.c_int, [C.POINTER(GemmDesc)]),
    "tribe_gemm_stream_k_workspace_bytes": (i64, [C.POINTER(GemmDesc)]),
    "tribe_gemm_stream_k_plan": (C.c_int, [i32, i32, C.POINTER(i32)]),
    "tribe_prof_begin": (C.c_int, [i32]),
    "tribe_prof_end": (C.c_int, [i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double)]),
    "tribe_pack_weight_bf16": (C.c_int, [vp, i64, i64, i64, vp, i64, i64, vp]),
    "tribe_pack_subject_weights": (C.c_int, [vp, i64, i64, i64, vp, i64, i64, vp]),
    "tribe_pack_features": (C.c_int, [vp, i32, i64, i64, i64, i64, i32, vp, i64, vp]),
    "tribe_projector_fwd": (C.c_int, [vp, i64, i64, i64, vp, vp, i64, vp, i64, i64, i32, vp, vp, vp, vp]),
    "tribe_projector_zero_fwd": (C.c_int, [i64, i64, i64, vp, i64, i64, vp, vp, vp, vp]),
    "tribe_scalenorm_fwd": (C.c_int, [vp, i64, i64, vp, f32, f32, vp, i32, vp]),
    "tribe_rotary_fwd": (C.c_int, [vp, i64, i64, i64, i32, i32, i32, vp, vp, i32, vp]),
    "tribe_attention_fwd_ex": (C.c_int, [C.POINTER(AttentionDesc), vp]),
    "tribe_embedding_fwd": (C.c_int, [vp, i32, vp, i64, i64, i64, vp, vp]),
    "tribe_rmsnorm_fwd": (C.c_int, [vp, i64, i64, vp, f32, vp, i32, vp]),
    "tribe_layernorm_fwd": (C.c_int, [vp, i64, i64, vp, vp, f32, vp, i32, vp]),
    "tribe_segment_mean_fwd": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, i64, vp]),
    "tribe_im2col3d_fwd": (C.c_int, [vp, i64, i32, i32, i32, i32, i32, i32, vp, i64, vp]),
    "tribe_vjepa2_workspace_bytes": (sz, [C.POINTER(Vjepa2Desc)]),
    "tribe_vjepa2_fwd": (C.c_int, [C.POINTER(Vjepa2Desc), vp, vp, sz, vp]),
    "tribe_dwconv_ln_swish_fwd": (C.c_int, [vp, i64, i64, i32, i32, vp, vp, vp, f32, vp, vp]),
    "tribe_gather_rows_fwd": (C.c_int, [vp, i64, i64, i64, vp, i64, vp, vp]),
    "tribe_w2vbert_workspace_bytes": (sz, [C.POINTER(W2vBertDesc)]),
    "tribe_w2vbert_fwd": (C.c_int, [C.POINTER(W2vBertDesc), vp, vp, sz, vp]),
    "tribe_transpose_bf16": (C.c_int, [vp, i32, i64, i64, i64, i64, i64, vp, i64, i64, vp]),
    "tribe_transpose_bf16_b2": (C.c_int, [vp, i32, i64, i64, i64, i64, i64, i64, i64, vp, i64, i64, vp]),
    "tribe_colsum_fwd": (C.c_int, [vp, i32, vp, i64, i64, i64, vp, i32, vp]),
    "tribe_scalenorm_bwd": (C.c_int, [vp, vp, i32, vp, f32, f32, i64, i64, vp, vp, vp, vp, vp]),
    "tribe_softmax_bwd": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, f32, vp, i64, vp]),
    "tribe_softmax_fwd": (C.c_int, [vp, i64, i64, i64, vp, i64, i64, vp]),
    "tribe_mse_bwd": (C.c_int, [vp, vp, i64, vp, vp, vp]),
    "tribe_adaptive_avg_pool_bwd": (C.c_int, [vp, i64, i64, i64, vp, vp]),
    "tribe_rowsum_scatter": (C.c_int, [vp, i64, i64, i64, vp, vp, vp]),
    "tribe_slab_scatter_sum": (C.c_int, [vp, i64, i64, vp, vp, vp]),
    "tribe_colsum_cast_fwd": (C.c_int, [vp, vp, i64, i64, i64, vp, vp, vp, i64, vp]),
    "tribe_scale_cols_fwd": (C.c_int, [vp, vp, i64, i64, vp, vp]),
    "tribe_pearson_loss_bwd": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, i64, vp, i32, vp, vp, vp]),
    "tribe_lse_rows_fwd": (C.c_int, [vp, i64, i64, vp, vp, vp]),
    "tribe_infonce_dlogits": (C.c_int, [vp, i64, i64, vp, vp, vp, vp, i64, vp]),
    "tribe_cast_bf16_fwd": (C.c_int, [vp, i64, vp, vp]),
    "tribe_group_mean_fwd": (C.c_int, [vp, i64, i64, i64, vp, vp, i32, vp, vp]),
    "tribe_segment_gather_fwd": (C.c_int, [vp, vp, i64, i64, i64, vp, i32, i64, vp]),
    "tribe_word_bag_fwd": (C.c_int, [vp, i64, i64, vp, vp, i64, vp, i64, vp]),
    "tribe_word_bag_f32_fwd": (C.c_int, [vp, i64, i64, vp, vp, i64, vp, vp]),
    "tribe_transpose_f32_fwd": (C.c_int, [vp, i64, i64, i64, vp, vp]),
    "tribe_gemm_fp8": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "tribe_rownorm_scale_fwd": (C.c_int, [vp, i64, i64, vp, f32, f32, vp, vp]),
    "tribe_rowdot_heads_bf16": (C.c_int, [vp, i64, vp, i64, i64, i64, i32, i32, f32, vp, vp]),
    "tribe_adam_chunk_elems": (i64, []),
    "tribe_adam_step": (C.c_int, [vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, i32, vp]),
    "tribe_swa_update": (C.c_int, [vp, vp, vp, i64, f32, vp]),
    "tribe_quantize_fp8_fwd": (C.c_int, [vp, i32, i64, i64, i64, f32, vp, i64, vp]),
    "tribe_norm_quantize_fp8_fwd": (C.c_int, [vp, i64, i64, vp, vp, i32, f32, f32, vp, vp]),
    "tribe_absmax_fwd": (C.c_int, [vp, i32, i64, i64, i64, vp, i32, vp]),
    "tribe_weighted_sum_fwd": (C.c_int, [vp, i64, i64, i64, vp, vp, vp, vp]),
    "tribe_corr_matrix_workspace_bytes": (sz, [i64]),
    "tribe_corr_matrix_fwd": (C.c_int, [vp, i64, i64, vp, vp, sz, vp]),
    "tribe_llama_workspace_bytes": (sz, [C.POINTER(LlamaDesc)]),
    "tribe_llama_fwd": (C.c_int, [C.POINTER(LlamaDesc), vp, vp, sz, vp]),
    "tribe_attention_workspace_bytes": (sz, [i64, i64, i32, i32]),
    "tribe_attention_lse_supported": (C.c_int, [i32, i32]),
    "tribe_attention_set_mode": (C.c_int, [i32]),
    "tribe_attention_fwd": (C.c_int, [vp, i64, i64, i32, i32, f32, vp, vp, sz, vp]),
    "tribe_encoder_workspace_bytes": (sz, [C.POINTER(EncoderDesc)]),
    "tribe_encoder_fwd": (C.c_int, [C.POINTER(EncoderDesc), vp, vp, i32, vp, sz, vp]),
    "tribe_voxel_head_fwd": (C.c_int, [vp, i64, i64, i64, vp, i64, i64, i64, vp, vp, vp, vp]),
    "tribe_adaptive_avg_pool_fwd": (C.c_int, [vp, i64, i64, vp, i64, vp]),
    "tribe_mse_fwd": (C.c_int, [vp, vp, i64, vp, vp, sz, vp]),
    "tribe_mse_workspace_bytes": (sz, [i64]),
    "tribe_pearson_stats_update": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, i64, vp, i64, vp, vp]),
    "tribe_pearson_from_stats": (C.c_int, [vp, i64, i64, vp, vp]),
    "tribe_pearson_loss_fwd": (C.c_int, [vp, vp, i64, i64, i64, i64, i64, i64, i32, vp, vp, sz, vp]),
    "tribe_pearson_loss_workspace_bytes": (sz, [i64]),
}

_lib = None


class TribeHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the shared object once; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("TRIBE_HIP_LIB", LIB_PATH))
        if not path.exists():
            raise TribeHipError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C algonauts-2025_amd/csrc`). There is no CPU / PyTorch fallback for the hot path."
            )
        handle = C.CDLL(str(path))
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the export is missing
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.tribe_version() != ABI_VERSION:
            raise TribeHipError(f"ABI version mismatch: library reports {handle.tribe_version()}, binding expects {ABI_VERSION}")
        got = (i64 * len(STRUCT_MIRRORS))()
        n = handle.tribe_abi_struct_sizes(got, len(STRUCT_MIRRORS))
        want = [m.itemsize if isinstance(m, _np.dtype) else C.sizeof(m) for _, m in STRUCT_MIRRORS]
        if n != len(want) or list(got) != want:
            bad = [f"{name}: library {g} / binding {w}" for (name, _), g, w in zip(STRUCT_MIRRORS, got, want) if g != w]
            raise TribeHipError(f"descriptor layout mismatch between {path.name} and its ctypes mirrors ({n} structs): " + "; ".join(bad))
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    """0 = OK; <0 argument error -> ValueError; >0 hipError_t -> TribeHipError."""
    if rc == 0:
        return
    msg = lib().tribe_last_error().decode(errors="replace")
    if rc < 0:
        raise ValueError(f"{what}: {msg}")
    raise TribeHipError(f"{what}: HIP error {rc}: {msg}")
