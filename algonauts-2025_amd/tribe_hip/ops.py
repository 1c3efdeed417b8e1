"""torch-tensor front end of the C ABI: validates shapes / devices on the host (mirroring
the reference's assert-style checks), hands raw device pointers + the current HIP
stream to libtribe_hip.so.  torch is used for memory, streams and nothing else."""

from __future__ import annotations

import ctypes as C
import typing as tp

import torch

from . import _lib
from ._lib import BF16, F32, F64, AttentionDesc, EncoderDesc, EncoderLayer, GemmDesc, LlamaDesc, LlamaLayer, check, lib

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float64: F64}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _cuda(t: torch.Tensor, dtype: torch.dtype | tuple[torch.dtype, ...] | None, name: str, contiguous: bool = True) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise _lib.TribeHipError(f"{name}: tensor is on {t.device}; the TRIBE hot path runs on the GPU only (no CPU fallback)")
    if dtype is not None:
        ok = t.dtype in dtype if isinstance(dtype, tuple) else t.dtype == dtype
        if not ok:
            raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


# --------------------------------------------------------------------------------------
# workspace: one grow-only byte buffer per device, owned by the torch caching allocator
# --------------------------------------------------------------------------------------
_WS: dict[tuple[int, str], torch.Tensor] = {}


def workspace(nbytes: int, device: torch.device, tag: str = "main") -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


# --------------------------------------------------------------------------------------
# generic MFMA GEMM
# --------------------------------------------------------------------------------------
def gemm_nt(
    a: torch.Tensor, b: torch.Tensor, *, bias: torch.Tensor | None = None, bias_row: bool = False, act: str | None = None,
    res: torch.Tensor | None = None, res_scale: torch.Tensor | None = None, alpha: float = 1.0,
    out_dtype: torch.dtype = torch.float32, out: torch.Tensor | None = None,
    rowadd: torch.Tensor | None = None, rowadd_period: int = 0,
    gadd: torch.Tensor | None = None, gadd_index: torch.Tensor | None = None, gadd_div: int = 0, tile_hint: int = 0,
    split_k: bool = False,
) -> torch.Tensor:
    """out[..., m, n] = epi(alpha * sum_k a[..., m, k] * b[..., n, k]); a, b bf16 with equal leading batch dim (or 2-D).
    split_k: let the launcher share the reduction of a small grid out over several workgroups per tile (tribe_gemm_desc.stream_k);
    gemm_nt.last_split tells whether it did."""
    _cuda(a, torch.bfloat16, "a")
    _cuda(b, torch.bfloat16, "b")
    if a.ndim == 2:
        a3, b3 = a[None], b[None]
    else:
        a3, b3 = a, b
    if a3.ndim != 3 or b3.ndim != 3 or a3.shape[0] != b3.shape[0] or a3.shape[2] != b3.shape[2]:
        raise ValueError(f"gemm_nt: incompatible shapes {tuple(a.shape)} x {tuple(b.shape)}")
    Z, M, K = a3.shape
    N = b3.shape[1]
    n_out = N // 2 if act in ("swiglu", "glu") else N  # SwiGLU / GLU fold adjacent column pairs
    if out is None:
        out = torch.empty((Z, M, n_out) if a.ndim == 3 else (M, n_out), dtype=out_dtype, device=a.device)
    _cuda(out, (torch.float32, torch.bfloat16), "out")
    d = GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, Z, 1
    d.A, d.lda, d.sA1 = a3.data_ptr(), K, M * K
    d.B, d.ldb, d.sB1 = b3.data_ptr(), K, N * K
    d.C, d.ldc, d.sC1 = out.data_ptr(), n_out, M * n_out
    d.c_dtype = _DT[out.dtype]
    d.alpha = alpha
    d.tile_hint = tile_hint
    if bias is not None:
        _cuda(bias, torch.float32, "bias")
        d.bias, d.bias_mode = bias.data_ptr(), (_lib.BIAS_ROW if bias_row else _lib.BIAS_COL)
    d.act = {"gelu": _lib.ACT_GELU, "swiglu": _lib.ACT_SWIGLU, "silu": _lib.ACT_SILU, "glu": _lib.ACT_GLU, None: _lib.ACT_NONE}[act]
    if res is not None:
        _cuda(res, torch.float32, "res")
        d.res, d.ldres, d.sRes1 = res.data_ptr(), N, M * N
    if res_scale is not None:
        d.res_scale = _cuda(res_scale, torch.float32, "res_scale").data_ptr()
    if rowadd is not None:
        _cuda(rowadd, torch.float32, "rowadd")
        d.rowadd, d.ld_rowadd, d.rowadd_period = rowadd.data_ptr(), rowadd.shape[-1], rowadd_period
    if gadd is not None:
        _cuda(gadd, torch.float32, "gadd")
        _cuda(gadd_index, torch.int64, "gadd_index")
        d.gadd, d.gadd_index, d.gadd_div, d.ld_gadd = gadd.data_ptr(), gadd_index.data_ptr(), gadd_div, gadd.shape[-1]
    gemm_nt.last_split = False
    if split_k:
        d.stream_k = 1
        nbytes = lib().tribe_gemm_stream_k_workspace_bytes(C.byref(d))
        gemm_nt.last_split = nbytes > 0
        if nbytes > 0:
            ws = workspace(nbytes, a.device, tag="streamk")
            d.stream_k_ws, d.stream_k_ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    check(lib().tribe_gemm_bf16(C.byref(d), _stream()), "tribe_gemm_bf16")
    return out


gemm_nt.last_split = False


def gemm_tn(at: torch.Tensor, bt: torch.Tensor, *, out_dtype: torch.dtype = torch.float32, alpha: float = 1.0,
            bias: torch.Tensor | None = None, stream_k: bool = False) -> torch.Tensor:
    """out[m, n] = alpha * sum_k at[k, m] * bt[k, n] (+ bias[n]); at bf16 [K, M], bt bf16 [K, N], K % 64 == 0, M and N multiples of 8
    (tribe_gemm_desc.trans_ab: the weight-gradient form dW = dY^T X without explicit transposes).  stream_k: the launcher may cut the
    last partial round of tiles over all CUs (plain f32 product only); gemm_tn.last_split tells whether it did."""
    _cuda(at, torch.bfloat16, "at")
    _cuda(bt, torch.bfloat16, "bt")
    if at.ndim != 2 or bt.ndim != 2 or at.shape[0] != bt.shape[0]:
        raise ValueError(f"gemm_tn: incompatible shapes {tuple(at.shape)} x {tuple(bt.shape)}")
    K, M = at.shape
    N = bt.shape[1]
    out = torch.empty(M, N, dtype=out_dtype, device=at.device)
    d = GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
    d.A, d.lda, d.B, d.ldb = at.data_ptr(), M, bt.data_ptr(), N
    d.C, d.ldc, d.c_dtype, d.alpha, d.trans_ab = out.data_ptr(), N, _DT[out_dtype], alpha, 1
    if bias is not None:
        d.bias, d.bias_mode = _cuda(bias, torch.float32, "bias").data_ptr(), _lib.BIAS_COL
    d.stream_k = int(stream_k)
    nbytes = lib().tribe_gemm_stream_k_workspace_bytes(C.byref(d)) if stream_k else 0
    gemm_tn.last_split = nbytes > 0
    if nbytes > 0:   # parts of split tiles travel through the workspace; a second launch sums them in order
        ws = workspace(nbytes, at.device, tag="streamk")
        d.stream_k_ws, d.stream_k_ws_bytes = ws.data_ptr(), ws.numel() * ws.element_size()
    check(lib().tribe_gemm_bf16(C.byref(d), _stream()), "tribe_gemm_bf16")
    return out


gemm_tn.last_split = False


# --------------------------------------------------------------------------------------
# packing
# --------------------------------------------------------------------------------------
def pack_weight(w: torch.Tensor, rows_pad: int | None = None, cols_pad: int | None = None) -> torch.Tensor:
    """f32 [rows, cols] -> bf16 [rows_pad, cols_pad] zero padded (cols padded to 64 by default: the GEMM's K step)."""
    _cuda(w, torch.float32, "w")
    rows, cols = w.shape
    rows_pad = rows if rows_pad is None else rows_pad
    cols_pad = round_up(cols, 64) if cols_pad is None else cols_pad
    out = torch.empty(rows_pad, cols_pad, dtype=torch.bfloat16, device=w.device)
    check(lib().tribe_pack_weight_bf16(w.data_ptr(), rows, cols, cols, out.data_ptr(), rows_pad, cols_pad, _stream()),
          "tribe_pack_weight_bf16")
    return out


def pack_subject_weights(w: torch.Tensor) -> torch.Tensor:
    """SubjectLayers.weights f32 [S, C, V] -> bf16 [S, V_pad, C_pad] (V_pad % 128 == 0, C_pad % 64 == 0)."""
    _cuda(w, torch.float32, "weights")
    S, Cc, V = w.shape
    V_pad, C_pad = round_up(V, 128), round_up(Cc, 64)
    out = torch.empty(S, V_pad, C_pad, dtype=torch.bfloat16, device=w.device)
    check(lib().tribe_pack_subject_weights(w.data_ptr(), S, Cc, V, out.data_ptr(), V_pad, C_pad, _stream()),
          "tribe_pack_subject_weights")
    return out


def pack_features(feat: torch.Tensor, layer_mean: bool, K_pad: int | None = None) -> torch.Tensor:
    """[B, L, D, T] or [B, D, T] (f32 / bf16 / f64) -> bf16 [B*T, K_pad] (model.py:146-155)."""
    _cuda(feat, (torch.float32, torch.bfloat16, torch.float64), "feat")
    if feat.ndim == 3:
        feat = feat[:, None]
    if feat.ndim != 4:
        raise ValueError(f"pack_features: expected [B, L, D, T] or [B, D, T], got {tuple(feat.shape)}")
    B, L, D, T = feat.shape
    K = D if layer_mean else L * D
    K_pad = round_up(K, 64) if K_pad is None else K_pad
    out = torch.empty(B * T, K_pad, dtype=torch.bfloat16, device=feat.device)
    check(lib().tribe_pack_features(feat.data_ptr(), _DT[feat.dtype], B, L, D, T, int(layer_mean), out.data_ptr(), K_pad, _stream()),
          "tribe_pack_features")
    return out


def projector_fwd(feat_packed: torch.Tensor, T: int, w_packed: torch.Tensor, bias: torch.Tensor | None, n_out: int,
                  x: torch.Tensor, col0: int, accumulate: bool, pos_embed: torch.Tensor | None,
                  subj_embed: torch.Tensor | None, subject_id: torch.Tensor | None) -> None:
    _cuda(feat_packed, torch.bfloat16, "feat_packed")
    _cuda(w_packed, torch.bfloat16, "w_packed")
    _cuda(x, torch.float32, "x")
    BT, K_pad = feat_packed.shape
    if w_packed.shape[1] != K_pad or w_packed.shape[0] < n_out:
        raise ValueError(f"projector_fwd: packed weight {tuple(w_packed.shape)} does not match K_pad={K_pad}, n_out={n_out}")
    if pos_embed is not None and (pos_embed.shape[-1] != x.shape[-1] or pos_embed.shape[-2] < T):
        raise ValueError(f"projector_fwd: time_pos_embed {tuple(pos_embed.shape)} shorter than T={T}")
    check(lib().tribe_projector_fwd(feat_packed.data_ptr(), BT, T, K_pad, w_packed.data_ptr(), _p(bias), n_out, x.data_ptr(),
                                    x.shape[-1], col0, int(accumulate), _p(pos_embed), _p(subj_embed), _p(subject_id), _stream()),
          "tribe_projector_fwd")


def projector_zero_fwd(BT: int, T: int, n_out: int, x: torch.Tensor, col0: int, pos_embed: torch.Tensor | None,
                       subj_embed: torch.Tensor | None, subject_id: torch.Tensor | None) -> None:
    _cuda(x, torch.float32, "x")
    check(lib().tribe_projector_zero_fwd(BT, T, n_out, x.data_ptr(), x.shape[-1], col0, _p(pos_embed), _p(subj_embed),
                                         _p(subject_id), _stream()), "tribe_projector_zero_fwd")


# --------------------------------------------------------------------------------------
# encoder pieces
# --------------------------------------------------------------------------------------
def scalenorm(x: torch.Tensor, g: torch.Tensor, gain_scale: float, eps: float, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    _cuda(x, torch.float32, "x")
    _cuda(g, torch.float32, "g")
    dim = x.shape[-1]
    rows = x.numel() // dim
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib().tribe_scalenorm_fwd(x.data_ptr(), rows, dim, g.data_ptr(), gain_scale, eps, y.data_ptr(), _DT[out_dtype], _stream()),
          "tribe_scalenorm_fwd")
    return y


def rotary_(qkv: torch.Tensor, T: int, heads: int, dim_head: int, rot_dim: int, cos: torch.Tensor, sin: torch.Tensor,
            interleaved: bool, heads_kv: int | None = None) -> torch.Tensor:
    """In-place rotary on the q heads and k heads of a fused [rows, (heads + 2*heads_kv) * dim_head] buffer."""
    _cuda(qkv, torch.bfloat16, "qkv")
    heads_kv = heads if heads_kv is None else heads_kv
    row_stride = (heads + 2 * heads_kv) * dim_head
    rows = qkv.numel() // row_stride
    if cos.shape != (T, rot_dim // 2) or sin.shape != cos.shape:
        raise ValueError(f"rotary_: tables must be [T, rot_dim/2] = {(T, rot_dim // 2)}, got {tuple(cos.shape)}")
    check(lib().tribe_rotary_fwd(qkv.data_ptr(), rows, T, row_stride, heads + heads_kv, dim_head, rot_dim,
                                 _cuda(cos, torch.float32, "cos").data_ptr(), _cuda(sin, torch.float32, "sin").data_ptr(),
                                 int(interleaved), _stream()), "tribe_rotary_fwd")
    return qkv


def attention_gqa(qkv: torch.Tensor, B: int, T: int, heads_q: int, heads_kv: int, dim_head: int, scale: float,
                  causal: bool) -> torch.Tensor:
    """Fused attention on a [B*T, (heads_q + 2*heads_kv) * dim_head] q|k|v buffer (grouped-query, optional causal mask)."""
    _cuda(qkv, torch.bfloat16, "qkv")
    width = (heads_q + 2 * heads_kv) * dim_head
    if qkv.numel() != B * T * width:
        raise ValueError("attention_gqa: qkv has the wrong number of elements")
    out = torch.empty(B * T, heads_q * dim_head, dtype=torch.bfloat16, device=qkv.device)
    d = AttentionDesc()
    base = qkv.data_ptr()
    d.q, d.k, d.v = base, base + 2 * heads_q * dim_head, base + 2 * (heads_q + heads_kv) * dim_head
    d.ld_q = d.ld_k = d.ld_v = width
    d.out, d.ld_out = out.data_ptr(), heads_q * dim_head
    d.B, d.T, d.heads_q, d.heads_kv, d.dim_head, d.causal, d.scale = B, T, heads_q, heads_kv, dim_head, int(causal), scale
    check(lib().tribe_attention_fwd_ex(C.byref(d), _stream()), "tribe_attention_fwd_ex")
    return out


def attention_relative_key(qkv: torch.Tensor, B: int, T: int, heads: int, dim_head: int, scale: float, qe: torch.Tensor,
                           left: int, right: int) -> torch.Tensor:
    """Bidirectional attention with the Wav2Vec-BERT relative_key bias (modeling_wav2vec2_bert.py:308-320):
    score[i, j] = (q_i . k_j + qe[i, h, clamp(j - i, -left, right) + left]) * scale, on a fused [B*T, 3*heads*dim_head] buffer.
    `qe` = q . E^T per query row and head, f32 [B*T, heads, stride >= left + right + 1]."""
    _cuda(qkv, torch.bfloat16, "qkv")
    _cuda(qe, torch.float32, "qe")
    if qkv.numel() != B * T * 3 * heads * dim_head or qe.dim() != 3 or qe.shape[0] != B * T or qe.shape[1] != heads:
        raise ValueError("attention_relative_key: qkv / qe have the wrong shape")
    out = torch.empty(B * T, heads * dim_head, dtype=torch.bfloat16, device=qkv.device)
    d = AttentionDesc()
    base, width = qkv.data_ptr(), 3 * heads * dim_head
    d.q, d.k, d.v = base, base + 2 * heads * dim_head, base + 4 * heads * dim_head
    d.ld_q = d.ld_k = d.ld_v = width
    d.out, d.ld_out = out.data_ptr(), heads * dim_head
    d.B, d.T, d.heads_q, d.heads_kv, d.dim_head, d.causal, d.scale = B, T, heads, heads, dim_head, 0, scale
    d.rel_qe, d.ld_rel_qe, d.rel_stride_h, d.rel_left, d.rel_right = qe.data_ptr(), heads * qe.shape[2], qe.shape[2], left, right
    check(lib().tribe_attention_fwd_ex(C.byref(d), _stream()), "tribe_attention_fwd_ex")
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    _cuda(x, torch.float32, "x")
    _cuda(w, torch.float32, "w")
    dim = x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib().tribe_rmsnorm_fwd(x.data_ptr(), x.numel() // dim, dim, w.data_ptr(), eps, y.data_ptr(), _DT[out_dtype], _stream()),
          "tribe_rmsnorm_fwd")
    return y


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None, eps: float, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    _cuda(x, torch.float32, "x")
    _cuda(w, torch.float32, "w")
    dim = x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib().tribe_layernorm_fwd(x.data_ptr(), x.numel() // dim, dim, w.data_ptr(), _p(b), eps, y.data_ptr(), _DT[out_dtype],
                                    _stream()), "tribe_layernorm_fwd")
    return y


def embedding(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    _cuda(table, (torch.float32, torch.bfloat16), "table")
    _cuda(ids, torch.int64, "ids")
    vocab, dim = table.shape
    x = torch.empty(ids.numel(), dim, dtype=torch.float32, device=table.device)
    check(lib().tribe_embedding_fwd(table.data_ptr(), _DT[table.dtype], ids.data_ptr(), ids.numel(), dim, vocab, x.data_ptr(), _stream()),
          "tribe_embedding_fwd")
    return x


def dwconv_ln_swish(x: torch.Tensor, B: int, T: int, w_kc: torch.Tensor, ln_w: torch.Tensor, ln_b: torch.Tensor, eps: float) -> torch.Tensor:
    """Causal depthwise Conv1d over time (left pad K - 1, no bias) + LayerNorm over channels + swish (Wav2Vec2BertConvolutionModule,
    modeling_wav2vec2_bert.py:214-222): x bf16 [B*T, C], w_kc f32 [K, C] (tap-major) -> bf16 [B*T, C]."""
    _cuda(x, torch.bfloat16, "x")
    _cuda(w_kc, torch.float32, "w_kc")
    K, Cc = w_kc.shape
    if x.shape != (B * T, Cc):
        raise ValueError(f"dwconv_ln_swish: x must be [{B * T}, {Cc}], got {tuple(x.shape)}")
    y = torch.empty_like(x)
    check(lib().tribe_dwconv_ln_swish_fwd(x.data_ptr(), B, T, Cc, K, w_kc.data_ptr(), _cuda(ln_w, torch.float32, "ln_w").data_ptr(),
                                          _cuda(ln_b, torch.float32, "ln_b").data_ptr(), eps, y.data_ptr(), _stream()), "tribe_dwconv_ln_swish_fwd")
    return y


def segment_mean(x: torch.Tensor, B: int, T: int, start: torch.Tensor | None, length: torch.Tensor | None) -> torch.Tensor:
    _cuda(x, torch.float32, "x")
    dim = x.shape[-1]
    out = torch.empty(B, dim, dtype=torch.float32, device=x.device)
    check(lib().tribe_segment_mean_fwd(x.data_ptr(), B, T, dim, _p(start), _p(length), out.data_ptr(), dim, _stream()),
          "tribe_segment_mean_fwd")
    return out


def attention_set_mode(mode: int) -> None:
    """0 = fused kernels, picked per head size and grid (default); 1 = materialised-scores path (cross-check); 2 = the 16-row-per-wave
    kernel at every head size; 3 = dim_head 384 on the key-split kernel; 4 / 5 = dim_head 64 on the 4-wave / the anti-phase 8-wave kernel."""
    check(lib().tribe_attention_set_mode(mode), "tribe_attention_set_mode")


def attention(qkv: torch.Tensor, B: int, T: int, heads: int, dim_head: int, scale: float) -> torch.Tensor:
    _cuda(qkv, torch.bfloat16, "qkv")
    if qkv.numel() != B * T * 3 * heads * dim_head:
        raise ValueError("attention: qkv has the wrong number of elements")
    out = torch.empty(B * T, heads * dim_head, dtype=torch.bfloat16, device=qkv.device)
    nbytes = lib().tribe_attention_workspace_bytes(B, T, heads, dim_head)
    ws = workspace(nbytes, qkv.device)
    check(lib().tribe_attention_fwd(qkv.data_ptr(), B, T, heads, dim_head, scale, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "tribe_attention_fwd")
    return out


def attention_lse_supported(dim_head: int) -> bool:
    """True when the fused bidirectional kernel of this head size can also write the base-2 log-sum-exp of its scores (attention_with_lse)."""
    return bool(lib().tribe_attention_lse_supported(dim_head, 0))


def attention_with_lse(qkv: torch.Tensor, B: int, T: int, heads: int, dim_head: int, scale: float) -> tuple[torch.Tensor, torch.Tensor]:
    """attention() plus lse2 [B, heads, T] f32 = log2 sum_j 2^(q.k_j * scale * log2 e): what a backward pass needs to rebuild
    P = exp2(q.k * scale * log2 e - lse2) inside a GEMM epilogue (ACT_EXP2) instead of materialising f32 scores and a softmax."""
    _cuda(qkv, torch.bfloat16, "qkv")
    inner = heads * dim_head
    if qkv.numel() != B * T * 3 * inner:
        raise ValueError("attention_with_lse: qkv has the wrong number of elements")
    out = torch.empty(B * T, inner, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, heads, T, dtype=torch.float32, device=qkv.device)
    d = AttentionDesc()
    base = qkv.data_ptr()
    d.q, d.k, d.v = base, base + 2 * inner, base + 4 * inner
    d.ld_q = d.ld_k = d.ld_v = 3 * inner
    d.out, d.ld_out = out.data_ptr(), inner
    d.B, d.T, d.heads_q, d.heads_kv, d.dim_head, d.causal, d.scale = B, T, heads, heads, dim_head, 0, scale
    d.lse = lse.data_ptr()
    check(lib().tribe_attention_fwd_ex(C.byref(d), _stream()), "tribe_attention_fwd_ex")
    return out, lse


def rowdot_heads(a: torch.Tensor, b: torch.Tensor, B: int, T: int, heads: int, dim_head: int, scale: float) -> torch.Tensor:
    """out[b, h, t] = scale * sum_d a[b T + t, h dh + d] * b[b T + t, h dh + d] (bf16 [B*T, heads*dim_head] inputs, f32 output)."""
    _cuda(a, torch.bfloat16, "a")
    _cuda(b, torch.bfloat16, "b")
    inner = heads * dim_head
    if a.shape != (B * T, inner) or b.shape != (B * T, inner) or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError(f"rowdot_heads: operands must be contiguous [{B * T}, {inner}]")
    out = torch.empty(B, heads, T, dtype=torch.float32, device=a.device)
    check(lib().tribe_rowdot_heads_bf16(a.data_ptr(), inner, b.data_ptr(), inner, B, T, heads, dim_head, scale, out.data_ptr(), _stream()),
          "tribe_rowdot_heads_bf16")
    return out


class EncoderPack:
    """Device-resident bf16 weights + f32 vectors of one x_transformers-style encoder, laid out for
    tribe_encoder_fwd.  Built from (and kept alive next to) the fp32 master parameters."""

    def __init__(self, dim: int, depth: int, heads: int, dim_head: int, ff_inner: int, rot_dim: int, rotary_interleaved: bool,
                 norm_gain_scale: float, norm_eps: float):
        self.dim, self.depth, self.heads, self.dim_head, self.ff_inner = dim, depth, heads, dim_head, ff_inner
        self.rot_dim, self.rotary_interleaved = rot_dim, rotary_interleaved
        self.norm_gain_scale, self.norm_eps = norm_gain_scale, norm_eps
        self.layers = (EncoderLayer * max(depth, 1))()
        self.keep: list[torch.Tensor] = []  # owns every tensor the pointer table refers to
        self.final_norm_g: torch.Tensor | None = None
        self._tabs: dict[tuple[int, int], tuple[torch.Tensor, torch.Tensor]] = {}
        self.inv_freq: torch.Tensor | None = None

    def tables(self, T: int, device: torch.device) -> tuple[torch.Tensor | None, torch.Tensor | None]:
        if self.rot_dim == 0:
            return None, None
        key = (T, device.index or 0)
        if key not in self._tabs:
            t = torch.arange(T, device=device, dtype=torch.float32)
            freqs = torch.einsum("i,j->ij", t, self.inv_freq.to(device=device, dtype=torch.float32))
            self._tabs[key] = (freqs.cos().contiguous(), freqs.sin().contiguous())
        return self._tabs[key]


def encoder_fwd(x: torch.Tensor, pack: EncoderPack, B: int, T: int, out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """x f32 [B*T, dim] is consumed (updated in place); returns final-normed y [B*T, dim]."""
    _cuda(x, torch.float32, "x")
    if x.shape != (B * T, pack.dim):
        raise ValueError(f"encoder_fwd: x must be [{B * T}, {pack.dim}], got {tuple(x.shape)}")
    cos, sin = pack.tables(T, x.device)
    d = EncoderDesc()
    d.B, d.T = B, T
    d.dim, d.depth, d.heads, d.dim_head, d.ff_inner = pack.dim, pack.depth, pack.heads, pack.dim_head, pack.ff_inner
    d.rot_dim, d.rotary_interleaved = pack.rot_dim, int(pack.rotary_interleaved)
    d.norm_gain_scale, d.norm_eps = pack.norm_gain_scale, pack.norm_eps
    d.layers_host = C.cast(pack.layers, C.POINTER(EncoderLayer))
    d.final_norm_g = pack.final_norm_g.data_ptr()
    d.cos_tab, d.sin_tab = _p(cos), _p(sin)
    y = torch.empty(B * T, pack.dim, dtype=out_dtype, device=x.device)
    nbytes = lib().tribe_encoder_workspace_bytes(C.byref(d))
    ws = workspace(nbytes, x.device)
    check(lib().tribe_encoder_fwd(C.byref(d), x.data_ptr(), y.data_ptr(), _DT[out_dtype], ws.data_ptr(), ws.numel(), _stream()),
          "tribe_encoder_fwd")
    return y


# --------------------------------------------------------------------------------------
# voxel head, pool, losses
# --------------------------------------------------------------------------------------
def voxel_head(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor | None, subjects: torch.Tensor, V: int) -> torch.Tensor:
    """x bf16 [B, T, C_pad]; returns f32 [B, V, T] (SubjectLayers.forward, common.py:45-67)."""
    _cuda(x, torch.bfloat16, "x")
    _cuda(w_packed, torch.bfloat16, "w_packed")
    _cuda(subjects, torch.int64, "subjects")
    B, T, C_pad = x.shape
    S, V_pad, C_pad2 = w_packed.shape
    if C_pad != C_pad2 or subjects.numel() != B:
        raise ValueError(f"voxel_head: x {tuple(x.shape)} / weights {tuple(w_packed.shape)} / subjects {tuple(subjects.shape)} mismatch")
    y = torch.empty(B, V, T, dtype=torch.float32, device=x.device)
    check(lib().tribe_voxel_head_fwd(x.data_ptr(), B, T, C_pad, w_packed.data_ptr(), S, V, V_pad, _p(bias), subjects.data_ptr(),
                                     y.data_ptr(), _stream()), "tribe_voxel_head_fwd")
    return y


def adaptive_avg_pool(x: torch.Tensor, t_out: int) -> torch.Tensor:
    _cuda(x, torch.float32, "x")
    t_in = x.shape[-1]
    rows = x.numel() // t_in
    y = torch.empty(*x.shape[:-1], t_out, dtype=torch.float32, device=x.device)
    check(lib().tribe_adaptive_avg_pool_fwd(x.data_ptr(), rows, t_in, y.data_ptr(), t_out, _stream()), "tribe_adaptive_avg_pool_fwd")
    return y


def mse(pred: torch.Tensor, true: torch.Tensor) -> torch.Tensor:
    _cuda(pred, torch.float32, "pred")
    _cuda(true, torch.float32, "true")
    if pred.shape != true.shape:
        raise ValueError(f"mse: shape mismatch {tuple(pred.shape)} vs {tuple(true.shape)}")
    n = pred.numel()
    out = torch.empty((), dtype=torch.float32, device=pred.device)
    ws = workspace(lib().tribe_mse_workspace_bytes(n), pred.device, "loss")
    check(lib().tribe_mse_fwd(pred.data_ptr(), true.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "tribe_mse_fwd")
    return out


def _bvt_strides(pred: torch.Tensor, true: torch.Tensor, what: str) -> tuple[int, int, int, int, int, int]:
    """Accept any strided 3-D [B, V, T] view (e.g. the transpose of a '(b t) d' matrix) -- no copy is made."""
    if pred.shape != true.shape or pred.ndim != 3:
        raise ValueError(f"{what}: expected equal [B, V, T] shapes, got {tuple(pred.shape)} / {tuple(true.shape)}")
    if pred.stride() != true.stride():
        raise ValueError(f"{what}: pred and true must share strides, got {pred.stride()} / {true.stride()}")
    B, V, T = pred.shape
    sb, sv, st = pred.stride()
    return B, V, T, sb, sv, st


def pearson_stats_update(stats: torch.Tensor, pred: torch.Tensor, true: torch.Tensor, group: torch.Tensor | None = None) -> None:
    """stats f64 [G, V, 6] += sufficient statistics of pred/true [B, V, T]; group int64 [B] or None."""
    _cuda(stats, torch.float64, "stats")
    _cuda(pred, torch.float32, "pred", contiguous=False)
    _cuda(true, torch.float32, "true", contiguous=False)
    B, V, T, sb, sv, st = _bvt_strides(pred, true, "pearson_stats_update")
    G = stats.shape[0]
    if stats.shape != (G, V, 6):
        raise ValueError(f"pearson_stats_update: stats must be [G, {V}, 6], got {tuple(stats.shape)}")
    if group is not None:
        _cuda(group, torch.int64, "group")
        if group.numel() != B:
            raise ValueError("pearson_stats_update: group must have B entries")
    check(lib().tribe_pearson_stats_update(pred.data_ptr(), true.data_ptr(), B, V, T, sb, sv, st, _p(group), G, stats.data_ptr(),
                                           _stream()), "tribe_pearson_stats_update")


def pearson_from_stats(stats: torch.Tensor) -> torch.Tensor:
    _cuda(stats, torch.float64, "stats")
    G, V, _ = stats.shape
    r = torch.empty(G, V, dtype=torch.float32, device=stats.device)
    check(lib().tribe_pearson_from_stats(stats.data_ptr(), G, V, r.data_ptr(), _stream()), "tribe_pearson_from_stats")
    return r


def pearson_loss(pred: torch.Tensor, true: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    """PearsonLoss over the '(b t) d' view of [B, V, T] tensors (losses.py:17-42); any strided view is accepted."""
    _cuda(pred, torch.float32, "pred", contiguous=False)
    _cuda(true, torch.float32, "true", contiguous=False)
    if reduction not in ("mean", "sum"):
        raise ValueError(f"Invalid reduction: {reduction}")
    B, V, T, sb, sv, st = _bvt_strides(pred, true, "pearson_loss")
    out = torch.empty((), dtype=torch.float32, device=pred.device)
    ws = workspace(lib().tribe_pearson_loss_workspace_bytes(V), pred.device, "loss")
    check(lib().tribe_pearson_loss_fwd(pred.data_ptr(), true.data_ptr(), B, V, T, sb, sv, st, int(reduction == "sum"),
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "tribe_pearson_loss_fwd")
    return out


# --------------------------------------------------------------------------------------
# measurement hook: HIP events around every GEMM launch, summed per operator role
# --------------------------------------------------------------------------------------
def prof_begin(max_records: int = 4096) -> None:
    check(lib().tribe_prof_begin(max_records), "tribe_prof_begin")


def prof_end() -> dict[str, dict[str, float]]:
    n = len(_lib.ROLES)
    ms, cnt, fl = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)()
    check(lib().tribe_prof_end(n, ms, cnt, fl), "tribe_prof_end")
    return {role: {"ms": ms[i], "launches": int(cnt[i]), "flops": fl[i]} for i, role in enumerate(_lib.ROLES) if cnt[i]}


__all__ = [n for n in dir() if not n.startswith("_")]
_ = tp


# --------------------------------------------------------------------------------------
# segment assembly from HBM-resident extractor outputs (csrc/features.hip)
# --------------------------------------------------------------------------------------
def group_mean(states: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    """states f32 [batch, n_states, plane...] -> f32 [batch, n_groups, plane...]: mean over layers lo[g]..hi[g]-1
    (`_aggregate_layers`, text.py:129-149).  lo / hi int32 device tensors."""
    _cuda(states, torch.float32, "states")
    _cuda(lo, torch.int32, "lo")
    _cuda(hi, torch.int32, "hi")
    if states.ndim < 3 or lo.shape != hi.shape or lo.ndim != 1:
        raise ValueError(f"group_mean: states {tuple(states.shape)} / bounds {tuple(lo.shape)}, {tuple(hi.shape)}")
    batch, n_states = states.shape[:2]
    plane = states[0, 0].numel()
    out = torch.empty((batch, lo.numel()) + tuple(states.shape[2:]), dtype=torch.float32, device=states.device)
    check(lib().tribe_group_mean_fwd(states.data_ptr(), batch, n_states, plane, lo.data_ptr(), hi.data_ptr(), lo.numel(), out.data_ptr(),
                                     _stream()), "tribe_group_mean_fwd")
    return out


def segment_gather(pieces: torch.Tensor, seg_ptr: torch.Tensor, B: int, C: int, T: int, packed: bool, C_pad: int | None = None) -> torch.Tensor:
    """Sum time slices of cached arrays into segment outputs.  pieces: uint8 device tensor holding `tribe_feature_piece`
    records, seg_ptr int32 [B + 1].  packed -> bf16 [B*T, C_pad]; else f32 [B, C, T]."""
    _cuda(pieces, torch.uint8, "pieces")
    _cuda(seg_ptr, torch.int32, "seg_ptr")
    if seg_ptr.numel() != B + 1 or pieces.numel() % _lib.FEATURE_PIECE_DTYPE.itemsize:
        raise ValueError("segment_gather: seg_ptr must hold B + 1 offsets and pieces whole records")
    if packed:
        C_pad = round_up(C, 64) if C_pad is None else C_pad
        out = torch.empty(B * T, C_pad, dtype=torch.bfloat16, device=pieces.device)
    else:
        C_pad = C
        out = torch.empty(B, C, T, dtype=torch.float32, device=pieces.device)
    check(lib().tribe_segment_gather_fwd(pieces.data_ptr(), seg_ptr.data_ptr(), B, C, T, out.data_ptr(), BF16 if packed else F32, C_pad,
                                         _stream()), "tribe_segment_gather_fwd")
    return out


def word_bag(table: torch.Tensor, row_ptr: torch.Tensor, word_idx: torch.Tensor, rows: int, C_pad: int | None = None,
             f32: bool = False) -> torch.Tensor:
    """table f32 [n_words, C]; CSR (row_ptr int32 [rows + 1], word_idx int32) -> bf16 [rows, C_pad] row sums
    (f32=True: the unrounded sums, f32 [rows, C])."""
    _cuda(table, torch.float32, "table")
    _cuda(row_ptr, torch.int32, "row_ptr")
    _cuda(word_idx, torch.int32, "word_idx")
    if table.ndim != 2 or row_ptr.numel() != rows + 1:
        raise ValueError(f"word_bag: table {tuple(table.shape)}, row_ptr {tuple(row_ptr.shape)} for {rows} rows")
    n_words, C = table.shape
    if f32:
        out = torch.empty(rows, C, dtype=torch.float32, device=table.device)
        if word_idx.numel() == 0:
            return out.zero_()
        check(lib().tribe_word_bag_f32_fwd(table.data_ptr(), n_words, C, row_ptr.data_ptr(), word_idx.data_ptr(), rows, out.data_ptr(),
                                           _stream()), "tribe_word_bag_f32_fwd")
        return out
    C_pad = round_up(C, 64) if C_pad is None else C_pad
    out = torch.empty(rows, C_pad, dtype=torch.bfloat16, device=table.device)
    if word_idx.numel() == 0:   # no word overlaps any row: the output is all zeros (and the kernel would get a null list)
        return out.zero_()
    check(lib().tribe_word_bag_fwd(table.data_ptr(), n_words, C, row_ptr.data_ptr(), word_idx.data_ptr(), rows, out.data_ptr(), C_pad,
                                   _stream()), "tribe_word_bag_fwd")
    return out


# --------------------------------------------------------------------------------------
# after the model: submission rows, ensemble averaging (csrc/features.hip)
# --------------------------------------------------------------------------------------
def transpose_f32(x: torch.Tensor) -> torch.Tensor:
    """f32 [Z, R, C] -> [Z, C, R] (predictions [B, V, T'] -> [B, T', V], callbacks.py:63-64)."""
    _cuda(x, torch.float32, "x")
    if x.ndim != 3:
        raise ValueError(f"transpose_f32: expected [Z, R, C], got {tuple(x.shape)}")
    Z, R, Cc = x.shape
    out = torch.empty(Z, Cc, R, dtype=torch.float32, device=x.device)
    check(lib().tribe_transpose_f32_fwd(x.data_ptr(), Z, R, Cc, out.data_ptr(), _stream()), "tribe_transpose_f32_fwd")
    return out


def weighted_sum(preds: torch.Tensor, w_column: torch.Tensor | None = None, w_set: torch.Tensor | None = None) -> torch.Tensor:
    """preds f32 [N, ..., V] -> sum over N of preds * w: w_column f32 [N, V] (f32 result) or w_set f64 [N] (f64 result)."""
    _cuda(preds, torch.float32, "preds")
    N, V = preds.shape[0], preds.shape[-1]
    M = preds[0].numel()
    if (w_column is None) == (w_set is None):
        raise ValueError("weighted_sum: give exactly one of w_column / w_set")
    if w_column is not None:
        _cuda(w_column, torch.float32, "w_column")
        if tuple(w_column.shape) != (N, V):
            raise ValueError(f"weighted_sum: w_column {tuple(w_column.shape)} != ({N}, {V})")
        out = torch.empty(preds.shape[1:], dtype=torch.float32, device=preds.device)
    else:
        _cuda(w_set, torch.float64, "w_set")
        if tuple(w_set.shape) != (N,):
            raise ValueError(f"weighted_sum: w_set {tuple(w_set.shape)} != ({N},)")
        out = torch.empty(preds.shape[1:], dtype=torch.float64, device=preds.device)
    check(lib().tribe_weighted_sum_fwd(preds.data_ptr(), N, M, V, _p(w_column), _p(w_set), out.data_ptr(), _stream()), "tribe_weighted_sum_fwd")
    return out


def corr_matrix(x: torch.Tensor) -> torch.Tensor:
    """np.corrcoef over the rows of x f32 [N, K] -> f64 [N, N]."""
    _cuda(x, torch.float32, "x")
    if x.ndim != 2:
        raise ValueError(f"corr_matrix: expected [N, K], got {tuple(x.shape)}")
    N, K = x.shape
    out = torch.empty(N, N, dtype=torch.float64, device=x.device)
    nbytes = lib().tribe_corr_matrix_workspace_bytes(N)
    ws = workspace(nbytes, x.device, tag="corr")
    check(lib().tribe_corr_matrix_fwd(x.data_ptr(), N, K, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "tribe_corr_matrix_fwd")
    return out


# --------------------------------------------------------------------------------------
# fp8 (e4m3) GEMM + per-tensor quantisation (csrc/gemm_fp8.hip)
# --------------------------------------------------------------------------------------
FP8_MAX = 448.0


def absmax(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """max |x| of a 2-D f32 / bf16 tensor as a device float (accumulates into `out` when given)."""
    _cuda(x, (torch.float32, torch.bfloat16), "x")
    x2 = x.reshape(-1, x.shape[-1])
    acc = out is not None
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(lib().tribe_absmax_fwd(x2.data_ptr(), _DT[x.dtype], x2.shape[0], x2.shape[1], x2.shape[1], out.data_ptr(), int(acc), _stream()),
          "tribe_absmax_fwd")
    return out


def quantize_fp8(x: torch.Tensor, scale: float, K_pad: int | None = None) -> torch.Tensor:
    """x f32 / bf16 [M, K] -> uint8 [M, K_pad] holding e4m3(clamp(x / scale)); K_pad defaults to K rounded up to 128."""
    _cuda(x, (torch.float32, torch.bfloat16), "x")
    if x.ndim != 2 or not scale > 0:
        raise ValueError(f"quantize_fp8: expected a 2-D tensor and a positive scale, got {tuple(x.shape)}, {scale}")
    M, K = x.shape
    K_pad = round_up(K, 128) if K_pad is None else K_pad
    out = torch.empty(M, K_pad, dtype=torch.uint8, device=x.device)
    check(lib().tribe_quantize_fp8_fwd(x.data_ptr(), _DT[x.dtype], M, K, K, 1.0 / scale, out.data_ptr(), K_pad, _stream()),
          "tribe_quantize_fp8_fwd")
    return out


def norm_quantize_fp8(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None, eps: float, scale: float, layernorm: bool) -> torch.Tensor:
    """quantize_fp8(rmsnorm(x) | layernorm(x) in bf16, scale) in one pass: x f32 [..., dim] -> uint8 [rows, dim] of e4m3 bytes."""
    _cuda(x, torch.float32, "x")
    _cuda(w, torch.float32, "w")
    if not scale > 0:
        raise ValueError(f"norm_quantize_fp8: scale must be positive, got {scale}")
    dim = x.shape[-1]
    rows = x.numel() // dim
    out = torch.empty(rows, dim, dtype=torch.uint8, device=x.device)
    check(lib().tribe_norm_quantize_fp8_fwd(x.data_ptr(), rows, dim, w.data_ptr(), _p(b), int(layernorm), eps, 1.0 / scale, out.data_ptr(),
                                            _stream()), "tribe_norm_quantize_fp8_fwd")
    return out


def gemm_fp8_nt(a: torch.Tensor, b: torch.Tensor, alpha: float, *, bias: torch.Tensor | None = None, act: str | None = None,
                res: torch.Tensor | None = None, out_dtype: torch.dtype = torch.float32, out: torch.Tensor | None = None) -> torch.Tensor:
    """out[m, n] = epi(alpha * sum_k a[m, k] * b[n, k]) with a, b uint8 tensors of e4m3 bytes ([M, K], [N, K], K % 128 == 0)."""
    _cuda(a, torch.uint8, "a")
    _cuda(b, torch.uint8, "b")
    if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
        raise ValueError(f"gemm_fp8_nt: incompatible shapes {tuple(a.shape)} x {tuple(b.shape)}")
    M, K = a.shape
    N = b.shape[0]
    n_out = N // 2 if act in ("swiglu", "glu") else N
    if out is None:
        out = torch.empty(M, n_out, dtype=out_dtype, device=a.device)
    _cuda(out, (torch.float32, torch.bfloat16), "out")
    d = GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch0 = M, N, K, 1, 1
    d.A, d.lda, d.B, d.ldb = a.data_ptr(), K, b.data_ptr(), K
    d.C, d.ldc, d.c_dtype, d.alpha = out.data_ptr(), n_out, _DT[out.dtype], alpha
    if bias is not None:
        d.bias, d.bias_mode = _cuda(bias, torch.float32, "bias").data_ptr(), _lib.BIAS_COL
    d.act = {"gelu": _lib.ACT_GELU, "swiglu": _lib.ACT_SWIGLU, "silu": _lib.ACT_SILU, "glu": _lib.ACT_GLU, None: _lib.ACT_NONE}[act]
    if res is not None:
        d.res, d.ldres = _cuda(res, torch.float32, "res").data_ptr(), N
    check(lib().tribe_gemm_fp8(C.byref(d), _stream()), "tribe_gemm_fp8")
    return out
