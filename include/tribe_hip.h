/*
 * tribe_hip.h -- C ABI of libtribe_hip.so: the MI355X (gfx950) native trimodal
 * fMRI-encode hot path of TRIBE (vovw/algonauts-2025).
 *
 * This is the drop-in boundary: plain pointers and sizes only, no torch types.
 * Every entry point below replaces one piece of the reference's Python/PyTorch
 * hot path; the reference line range it replaces is cited (paths relative to
 * the reference checkout).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless a name ends in _host.
 *  - The caller owns every buffer including the workspace; the library never
 *    allocates or frees device memory and keeps no pointer past return.
 *  - Every call is asynchronous on the `hipStream_t` passed as `void* stream`
 *    (torch: torch.cuda.current_stream().cuda_stream).
 *  - Return value: 0 = OK; < 0 = argument / shape / alignment error (message
 *    via tribe_last_error()); > 0 = hipError_t from the launch.  No exception
 *    crosses the ABI.
 *  - "bf16" buffers are uint16_t bit patterns (round-to-nearest-even from f32).
 *  - Row-major everywhere; "ld*" are leading dimensions in ELEMENTS.
 */
#ifndef TRIBE_HIP_H
#define TRIBE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRIBE_ABI_VERSION 5   /* bumped whenever a descriptor struct changes layout (round 2 changed three without a bump) */

enum tribe_dtype { TRIBE_F32 = 0, TRIBE_BF16 = 1, TRIBE_F64 = 2 };
enum tribe_act {
  TRIBE_ACT_NONE = 0,
  TRIBE_ACT_GELU = 1,   /* exact (erf) GELU */
  TRIBE_ACT_SWIGLU = 2, /* columns come in (gate, up) pairs: out[:, j] = silu(v[:, 2j]) * v[:, 2j+1]; C has N/2 columns */
  TRIBE_ACT_SILU = 3,
  TRIBE_ACT_GLU = 4,    /* (a, b) column pairs: out[:, j] = v[:, 2j] * sigmoid(v[:, 2j+1]); C has N/2 columns (nn.GLU) */
  TRIBE_ACT_GELU_BWD = 5, /* backward of GELU: out = v * gelu'(aux[m][n]), aux = saved pre-activation (bf16) */
  /* attention backward without [B, h, T, T] f32 tensors (round 3): with alpha = scale * log2(e) and a ROW bias of -lse2 (the forward kernel's
   * base-2 log-sum-exp) the score GEMM writes P = exp2(alpha * q.k - lse2[row]) directly ... */
  TRIBE_ACT_EXP2 = 6,
  /* ... and with alpha = scale and a row bias of -scale * D[row] (D = rowsum(dO * O)) the dP GEMM writes dS = (alpha * dO.v + bias[row]) * aux[m][n],
   * aux = P (bf16, same layout AND batch strides as C) */
  TRIBE_ACT_MUL_AUX = 7
};
enum tribe_bias_mode { TRIBE_BIAS_NONE = 0, TRIBE_BIAS_COL = 1, TRIBE_BIAS_ROW = 2 };
/* which operator of the path a GEMM launch serves: selects the kernel SYMBOL (per-operator rows in
 * rocprofv3 --stats and in the tribe_prof_* event profile); arithmetic is identical for all roles */
enum tribe_gemm_role {
  TRIBE_ROLE_GENERIC = 0, TRIBE_ROLE_PROJECTOR = 1, TRIBE_ROLE_QKV = 2, TRIBE_ROLE_ATTN_SCORES = 3, TRIBE_ROLE_ATTN_PV = 4,
  TRIBE_ROLE_OUT_PROJ = 5, TRIBE_ROLE_FF1 = 6, TRIBE_ROLE_FF2 = 7, TRIBE_ROLE_VOXEL_HEAD = 8,
  TRIBE_ROLE_ATTENTION = 9, /* the fused attention launches (not a GEMM role: profile slot only; flops = 4 * T * dim_head per query row and head) */
  TRIBE_ROLE_COUNT = 10
};

int tribe_version(void);
/* sizeof() of every descriptor struct of this header, in declaration order (gemm_desc, attention_desc, encoder_layer, encoder_desc,
 * vit_layer, vit_fp8_layer, vjepa2_desc, conformer_layer, conformer_fp8_layer, w2vbert_desc, llama_layer, llama_fp8_layer, llama_desc,
 * feature_piece, adam_tensor): a binding compares them with its own mirrors at load time, so that a stale library or a stale mirror
 * fails loudly instead of mis-striding a table.  Writes min(n, 15) entries, returns 15. */
int tribe_abi_struct_sizes(int64_t* sizes, int32_t n);
/* thread-local, valid until the next failing call on this thread */
const char* tribe_last_error(void);

/* ------------------------------------------------------------------------- *
 * Generic MFMA GEMM (the workhorse every dense contraction below lowers to)
 *   C[b1][b0][m][n] = epi( alpha * sum_k A[b1][b0][m][k] * B[b1][b0][n][k] )
 * A, B bf16 with K contiguous ("NT" form == nn.Linear's [out,in] weights).
 * K must be a multiple of 64 (callers zero-pad K); M, N arbitrary.
 * epi(v) = act(v + bias) + res * res_scale + rowadd[m % period] + gadd[gidx[m / div]]
 * Replaces torch.nn.functional.linear / torch.einsum / torch.bmm call sites of
 * model.py:157, common.py:64 and the x_transformers Attention / FeedForward.
 * ------------------------------------------------------------------------- */
typedef struct tribe_gemm_desc {
  int64_t M, N, K;
  int64_t batch1, batch0; /* grid z = batch1 * batch0; b1 = z / batch0, b0 = z % batch0 */
  const void* A; int64_t lda, sA1, sA0;
  const void* B; int64_t ldb, sB1, sB0;
  void* C; int64_t ldc, sC1, sC0;
  int32_t c_dtype;         /* TRIBE_F32 or TRIBE_BF16 */
  float alpha;
  const int64_t* gather1;  /* optional [batch1]: index replacing b1 for A (gather_a) and bias (gather_bias) */
  int32_t gather_a, gather_bias;
  const float* bias; int32_t bias_mode; int64_t sBias1; /* f32; per-col [N] or per-row [M]; + b1' * sBias1 */
  int32_t act;
  const float* res; int64_t ldres, sRes1, sRes0; /* f32 residual, may alias C when c_dtype == F32 */
  const float* res_scale;  /* f32 [N] or NULL (= 1) */
  const float* rowadd; int64_t ld_rowadd, rowadd_period; /* + rowadd[(m % period)][n] */
  const float* gadd; const int64_t* gadd_index; int64_t gadd_div, ld_gadd; /* + gadd[gadd_index[m / div]][n] */
  /* aux bf16 [M, N] (ld_aux): with act == GELU the PRE-activation is also stored there (training forward);
   * with act == GELU_BWD it is read.  NULL = unused. */
  void* aux; int64_t ld_aux;
  int32_t gather_b;        /* gather1 also replaces b1 for the B operand */
  int32_t role;            /* enum tribe_gemm_role */
  int32_t tile_hint;       /* 0 = automatic; tests / tuning: 1 = 128x128 double-buffered, 2 = 256x256, 3 = 128x128 ring, 4 = 256x192, 5 = 256x256 one-wave-per-SIMD */
  /* 1 = the operands are given TRANSPOSED: A is At [K, M] (lda >= M), B is Bt [K, N] (ldb >= N), C[m][n] = sum_k At[k][m] Bt[k][n] --
   * the weight gradient dW = dY^T X straight from the row-major dY [tokens, N_out] and X [tokens, K_in] the forward produced, without
   * the explicit transposes (torch.nn.functional.linear's backward in the reference).  M, N multiples of 8; plain epilogue only. */
  int32_t trans_ab;
  /* ScaleNorm folded into the GEMMs either side of it (x_transformers pre-norm: y = W . (x * s_m), s_m = g sqrt(d) / |x_m|):
   * the PRODUCER of x (f32 C) also emits a bf16 copy and per-row partial sums of squares, one slot per wave column group
   * (row_sumsq[m * ld_row_sumsq + n / w], w = 64 or 48 columns by the tile the launch gets: tribe_gemm_sumsq_slots() says how
   * many slots per row THIS descriptor will write -- pass it on as n_partial); tribe_rownorm_scale_fwd turns them into s_m; the CONSUMER multiplies its
   * accumulator rows by row_scale[m] before bias / activation (the scaling commutes with the product).  Un-batched
   * launches whose N is a multiple of the tile and whose operands are 16-byte aligned only; NULL = unused. */
  uint16_t* c_bf16; int64_t ld_c_bf16;
  float* row_sumsq; int64_t ld_row_sumsq;
  const float* row_scale;
  int64_t sBias0;          /* + b0 * sBias0 on the bias pointer (a ROW bias per (b1, b0) batch: the attention backward's -lse2 / -scale * D); 0 = none */
  /* 1 = the launcher may share out the REDUCTION of a launch whose output tiles do not fill the chip over more workgroups; the parts travel
   * through stream_k_ws and a second launch on the same stream sums them in a fixed order (bit-reproducible) and writes C.
   *   trans_ab (the weight-gradient form: dW of a 3072 x 3072 layer is 144 tiles on 256 CUs): stream-K -- the tiles of the last, partial round
   *     are cut into equal runs of K-steps over all CUs.  Plain f32 product only (no bias / activation / residual / batch).
   *   otherwise (NT form): split-K of grids of at most half a round of 128 x 128 tiles (M = 128 rows: BASELINE config 1) -- 2..8 workgroups
   *     per tile; the second launch applies alpha, row_scale, bias, GELU, the scaled residual or periodic row add, c_bf16 and row_sumsq.
   * The workspace must hold tribe_gemm_stream_k_workspace_bytes(desc) bytes (0 = this launch is not split), 16-byte aligned, and is free
   * again when the launch is done. */
  int32_t stream_k; int32_t reserved0;
  void* stream_k_ws; int64_t stream_k_ws_bytes;
} tribe_gemm_desc;

int tribe_gemm_bf16(const tribe_gemm_desc* desc, void* stream);
/* number of row_sumsq slots per row the launch of `desc` writes (N / 64, or N / 48 when the launch gets 256 x 192 tiles); < 0 on a bad
 * descriptor.  Depends on M, N, K, the batch counts, tile_hint and the fused-norm operands only. */
int tribe_gemm_sumsq_slots(const tribe_gemm_desc* desc);
/* bytes of desc->stream_k_ws this launch needs: 0 when it would not be split (stream_k clear, not trans_ab, or no partial round worth cutting) */
int64_t tribe_gemm_stream_k_workspace_bytes(const tribe_gemm_desc* desc);
/* The stream-K schedule of a launch with `tiles` output tiles of `nk` K-steps on this device (host-only, for inspection and tests):
 * out[0..3] = {tiles computed whole, K-steps in the split region, K-steps per run, runs}, out[4 .. 259] = the order in which the runs' second
 * parts are queued.  Returns the grid size (== tiles when nothing is split, and then out[0] >= tiles). */
int tribe_gemm_stream_k_plan(int32_t tiles, int32_t nk, int32_t* out);
/* scale[m] = g[0] * gain_scale / max(sqrt(sum_p partial[m, p]), eps): the ScaleNorm factor from the partial sums of squares a
 * GEMM epilogue left in row_sumsq (partial f32 [rows, n_partial], row-major). */
int tribe_rownorm_scale_fwd(const float* partial, int64_t rows, int64_t n_partial, const float* g, float gain_scale, float eps,
                            float* scale, void* stream);

/* D[row][h] = sum_d a[row][h * dim_head + d] * b[row][h * dim_head + d] * scale (bf16 inputs [rows, heads * dim_head], row strides in elements,
 * f32 output [B, heads, T] with row = b * T + t): the rowsum(dO * O) term of the attention backward, written negated and pre-scaled as the row bias
 * of the dS GEMM wants it when scale < 0. */
int tribe_rowdot_heads_bf16(const uint16_t* a, int64_t ld_a, const uint16_t* b, int64_t ld_b, int64_t B, int64_t T, int32_t heads, int32_t dim_head,
                            float scale, float* out, void* stream);

/* Measurement hook (bench.py): while enabled, every GEMM launch is bracketed by HIP events on ITS
 * stream.  tribe_prof_end synchronises them and returns, per role, the summed kernel time (ms), the
 * launch count and the summed algorithmic FLOPs (2*M*N*K*batches).  Host arrays of >= TRIBE_ROLE_COUNT.
 * PROCESS-WIDE state (one mutex-guarded record list): the compute entry points are re-entrant and keep no state, but this
 * diagnostic pair, tribe_attention_set_mode and the one-time hipFuncSetAttribute flags of the kernels are per process, not
 * per device or per thread -- in line with the deployment model (one process per GPU, main.py:388-395).  Call begin / end
 * from the thread that launches. */
int tribe_prof_begin(int32_t max_records);
int tribe_prof_end(int32_t n_roles, double* total_ms_host, int64_t* count_host, double* flops_host);

/* ------------------------------------------------------------------------- *
 * Packing (one-time per parameter version; fp32 master weights stay in torch)
 * ------------------------------------------------------------------------- */
/* f32 [rows, cols] (ld = ld_src) -> bf16 [rows_pad, cols_pad], zero padded */
int tribe_pack_weight_bf16(const float* src, int64_t rows, int64_t cols, int64_t ld_src,
                           uint16_t* dst, int64_t rows_pad, int64_t cols_pad, void* stream);
/* SubjectLayers weights (common.py:26) f32 [S, C, V] -> bf16 [S, V_pad, C_pad] (transposed, zero padded) */
int tribe_pack_subject_weights(const float* w, int64_t S, int64_t C, int64_t V,
                               uint16_t* dst, int64_t V_pad, int64_t C_pad, void* stream);

/* ------------------------------------------------------------------------- *
 * a3: FmriEncoder.aggregate_features prologue (model.py:146-155)
 *   feat [B, L, D, T] (f32 / bf16 / f64) -> bf16 [B*T, K_pad]
 *   layer_mean == 0: "b l d t -> b t (l d)"; == 1: mean over l then "b d t -> b t d"
 * ------------------------------------------------------------------------- */
int tribe_pack_features(const void* feat, int32_t dtype, int64_t B, int64_t L, int64_t D, int64_t T,
                        int32_t layer_mean, uint16_t* dst, int64_t K_pad, void* stream);

/* a3/a4: one modality's projector nn.Linear (model.py:157) writing its column
 * slice of the fused [B*T, hidden] f32 stream (torch.cat, model.py:161-162) and
 * adding time_pos_embed[:, :T] (+ subject_embed[subject_id]) (model.py:169-172).
 * accumulate != 0 adds into x (feature_aggregation == "sum", model.py:163-164). */
int tribe_projector_fwd(const uint16_t* feat_packed, int64_t BT, int64_t T, int64_t K_pad,
                        const uint16_t* w_packed /*[N_out, K_pad]*/, const float* bias /*[N_out]*/, int64_t N_out,
                        float* x, int64_t hidden, int64_t col0, int32_t accumulate,
                        const float* pos_embed /*[>=T, hidden] or NULL*/,
                        const float* subj_embed /*[S, hidden] or NULL*/, const int64_t* subject_id /*[B]*/,
                        void* stream);
/* zero-filled block for a modality without projector (model.py:143-144) incl. the pos/subject adds */
int tribe_projector_zero_fwd(int64_t BT, int64_t T, int64_t N_out, float* x, int64_t hidden, int64_t col0,
                             const float* pos_embed, const float* subj_embed, const int64_t* subject_id,
                             void* stream);

/* ------------------------------------------------------------------------- *
 * a6: x_transformers.Encoder (called at model.py:173)
 * ------------------------------------------------------------------------- */
/* ScaleNorm: y = x / max(||x||_2, eps) * gain_scale * g[0]  (g read on device) */
int tribe_scalenorm_fwd(const float* x, int64_t rows, int64_t dim, const float* g, float gain_scale, float eps,
                        void* y, int32_t y_dtype, void* stream);
/* (partial) rotary embedding, in place, on the first n_heads heads of every row of a bf16 buffer with row stride
 * row_stride elements (a fused qkv buffer: q heads followed by k heads).  Position of row r is r % T.
 * cos/sin: f32 [T, rot_dim/2]; interleaved != 0 pairs (2i,2i+1) (x_transformers 2.x), else (i, i+rot_dim/2)
 * (x_transformers 1.27 and HF rotate_half, modeling_llama.py). */
/* interleaved == 2: per-ELEMENT tables f32 [T, rot_dim] with (2i, 2i+1) pairs -- out[2i] = x[2i] C[2i] - x[2i+1] S[2i],
 * out[2i+1] = x[2i+1] C[2i+1] + x[2i] S[2i+1] (VJEPA2RopeAttention.apply_rotary_embeddings, modeling_vjepa2.py:180-294) */
int tribe_rotary_fwd(uint16_t* x, int64_t rows, int64_t T, int64_t row_stride, int32_t n_heads, int32_t dim_head,
                     int32_t rot_dim, const float* cos_tab, const float* sin_tab, int32_t interleaved, void* stream);
/* softmax(q k^T * scale) v for all (batch, head); qkv as above; out bf16 [rows, heads*dim_head] */
size_t tribe_attention_workspace_bytes(int64_t B, int64_t T, int32_t heads, int32_t dim_head);
/* 0 (default): fused flash-style kernel for dim_head in {64,128,192,384}, else the materialised path;
 * 1: always materialise scores (QK^T GEMM -> f32 softmax -> PV GEMM), kept as a cross-check;
 * 2: fused, but dim_head 384 and 64 on the 16-query-row kernel of the other head sizes instead of their own kernels
 *    (A/B measurements);
 * 3: fused, dim_head 384 on the key-split kernel (two waves per SIMD, each owning half of the head dimension);
 * 4: fused, dim_head 64 (bidirectional) on the 64-rows-per-wave kernel in phase-separated order (the default interleaves the two row
 *    tiles of a wave); 5: dim_head 64 on the anti-phase 8-wave kernel.  Modes 2-5 exist for A/B measurements (csrc/attention_d64.hip).
 * Process-wide switch: set it from one thread, between launches. */
int tribe_attention_set_mode(int32_t mode);
int tribe_attention_fwd(const uint16_t* qkv, int64_t B, int64_t T, int32_t heads, int32_t dim_head, float scale,
                        uint16_t* out, void* workspace, size_t workspace_bytes, void* stream);
/* general form of the fused kernel: separate q / k / v views (row = b*T + t, row strides in elements, ld_k == ld_v),
 * grouped-query attention (heads_q % heads_kv == 0; query head h reads kv head h / (heads_q/heads_kv)) and an
 * optional causal mask (key <= query).  dim_head in {64, 128, 192, 384}.  out[row][h*dim_head + d]. */
typedef struct tribe_attention_desc {
  const uint16_t* q; const uint16_t* k; const uint16_t* v;
  int64_t ld_q, ld_k, ld_v;
  uint16_t* out; int64_t ld_out;
  int64_t B, T;
  int32_t heads_q, heads_kv, dim_head, causal;
  float scale;
  /* optional "relative_key" position bias (Wav2Vec2BertSelfAttention, modeling_wav2vec2_bert.py:308-320):
   * score[i][j] += rel_qe[b*T + i][h * rel_stride_h + clamp(j - i, -rel_left, rel_right) + rel_left], where
   * rel_qe = q . distance_embedding^T (f32, computed by a GEMM beforehand).  NULL = none.  dim_head 64 only. */
  const float* rel_qe; int64_t ld_rel_qe; int32_t rel_stride_h, rel_left, rel_right;
  /* optional: base-2 log-sum-exp of every query's scaled scores, f32 [B, heads_q, T]: lse2 = max + log2(sum 2^(s * scale * log2 e - max)), so
   * that P = exp2(s * scale * log2 e - lse2) -- what a backward pass needs to recompute P without a softmax.  dim_head 384, bidirectional,
   * default mode only (the one-wave-per-SIMD kernel): tribe_attention_fwd_ex fails otherwise.  NULL = not wanted. */
  float* lse;
} tribe_attention_desc;
int tribe_attention_fwd_ex(const tribe_attention_desc* desc, void* stream);
/* 1 when tribe_attention_fwd_ex can write desc->lse for this head size under the current tribe_attention_set_mode */
int tribe_attention_lse_supported(int32_t dim_head, int32_t causal);

typedef struct tribe_encoder_layer {
  /* attention block */
  const float* attn_norm_g;       /* [1] */
  const uint16_t* w_qkv;          /* bf16 [3*inner, dim]  rows: q | k | v */
  const uint16_t* w_out;          /* bf16 [dim, inner] */
  const float* attn_res_scale;    /* [dim] or NULL */
  /* feed-forward block */
  const float* ff_norm_g;         /* [1] */
  const uint16_t* w_ff1;          /* bf16 [ff_inner, dim] */
  const float* b_ff1;             /* [ff_inner] */
  const uint16_t* w_ff2;          /* bf16 [dim, ff_inner] */
  const float* b_ff2;             /* [dim] */
  const float* ff_res_scale;      /* [dim] or NULL */
} tribe_encoder_layer;

typedef struct tribe_encoder_desc {
  int64_t B, T;
  int32_t dim, depth, heads, dim_head, ff_inner, rot_dim, rotary_interleaved;
  float norm_gain_scale, norm_eps;   /* ScaleNorm flavour */
  const tribe_encoder_layer* layers_host; /* HOST array [depth] of device pointers */
  const float* final_norm_g;         /* [1] */
  const float* cos_tab; const float* sin_tab; /* f32 [T, rot_dim/2] or NULL when rot_dim == 0 */
} tribe_encoder_desc;

size_t tribe_encoder_workspace_bytes(const tribe_encoder_desc* d);
/* x: f32 [B*T, dim] residual stream, updated in place; y: final-normed output (bf16 or f32) [B*T, dim] */
int tribe_encoder_fwd(const tribe_encoder_desc* d, float* x, void* y, int32_t y_dtype,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * a16-a18: building blocks of the frozen extractors (HF transformers architectures)
 * ------------------------------------------------------------------------- */
/* x[i, :] = table[ids[i], :]  (nn.Embedding; table f32 or bf16 -> x f32) */
int tribe_embedding_fwd(const void* table, int32_t table_dtype, const int64_t* ids, int64_t n, int64_t dim, int64_t vocab,
                        float* x, void* stream);
/* LlamaRMSNorm (modeling_llama.py:51-66): y = x * rsqrt(mean(x^2) + eps) * w */
int tribe_rmsnorm_fwd(const float* x, int64_t rows, int64_t dim, const float* w, float eps, void* y, int32_t y_dtype,
                      void* stream);
/* nn.LayerNorm over the last axis: y = (x - mean) * rsqrt(var + eps) * w + b */
int tribe_layernorm_fwd(const float* x, int64_t rows, int64_t dim, const float* w, const float* b, float eps, void* y,
                        int32_t y_dtype, void* stream);
/* out[b, :] = mean_{t in [start[b], start[b]+len[b])} x[b*T + t, :]   (text.py:245-254: mean of the last len(word)
 * non-pad positions; video.py:228: mean over all tokens).  x f32 [B*T, dim] -> out f32, row stride ld_out. */
int tribe_segment_mean_fwd(const float* x, int64_t B, int64_t T, int64_t dim, const int64_t* start, const int64_t* len,
                           float* out, int64_t ld_out, void* stream);

/* Conv3d patch embedding with stride == kernel (VJEPA2PatchEmbeddings3D, modeling_vjepa2.py:84-117) as im2col:
 * pixels f32 [B, frames, chans, H, W] -> bf16 [B * tokens, K_pad], K = chans*tubelet*patch*patch in Conv3d weight order */
int tribe_im2col3d_fwd(const float* pixels, int64_t B, int32_t frames, int32_t chans, int32_t height, int32_t width,
                       int32_t tubelet, int32_t patch, uint16_t* out, int64_t K_pad, void* stream);

typedef struct tribe_vit_layer {
  const float* norm1_w; const float* norm1_b;
  const uint16_t* w_qkv; const float* b_qkv;   /* bf16 [3*dim, dim] rows q | k | v, f32 [3*dim] */
  const uint16_t* w_proj; const float* b_proj; /* bf16 [dim, dim] */
  const float* norm2_w; const float* norm2_b;
  const uint16_t* w_fc1; const float* b_fc1;   /* bf16 [mlp, dim] */
  const uint16_t* w_fc2; const float* b_fc2;   /* bf16 [dim, mlp] */
} tribe_vit_layer;

/* fp8 variant of a ViT layer's four Linear weights (see tribe_llama_fp8_layer): i = 0 qkv, 1 proj, 2 fc1, 3 fc2 */
typedef struct tribe_vit_fp8_layer {
  const uint8_t* w_qkv; const uint8_t* w_proj; const uint8_t* w_fc1; const uint8_t* w_fc2;
  float w_scale[4];
  float in_scale[4];
} tribe_vit_fp8_layer;

typedef struct tribe_vjepa2_desc {
  int64_t B;                                   /* clips */
  int32_t frames, chans, height, width, tubelet, patch;
  int32_t dim, depth, heads, dim_head, mlp;
  float ln_eps;
  const uint16_t* w_patch; const float* b_patch; int64_t K_pad;  /* bf16 [dim, K_pad] */
  const tribe_vit_layer* layers_host;          /* HOST array [depth] */
  const float* cos_tab; const float* sin_tab;  /* f32 [tokens, dim_head] per-element 3-D rope tables */
  const float* pixels;                         /* f32 [B, frames, chans, H, W] (output of the HF video processor) */
  const tribe_vit_fp8_layer* fp8_host;         /* HOST array [depth] or NULL: e4m3 GEMMs (dim and mlp multiples of 128) */
  float* amax_out;                             /* device f32 [depth, 4] or NULL: calibration of the bf16 path (max-accumulated) */
} tribe_vjepa2_desc;

size_t tribe_vjepa2_workspace_bytes(const tribe_vjepa2_desc* d);
/* VJEPA2Model encoder forward with output_hidden_states (video.py:247-274), fused with the token mean of
 * video.py:228: states f32 [depth + 1, B, dim] (state 0 = patch embeddings, state l = output of layer l). */
int tribe_vjepa2_fwd(const tribe_vjepa2_desc* d, float* states, void* workspace, size_t workspace_bytes, void* stream);

/* causal depthwise Conv1d over time (left pad K-1, no bias) + LayerNorm over channels + swish
 * (Wav2Vec2BertConvolutionModule, modeling_wav2vec2_bert.py:214-222).  x, y bf16 [B*T, C]; w_kc f32 [K, C] (tap-major). */
int tribe_dwconv_ln_swish_fwd(const uint16_t* x, int64_t B, int64_t T, int32_t C, int32_t K, const float* w_kc,
                              const float* ln_w, const float* ln_b, float eps, uint16_t* y, void* stream);
/* out[b, i, :] = x[b*T + idx[i], :]  (nearest-neighbour F.interpolate along time, audio.py:163-171) */
int tribe_gather_rows_fwd(const float* x, int64_t B, int64_t T, int64_t dim, const int64_t* idx, int64_t n, float* out,
                          void* stream);

typedef struct tribe_conformer_layer {
  const float* ffn1_ln_w; const float* ffn1_ln_b;
  const uint16_t* w_ffn1_in; const float* b_ffn1_in;     /* bf16 [inter, dim] */
  const uint16_t* w_ffn1_out; const float* b_ffn1_out_half; /* bf16 [dim, inter]; bias pre-multiplied by 0.5 */
  const float* attn_ln_w; const float* attn_ln_b;
  const uint16_t* w_qkv; const float* b_qkv;             /* bf16 [3*dim, dim] */
  const uint16_t* dist_emb;                              /* bf16 [left + right + 1, dim_head] */
  const uint16_t* w_attn_out; const float* b_attn_out;   /* bf16 [dim, dim] */
  const float* conv_ln_w; const float* conv_ln_b;
  const uint16_t* w_pw1;                                 /* bf16 [2*dim, dim], rows interleaved (a_0, b_0, a_1, b_1, ...) for the GLU epilogue */
  const float* w_dw_kc;                                  /* f32 [K, dim] depthwise taps, tap-major */
  const float* dw_ln_w; const float* dw_ln_b;
  const uint16_t* w_pw2;                                 /* bf16 [dim, dim] */
  const float* ffn2_ln_w; const float* ffn2_ln_b;
  const uint16_t* w_ffn2_in; const float* b_ffn2_in;
  const uint16_t* w_ffn2_out; const float* b_ffn2_out_half;
  const float* final_ln_w; const float* final_ln_b;
} tribe_conformer_layer;

/* fp8 variant of a Conformer layer's four feed-forward weights (see tribe_llama_fp8_layer): i = 0 ffn1.intermediate_dense, 1 ffn1.output_dense,
 * 2 ffn2.intermediate_dense, 3 ffn2.output_dense -- 70 % of the layer's GEMM flops; the attention and convolution-module projections stay bf16 */
typedef struct tribe_conformer_fp8_layer {
  const uint8_t* w_ffn1_in; const uint8_t* w_ffn1_out; const uint8_t* w_ffn2_in; const uint8_t* w_ffn2_out;
  float w_scale[4];
  float in_scale[4];
} tribe_conformer_fp8_layer;

typedef struct tribe_w2vbert_desc {
  int64_t B, T;                       /* chunks x frames (no padding mask: the reference passes single unpadded chunks) */
  int32_t feat_dim, feat_pad;         /* 160, padded to a multiple of 64 */
  int32_t dim, depth, heads, dim_head, inter, conv_kernel, rel_left, rel_right;
  float ln_eps;
  const float* fp_ln_w; const float* fp_ln_b;            /* feature_projection.layer_norm */
  const uint16_t* w_fp; const float* b_fp;               /* bf16 [dim, feat_pad] */
  const tribe_conformer_layer* layers_host;              /* HOST array [depth] */
  const float* features;              /* f32 [B*T, feat_dim] (output of the HF SeamlessM4T feature extractor) */
  const int64_t* out_index; int64_t n_out;               /* frame index of every output time point (nearest interpolation) */
  const tribe_conformer_fp8_layer* fp8_host;             /* HOST array [depth] or NULL: e4m3 feed-forward GEMMs (BASELINE config 5) */
  float* amax_out;                                       /* device f32 [depth, 4] or NULL: calibration of the bf16 path (max-accumulated) */
} tribe_w2vbert_desc;

size_t tribe_w2vbert_workspace_bytes(const tribe_w2vbert_desc* d);
/* Wav2Vec2BertModel forward with output_hidden_states (audio.py:253-263) fused with the nearest-neighbour resampling
 * to 2 Hz (audio.py:163-171): states f32 [depth + 1, B, n_out, dim]. */
int tribe_w2vbert_fwd(const tribe_w2vbert_desc* d, float* states, void* workspace, size_t workspace_bytes, void* stream);

typedef struct tribe_llama_layer {
  const float* input_norm_w;     /* [dim] */
  const uint16_t* w_qkv;         /* bf16 [(hq + 2 hkv) * dh, dim]: q | k | v rows */
  const uint16_t* w_o;           /* bf16 [dim, hq * dh] */
  const float* post_norm_w;      /* [dim] */
  const uint16_t* w_gate_up;     /* bf16 [2 * inter, dim], rows interleaved: gate_0, up_0, gate_1, up_1, ... */
  const uint16_t* w_down;        /* bf16 [dim, inter] */
} tribe_llama_layer;

/* fp8 variant of a layer's four Linear weights (BASELINE config 5): e4m3 bytes in the same row order as the bf16 packs,
 * per-tensor scales w_scale[i] (weight = byte value * scale) and static per-tensor input scales in_scale[i] for the
 * four GEMM inputs, i = 0 qkv (normed x), 1 o_proj (attention output), 2 gate_up (normed x), 3 down (SwiGLU output). */
typedef struct tribe_llama_fp8_layer {
  const uint8_t* w_qkv; const uint8_t* w_o; const uint8_t* w_gate_up; const uint8_t* w_down;
  float w_scale[4];
  float in_scale[4];
} tribe_llama_fp8_layer;

typedef struct tribe_llama_desc {
  int64_t B, T;                  /* right-padded batch of token ids */
  int32_t dim, depth, heads_q, heads_kv, dim_head, inter;
  float rms_eps;
  const void* embed; int32_t embed_dtype; int64_t vocab;
  const tribe_llama_layer* layers_host;   /* HOST array [depth] */
  const float* final_norm_w;
  const float* cos_tab; const float* sin_tab;   /* f32 [T, dim_head/2] (rope scaling already applied) */
  const int64_t* ids;            /* [B*T] */
  const int64_t* pool_start; const int64_t* pool_len;  /* [B] token window averaged per hidden state */
  const tribe_llama_fp8_layer* fp8_host;  /* HOST array [depth] or NULL: run the four Linear GEMMs of every layer in fp8
                                             (needs dim, heads_q * dim_head and inter to be multiples of 128) */
  float* amax_out;               /* device f32 [depth, 4] or NULL: max |input| of the four GEMMs per layer (calibration run
                                    of the bf16 path; accumulated with max, zero it first) */
} tribe_llama_desc;

size_t tribe_llama_workspace_bytes(const tribe_llama_desc* d);
/* LlamaModel forward with output_hidden_states (text.py:236-240) fused with the per-word pooling of
 * text.py:245-254: states f32 [depth + 1, B, dim] (state 0 = embeddings, last = after the final RMSNorm). */
int tribe_llama_fwd(const tribe_llama_desc* d, float* states, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * a7 + a8: SubjectLayers.forward (common.py:45-67) and AdaptiveAvgPool1d (model.py:119-120)
 *   y[b, v, t] = sum_c x[b, t, c] * w[subj[b], c, v] + bias[subj[b], v]
 * x bf16 [B, T, C_pad]; w_packed from tribe_pack_subject_weights; y f32 [B, V, T].
 * subjects: int64 [B] device; every entry must be < S (checked by the host wrapper,
 * mirroring the assert at common.py:53-55).
 * ------------------------------------------------------------------------- */
int tribe_voxel_head_fwd(const uint16_t* x, int64_t B, int64_t T, int64_t C_pad,
                         const uint16_t* w_packed, int64_t S, int64_t V, int64_t V_pad,
                         const float* bias /*[S, V] or NULL*/, const int64_t* subjects,
                         float* y, void* stream);
int tribe_adaptive_avg_pool_fwd(const float* x, int64_t rows, int64_t T_in, float* y, int64_t T_out, void* stream);

/* ------------------------------------------------------------------------- *
 * a10-a14: loss and per-voxel Pearson on [B, V, T'] predictions / targets
 * (pl_module.py:54-56 flattens "b d t -> (b t) d"; the reductions below are
 * order-independent so the flatten is never materialised).
 * ------------------------------------------------------------------------- */
/* nn.MSELoss(): out[0] = mean((pred-true)^2) over all elements (f32 scalar) */
int tribe_mse_fwd(const float* pred, const float* truth, int64_t n, float* out,
                  void* workspace, size_t workspace_bytes, void* stream);
size_t tribe_mse_workspace_bytes(int64_t n);
/* accumulate f64 sufficient statistics per (group, voxel):
 *   stats[g][v][0..5] += {sum x, sum y, sum x^2, sum y^2, sum xy, count}
 * x = pred, y = true, over rows b with group[b] == g (group == NULL: all rows -> g = 0) and all t.
 * metrics/base.py:26-29,52-78 (MultidimPearsonCorrCoef / GroupedMetric) and main.py:459-477. */
int tribe_pearson_stats_update(const float* pred, const float* truth, int64_t B, int64_t V, int64_t T,
                               int64_t sb, int64_t sv, int64_t st, /* element strides of b, v, t ([B,V,T] contiguous: V*T, T, 1) */
                               const int64_t* group, int64_t n_groups, double* stats, void* stream);
/* r[g][v] from stats (f32 out); count < 2 or zero variance -> NaN (scipy/torchmetrics behaviour) */
int tribe_pearson_from_stats(const double* stats, int64_t n_groups, int64_t V, float* r, void* stream);
/* PearsonLoss (losses.py:17-42) over the flattened [(B T'), V] view: out[0] = mean|sum_v (1 - r_v), eps 1e-8 */
int tribe_pearson_loss_fwd(const float* pred, const float* truth, int64_t B, int64_t V, int64_t T,
                           int64_t sb, int64_t sv, int64_t st,
                           int32_t reduction_sum, float* out, void* workspace, size_t workspace_bytes, void* stream);
size_t tribe_pearson_loss_workspace_bytes(int64_t V);

/* ------------------------------------------------------------------------- *
 * Backward building blocks (pl_module.training_step: loss.backward() of the path).
 * Composition lives in the host-side autograd functions (modeling_utils/autograd.py).
 * ------------------------------------------------------------------------- */
/* out[z, c, r] = in[z, r, c] as bf16, rows of `out` zero-padded to R_pad; `in` is f32 or bf16 with element strides
 * (s_z, s_r, 1).  Used for the transposed operands of the weight-gradient GEMMs and of attention backward. */
int tribe_transpose_bf16(const void* in, int32_t in_dtype, int64_t Z, int64_t R, int64_t C, int64_t s_z, int64_t s_r,
                         uint16_t* out, int64_t so_z, int64_t R_pad, void* stream);
/* the same with two batch levels (z = z1 * Z0 + z0 reads in + z1 * s_z1 + z0 * s_z0; out batch z contiguous at so_z):
 * all (batch, head) slices of a fused [B*T, heads*dim] activation in one launch */
int tribe_transpose_bf16_b2(const void* in, int32_t in_dtype, int64_t Z1, int64_t Z0, int64_t R, int64_t C, int64_t s_z1, int64_t s_z0,
                            int64_t s_r, uint16_t* out, int64_t so_z, int64_t R_pad, void* stream);
/* out[n] (+)= sum_m a[m, n] * (b ? b[m, n] : 1)   (bias, residual_scale and positional-embedding gradients) */
int tribe_colsum_fwd(const void* a, int32_t a_dtype, const float* b, int64_t M, int64_t N, int64_t ld, float* out,
                     int32_t accumulate, void* stream);
/* ScaleNorm backward: y = x * s / max(||x||, eps), s = gain_scale * g.
 *   dx[m, :] = (dres ? dres[m, :] * (rs ? rs : 1) : 0) + s / ||x|| * (dy - xhat * <xhat, dy>);   dg += gain_scale * sum_m <xhat, dy>
 * dy bf16 or f32 [M, dim]; dres / dx f32 (may alias). */
int tribe_scalenorm_bwd(const float* x, const void* dy, int32_t dy_dtype, const float* g, float gain_scale, float eps,
                        int64_t rows, int64_t dim, const float* dres, const float* rs, float* dx, float* dg, void* stream);
/* dS = P * (dP - rowsum(P * dP)) * scale   (softmax backward; P bf16, dP f32, dS bf16; row strides ld_p, ld_dp, ld_ds;
 * columns >= T of dS are zero-filled up to T_pad) */
int tribe_softmax_bwd(const uint16_t* P, const float* dP, int64_t rows, int64_t T, int64_t T_pad, int64_t ld_p, int64_t ld_dp,
                      float scale, uint16_t* dS, int64_t ld_ds, void* stream);
/* row softmax used by the materialised attention of the training path: S f32 [rows, T] -> P bf16 [rows, T_pad] */
int tribe_softmax_fwd(const float* S, int64_t rows, int64_t T, int64_t ld_s, uint16_t* P, int64_t T_pad, int64_t ld_p, void* stream);
/* d pred = gscale[0] * 2 / n * (pred - true)   (nn.MSELoss backward) */
int tribe_mse_bwd(const float* pred, const float* truth, int64_t n, const float* gscale, float* dpred, void* stream);
/* adaptive average pool backward: dx[r, t] = sum_{i: t in window i} dy[r, i] / |window i| */
int tribe_adaptive_avg_pool_bwd(const float* dy, int64_t rows, int64_t T_in, int64_t T_out, float* dx, void* stream);
/* out[idx[b], v] += sum_t x[b, v, t]   (SubjectLayers bias gradient) */
int tribe_rowsum_scatter(const float* x, int64_t B, int64_t V, int64_t T, const int64_t* idx, float* out, void* stream);
/* One pass over a gradient dy = a f32 [M, N] (row stride ld): sum_a[n] = sum_m a[m, n] (bias gradient), sum_ab[n] = sum_m a[m, n] b[m, n]
 * (res_scale gradient: b = the residual input), a_bf16 = bf16(a) with row stride ld_bf16 (the GEMM operand of dgrad / wgrad).  Any of the
 * three outputs may be NULL; N % 4 == 0.  Replaces `grad.sum(0)`, `(grad * res).sum(0)` and the cast of autograd's Linear backward. */
int tribe_colsum_cast_fwd(const float* a, const float* b, int64_t M, int64_t N, int64_t ld, float* sum_a, float* sum_ab, uint16_t* a_bf16,
                          int64_t ld_bf16, void* stream);
/* dst[idx[b], :] += src[b, :] for b = 0 .. B-1 in that order (deterministic; no atomics): src f32 [B, n], dst f32 [S, n], n % 4 == 0.
 * SubjectLayers weight gradient: the per-sample products x_b^T dy_b of one batched GEMM summed into their subject's slab
 * (modeling_utils/models/common.py:60-76 is the forward it differentiates). */
int tribe_slab_scatter_sum(const float* src, int64_t B, int64_t n, const int64_t* idx, float* dst, void* stream);
/* y = x * rs (columns), f32 [M, N]; rs NULL = copy */
int tribe_scale_cols_fwd(const float* x, const float* rs, int64_t M, int64_t N, float* y, void* stream);
/* PearsonLoss backward (losses.py:17-42) on [B, V, T] views (element strides sb, sv, st shared by pred / true / dpred
 * which is contiguous [B, V, T]): stats f64 [V, 6] as produced by tribe_pearson_stats_update (one group);
 * dpred = gscale[0] * w * d(1 - r_v)/d pred,  w = 1/V (mean) or 1 (sum). */
int tribe_pearson_loss_bwd(const float* pred, const float* truth, int64_t B, int64_t V, int64_t T, int64_t sb, int64_t sv,
                           int64_t st, const double* stats, int32_t reduction_sum, const float* gscale, float* dpred, void* stream);
/* InfoNCE pieces (model.py:208-221): lse[r] = log sum_j exp(S[r, j]) and diag[r] = S[r, r] of a square f32 matrix;
 * dL[i, j] = gscale[0] * 0.5 / N * (exp(S[i,j] - lse_r[i]) + exp(S[i,j] - lse_c[j]) - 2 [i == j])  as bf16 [N, N_pad] */
int tribe_lse_rows_fwd(const float* S, int64_t N, int64_t ld, float* lse, float* diag, void* stream);
int tribe_infonce_dlogits(const float* S, int64_t N, int64_t ld, const float* lse_r, const float* lse_c, const float* gscale,
                          uint16_t* dL, int64_t N_pad, void* stream);
/* f32 -> bf16 elementwise (gradient casts) */
int tribe_cast_bf16_fwd(const float* x, int64_t n, uint16_t* y, void* stream);

/* ---- segment assembly from HBM-resident extractor outputs --------------------------------
 * Replaces the host-side numpy assembly of a segment's feature tensor: `_aggregate_layers`
 * (data_utils/features/text.py:129-149, audio.py:174-194, video.py:147-167), `TimedArray.overlap` / `__iadd__`
 * (data_utils/base.py:130-211) driven by the feature `__call__`s (text.py:85-124, audio.py:78-120, neuro.py:60-106),
 * plus the fp32 H2D copy and the "b (l d) t -> b t (l d)" transpose of model.py:146-155.  The index decisions
 * (rounding, clamping) stay on the host (data_utils/base.py of this build); these kernels move the bytes. */

/* one time slice of a cached, layer-aggregated array that lands in a segment's output (32 bytes, device memory) */
typedef struct tribe_feature_piece {
  const float* src;   /* device pointer to an f32 [C, ld] array: channel c starts at src + c * ld              */
  int64_t ld;         /* samples per channel row                                                               */
  int32_t src_first;  /* first source sample                                                                   */
  int32_t src_count;  /* == dst_count, or 1 = that one sample is broadcast over the destination run            */
  int32_t dst_first;  /* first destination step (0 <= dst_first, dst_first + dst_count <= T)                   */
  int32_t dst_count;  /* destination run length                                                                */
} tribe_feature_piece;

/* out[b, g, i] = mean over s in [lo[g], hi[g]) of states[b, s, i]  (f32; sequential adds in layer order, one divide:
 * bit-identical to numpy's latents[l1:l2].mean(0)); groups of one select single layers (layer_aggregation = None).
 * states [batch, n_states, plane], out [batch, n_groups, plane]; lo / hi int32 device arrays of n_groups entries. */
int tribe_group_mean_fwd(const float* states, int64_t batch, int64_t n_states, int64_t plane, const int32_t* lo,
                         const int32_t* hi, int32_t n_groups, float* out, void* stream);
/* Sum the pieces of each segment (seg_ptr: int32 [B + 1] CSR offsets into pieces, in the order the reference adds
 * them) into a zero-initialised output:
 *   out_dtype = TRIBE_BF16: packed projector operand, bf16 [B * T, C_pad] rows (time-major, channels contiguous,
 *                           columns >= C zero) -- what tribe_pack_features would have produced from [B, C, T];
 *   out_dtype = TRIBE_F32 : reference layout f32 [B, C, T] (targets such as the fMRI array; C_pad ignored).
 * The caller guarantees src_first + src_count <= ld and the dst bounds above (checked on the host by the binding). */
int tribe_segment_gather_fwd(const tribe_feature_piece* pieces, const int32_t* seg_ptr, int64_t B, int64_t C, int64_t T,
                             void* out, int32_t out_dtype, int64_t C_pad, void* stream);
/* Word features (frequency-0 arrays held for a word's duration, text.py:190-202): out[row] = bf16(sum of
 * table[word_idx[k]] for k in [row_ptr[row], row_ptr[row + 1])), rows = B * T, table f32 [n_words, C],
 * out bf16 [rows, C_pad]; lists keep the segment's event order, so the f32 sum is the reference's sum. */
int tribe_word_bag_fwd(const float* table, int64_t n_words, int64_t C, const int32_t* row_ptr, const int32_t* word_idx,
                       int64_t rows, uint16_t* out, int64_t C_pad, void* stream);
/* Same sums kept in f32, out f32 [rows, C]: the exact tensor a feature plugin's __call__ returns (text.py:85-124,
 * `out += ta` over the segment's words) before any rounding; the plugin transposes it to [C, T] with
 * tribe_transpose_f32_fwd. */
int tribe_word_bag_f32_fwd(const float* table, int64_t n_words, int64_t C, const int32_t* row_ptr, const int32_t* word_idx,
                           int64_t rows, float* out, void* stream);

/* ---- steps after the model ------------------------------------------------------------ */
/* out[z, c, r] = in[z, r, c], f32: predictions [B, V, T'] -> [B, T', V] rows for the submission writer
 * (`pred = y_pred[i].cpu().numpy().T`, algonauts2025/callbacks.py:63-64), one launch + one D2H copy per batch. */
int tribe_transpose_f32_fwd(const float* in, int64_t Z, int64_t R, int64_t C, float* out, void* stream);
/* Ensemble averaging (algonauts2025/grids/average_submissions.py:107-125): out[i] = sum_n preds[n, i] * w, i < M,
 * sequential in n, product and sum rounded separately (numpy's `np.sum(preds * weights, axis=0)`):
 *   w_column f32 [N, V] (per-voxel weights, :96-99): w = w_column[n, i % V], f32 arithmetic, out f32 [M];
 *   w_set    f64 [N]    (score weights,    :100-103): w = w_set[n], f64 arithmetic, out f64 [M].
 * Exactly one of the two is non-NULL.  The unweighted mean (:121) is tribe_group_mean_fwd with one group [0, N). */
int tribe_weighted_sum_fwd(const float* preds, int64_t N, int64_t M, int64_t V, const float* w_column, const double* w_set,
                           void* out, void* stream);
/* corr f64 [N, N] = np.corrcoef of the N rows of x f32 [N, K] (average_submissions.py:38-53; N <= 64): f64 row means,
 * f64 products of the centred rows accumulated per K slice and merged with f64 atomics (order not fixed: agreement
 * with numpy to ~1e-12, not bit-exact). */
size_t tribe_corr_matrix_workspace_bytes(int64_t N);
int tribe_corr_matrix_fwd(const float* x, int64_t N, int64_t K, double* corr, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fp8 (OCP e4m3) GEMM for the frozen extractors (BASELINE config 5: "fp8 MFMA extractor GEMMs") ----------------
 * Same descriptor and epilogue as tribe_gemm_bf16 (the nn.Linear calls of transformers' modeling_llama.py /
 * modeling_vjepa2.py / modeling_wav2vec2_bert.py), but A [M, K] and B [N, K] hold e4m3 bytes (K contiguous; lda / ldb /
 * batch strides in elements = bytes, multiples of 16; K a multiple of 128) and `alpha` carries scale_A * scale_B of the
 * per-tensor quantisation.  v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, f32 accumulation. */
int tribe_gemm_fp8(const tribe_gemm_desc* d, void* stream);
/* out[m, k] = e4m3(clamp(x[m, k] * inv_scale, -448, 448)) (round to nearest even), x f32 | bf16 [M, K] with row stride ld,
 * out [M, K_pad] bytes, columns K..K_pad zero; K_pad a multiple of 16. */
int tribe_quantize_fp8_fwd(const void* x, int32_t x_dtype, int64_t M, int64_t K, int64_t ld, float inv_scale, uint8_t* out,
                           int64_t K_pad, void* stream);
/* The pre-norm of an extractor layer fused with the quantisation of the Linear input that follows it (transformers' LlamaRMSNorm /
 * nn.LayerNorm inside modeling_llama.py / modeling_vjepa2.py, then the e4m3 operand): y8 = tribe_quantize_fp8_fwd(bf16 output of
 * tribe_rmsnorm_fwd (layernorm = 0, b NULL) or tribe_layernorm_fwd (layernorm = 1, b may be NULL)), bit for bit, in ONE pass over x
 * (f32 [rows, dim], dim % 16 == 0; y8 [rows, dim] bytes). */
int tribe_norm_quantize_fp8_fwd(const float* x, int64_t rows, int64_t dim, const float* w, const float* b, int32_t layernorm, float eps,
                                float inv_scale, uint8_t* y8, void* stream);
/* out[0] = max(out[0] if accumulate else 0, max |x|)  (device float; calibration of the per-tensor scales) */
int tribe_absmax_fwd(const void* x, int32_t x_dtype, int64_t M, int64_t K, int64_t ld, float* out, int32_t accumulate, void* stream);

/* ---- optimiser step of pl_module.training_step's loop (grids/defaults.py:126-133: torch.optim.Adam, lr 1e-4, wd 0) ----
 * One launch per parameter group: p, m (exp_avg), v (exp_avg_sq) updated in place from g with torch.optim.Adam's
 * arithmetic (decoupled = 1: AdamW).  `step` is the 1-based count used for the bias corrections.  The work list cuts
 * every tensor into chunks of tribe_adam_chunk_elems() elements: chunk c = elements [chunk_start[c], +chunk) of
 * table[chunk_tensor[c]].  All three arrays live in device memory. */
typedef struct tribe_adam_tensor {
  float* p; const float* g; float* m; float* v; int64_t n;
  uint16_t* p_bf16;   /* optional (NULL = none): bf16 copy of the updated parameter, same element order -- the packed GEMM operand of the next
                       * forward, written by the optimiser step instead of a separate cast pass (tribe_adam_step only) */
} tribe_adam_tensor;
int64_t tribe_adam_chunk_elems(void);
int tribe_adam_step(const tribe_adam_tensor* table, const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int64_t step, int32_t decoupled, void* stream);
/* Running weight average of the reference's StochasticWeightAveraging callback (algonauts2025/main.py:365-373; the averaging rule is
 * torch.optim.swa_utils.AveragedModel's default: avg += (p - avg) * weight with weight = 1 / (n_averaged + 1), weight 1 = copy).
 * Same table and work list as tribe_adam_step: table[i].p = the average (updated in place), table[i].g = the live parameter;
 * m and v are not touched and may be NULL. */
int tribe_swa_update(const tribe_adam_tensor* table, const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks, float weight,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRIBE_HIP_H */
