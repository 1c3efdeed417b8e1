// Op-level entry points of the TRIBE encode path: each lowers one reference
// function to launches of the MFMA GEMM (gemm.hip) and the streaming kernels
// (elementwise.hip) on the caller's stream.  Host code only; no allocation, no sync.
#include <string.h>

#include "common.h"

int tribe_internal_softmax(const float* S, int64_t R, int64_t T, int64_t ld_s, uint16_t* P, int64_t T_pad, int64_t ld_p,
                           hipStream_t stream);
int tribe_internal_transpose_v(const uint16_t* qkv, int64_t B, int64_t T, int heads, int dim_head, uint16_t* vt, int64_t T_pad,
                               hipStream_t stream);
int tribe_internal_attention_fused_supported(int dim_head);
int tribe_internal_attention_fused(const uint16_t* qkv, int64_t B, int64_t T, int heads, int dim_head, float scale, uint16_t* out,
                                   hipStream_t s);

void tribe_internal_attention_set_wide384(int on);
void tribe_internal_attention_set_d64_variant(int v);
int tribe_internal_attention_rotates_q(int dim_head);
int tribe_internal_attention_fused_qrot(const uint16_t* qkv, int64_t B, int64_t T, int heads, int dim_head, float scale, uint16_t* out,
                                        const float* cos_tab, const float* sin_tab, int rot_dim, hipStream_t s);
static int g_attn_mode = 0;  // 0 = fused kernel when the head size has one, 1 = always the 3-kernel (materialised) path
extern "C" int tribe_attention_set_mode(int32_t mode) {
  TRIBE_REQUIRE(mode >= 0 && mode <= 5, "tribe_attention_set_mode: mode must be 0 (auto), 1 (materialised scores), 2 (fused, 16-row waves at every head size), 3 (dim_head 384 on the key-split kernel), 4 / 5 (dim_head 64 on the 4-wave / the anti-phase 8-wave kernel)");
  tribe_internal_attention_set_wide384(mode == 2 ? 0 : (mode == 3 ? 2 : 1));
  tribe_internal_attention_set_d64_variant(mode == 4 ? 1 : (mode == 5 ? 2 : 0));
  g_attn_mode = mode == 1 ? 1 : 0;
  return 0;
}

namespace {

inline tribe_gemm_desc gemm_zero() {
  tribe_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.batch1 = d.batch0 = 1;
  d.alpha = 1.0f;
  d.c_dtype = TRIBE_F32;
  return d;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// attention is processed in batch chunks so that the f32 score block stays near the Infinity Cache size
constexpr size_t ATTN_SCORE_BUDGET = 192ull << 20;

struct AttnPlan {
  int64_t T_pad, chunk;
  size_t s_bytes, p_bytes, vt_bytes;
};

inline AttnPlan attn_plan(int64_t B, int64_t T, int heads, int dim_head) {
  AttnPlan p;
  p.T_pad = round_up(T, 64);
  const size_t per_b = (size_t)heads * T * p.T_pad * sizeof(float);
  int64_t chunk = (int64_t)(ATTN_SCORE_BUDGET / per_b);
  if (chunk < 1) chunk = 1;
  if (chunk > B) chunk = B;
  p.chunk = chunk;
  p.s_bytes = align256((size_t)chunk * per_b);
  p.p_bytes = align256((size_t)chunk * heads * T * p.T_pad * 2);
  p.vt_bytes = align256((size_t)chunk * heads * dim_head * p.T_pad * 2);
  return p;
}

}  // namespace

extern "C" int tribe_projector_fwd(const uint16_t* feat_packed, int64_t BT, int64_t T, int64_t K_pad, const uint16_t* w_packed,
                                   const float* bias, int64_t N_out, float* x, int64_t hidden, int64_t col0, int32_t accumulate,
                                   const float* pos_embed, const float* subj_embed, const int64_t* subject_id, void* stream) {
  TRIBE_REQUIRE(feat_packed && w_packed && x, "tribe_projector_fwd: null pointer");
  TRIBE_REQUIRE(BT > 0 && T > 0 && BT % T == 0, "tribe_projector_fwd: BT=%lld must be a positive multiple of T=%lld",
                (long long)BT, (long long)T);
  TRIBE_REQUIRE(N_out > 0 && col0 >= 0 && col0 + N_out <= hidden, "tribe_projector_fwd: column slice [%lld, %lld) outside hidden=%lld",
                (long long)col0, (long long)(col0 + N_out), (long long)hidden);
  TRIBE_REQUIRE(!subj_embed || subject_id, "tribe_projector_fwd: subject_embed without subject_id");
  tribe_gemm_desc d = gemm_zero();
  d.M = BT; d.N = N_out; d.K = K_pad;
  d.A = feat_packed; d.lda = K_pad;
  d.B = w_packed; d.ldb = K_pad;
  d.C = x + col0; d.ldc = hidden; d.c_dtype = TRIBE_F32;
  if (bias) { d.bias = bias; d.bias_mode = TRIBE_BIAS_COL; }
  if (accumulate) { d.res = x + col0; d.ldres = hidden; }
  if (pos_embed) { d.rowadd = pos_embed + col0; d.ld_rowadd = hidden; d.rowadd_period = T; }
  if (subj_embed) { d.gadd = subj_embed + col0; d.gadd_index = subject_id; d.gadd_div = T; d.ld_gadd = hidden; }
  d.role = TRIBE_ROLE_PROJECTOR;
  return tribe_gemm_bf16(&d, stream);
}

extern "C" size_t tribe_attention_workspace_bytes(int64_t B, int64_t T, int32_t heads, int32_t dim_head) {
  if (B <= 0 || T <= 0 || heads <= 0 || dim_head <= 0) return 0;
  const AttnPlan p = attn_plan(B, T, heads, dim_head);
  return p.s_bytes + p.p_bytes + p.vt_bytes;
}

extern "C" int tribe_attention_fwd(const uint16_t* qkv, int64_t B, int64_t T, int32_t heads, int32_t dim_head, float scale,
                                   uint16_t* out, void* workspace, size_t workspace_bytes, void* stream) {
  TRIBE_REQUIRE(qkv && out && workspace, "tribe_attention_fwd: null pointer");
  TRIBE_REQUIRE(B > 0 && T > 0 && heads > 0 && dim_head > 0, "tribe_attention_fwd: bad shape");
  TRIBE_REQUIRE(dim_head % 64 == 0, "tribe_attention_fwd: dim_head=%d must be a multiple of 64", dim_head);
  TRIBE_REQUIRE(workspace_bytes >= tribe_attention_workspace_bytes(B, T, heads, dim_head), "tribe_attention_fwd: workspace too small");
  TRIBE_REQUIRE(((uintptr_t)workspace % 256) == 0, "tribe_attention_fwd: workspace must be 256-byte aligned");
  if (g_attn_mode == 0 && tribe_internal_attention_fused_supported(dim_head))
    return tribe_internal_attention_fused(qkv, B, T, heads, dim_head, scale, out, (hipStream_t)stream);
  const AttnPlan p = attn_plan(B, T, heads, dim_head);
  const int64_t inner = (int64_t)heads * dim_head, Tp = p.T_pad;
  float* S = (float*)workspace;
  uint16_t* P = (uint16_t*)((char*)workspace + p.s_bytes);
  uint16_t* Vt = (uint16_t*)((char*)workspace + p.s_bytes + p.p_bytes);
  hipStream_t s = (hipStream_t)stream;
  for (int64_t b0 = 0; b0 < B; b0 += p.chunk) {
    const int64_t nb = (B - b0 < p.chunk) ? B - b0 : p.chunk;
    const uint16_t* q = qkv + b0 * T * 3 * inner;
    int rc = tribe_internal_transpose_v(q, nb, T, heads, dim_head, Vt, Tp, s);
    if (rc) return rc;
    // S[b,h] = scale * Q K^T
    tribe_gemm_desc d = gemm_zero();
    d.M = T; d.N = T; d.K = dim_head; d.batch1 = nb; d.batch0 = heads;
    d.A = q; d.lda = 3 * inner; d.sA1 = T * 3 * inner; d.sA0 = dim_head;
    d.B = q + inner; d.ldb = 3 * inner; d.sB1 = T * 3 * inner; d.sB0 = dim_head;
    d.C = S; d.ldc = Tp; d.sC1 = (int64_t)heads * T * Tp; d.sC0 = T * Tp; d.c_dtype = TRIBE_F32;
    d.alpha = scale;
    d.role = TRIBE_ROLE_ATTN_SCORES;
    rc = tribe_gemm_bf16(&d, stream);
    if (rc) return rc;
    // softmax in f32 (x_transformers Attend: softmax(dtype=float32)), P rounded to bf16 for the second MFMA product
    rc = tribe_internal_softmax(S, nb * heads * T, T, Tp, P, Tp, Tp, s);
    if (rc) return rc;
    // O[b,h] = P V   (B operand = V^T, K = T_pad zero padded on both sides)
    d = gemm_zero();
    d.M = T; d.N = dim_head; d.K = Tp; d.batch1 = nb; d.batch0 = heads;
    d.A = P; d.lda = Tp; d.sA1 = (int64_t)heads * T * Tp; d.sA0 = T * Tp;
    d.B = Vt; d.ldb = Tp; d.sB1 = (int64_t)heads * dim_head * Tp; d.sB0 = (int64_t)dim_head * Tp;
    d.C = out + b0 * T * inner; d.ldc = inner; d.sC1 = T * inner; d.sC0 = dim_head; d.c_dtype = TRIBE_BF16;
    d.role = TRIBE_ROLE_ATTN_PV;
    rc = tribe_gemm_bf16(&d, stream);
    if (rc) return rc;
  }
  return 0;
}

namespace {
struct EncPlan {
  int64_t M, inner;
  size_t xn_bytes, big_bytes, attn_bytes, norm_bytes, split_bytes;
  bool fuse_norm;   // ScaleNorm folded into the GEMMs either side of it (needs whole 256-column tiles)
};
inline EncPlan enc_plan(const tribe_encoder_desc* d) {
  EncPlan p;
  p.M = d->B * d->T;
  p.inner = (int64_t)d->heads * d->dim_head;
  p.xn_bytes = align256((size_t)p.M * d->dim * 2);
  const int64_t wide = (4 * p.inner > d->ff_inner) ? 4 * p.inner : d->ff_inner;  // qkv | attn_out  aliases  ff hidden
  p.big_bytes = align256((size_t)p.M * wide * 2);
  p.attn_bytes = align256(tribe_attention_workspace_bytes(d->B, d->T, d->heads, d->dim_head));
  p.fuse_norm = d->dim % 256 == 0 && p.inner % 256 == 0 && d->ff_inner % 256 == 0;
  p.norm_bytes = p.fuse_norm ? align256((size_t)p.M * (d->dim / 32 + 1) * 4) : 0;   // partial sums of squares (<= dim / 32 slots per row, by the producer's tile) + the row factors
  // few rows (BASELINE config 1: M = 128): the four GEMMs of a layer are well under one round of tiles and run split over K
  // (tribe_gemm_desc.stream_k) -- up to 8 f32 shares of the widest output
  const int64_t widest = 3 * p.inner > d->ff_inner ? (3 * p.inner > d->dim ? 3 * p.inner : d->dim) : (d->ff_inner > d->dim ? d->ff_inner : d->dim);
  p.split_bytes = p.M <= 512 ? align256((size_t)8 * p.M * widest * 4) : 0;
  return p;
}
inline int enc_validate(const tribe_encoder_desc* d) {
  TRIBE_REQUIRE(d != nullptr, "tribe_encoder: null descriptor");
  TRIBE_REQUIRE(d->B > 0 && d->T > 0 && d->dim > 0 && d->depth >= 0 && d->heads > 0 && d->dim_head > 0 && d->ff_inner > 0,
                "tribe_encoder: bad shape");
  TRIBE_REQUIRE(d->dim % 64 == 0 && d->dim_head % 64 == 0 && d->ff_inner % 64 == 0,
                "tribe_encoder: dim=%d, dim_head=%d, ff_inner=%d must be multiples of 64", d->dim, d->dim_head, d->ff_inner);
  TRIBE_REQUIRE(d->rot_dim == 0 || (d->cos_tab && d->sin_tab), "tribe_encoder: rotary tables missing");
  TRIBE_REQUIRE(d->depth == 0 || d->layers_host, "tribe_encoder: layers_host missing");
  TRIBE_REQUIRE(d->final_norm_g, "tribe_encoder: final_norm_g missing");
  return 0;
}
}  // namespace

extern "C" size_t tribe_encoder_workspace_bytes(const tribe_encoder_desc* d) {
  if (!d || d->B <= 0 || d->T <= 0) return 0;
  const EncPlan p = enc_plan(d);
  return p.xn_bytes + p.big_bytes + p.attn_bytes + p.norm_bytes + p.split_bytes;
}

extern "C" int tribe_encoder_fwd(const tribe_encoder_desc* d, float* x, void* y, int32_t y_dtype, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  int rc = enc_validate(d);
  if (rc) return rc;
  TRIBE_REQUIRE(x && y && workspace, "tribe_encoder_fwd: null pointer");
  TRIBE_REQUIRE(((uintptr_t)workspace % 256) == 0, "tribe_encoder_fwd: workspace must be 256-byte aligned");
  TRIBE_REQUIRE(workspace_bytes >= tribe_encoder_workspace_bytes(d), "tribe_encoder_fwd: workspace too small (%zu < %zu)",
                workspace_bytes, tribe_encoder_workspace_bytes(d));
  const EncPlan p = enc_plan(d);
  const int64_t M = p.M, inner = p.inner, dim = d->dim;
  uint16_t* xn = (uint16_t*)workspace;
  uint16_t* big = (uint16_t*)((char*)workspace + p.xn_bytes);
  uint16_t* qkv = big;                         // [M, 3*inner]
  uint16_t* ao = big + (size_t)M * 3 * inner;  // [M, inner]
  uint16_t* hbuf = big;                        // [M, ff_inner]  (qkv/ao are dead by then)
  void* attn_ws = (char*)workspace + p.xn_bytes + p.big_bytes;
  float* ssq = (float*)((char*)workspace + p.xn_bytes + p.big_bytes + p.attn_bytes);   // [M, n_part] partial sums of squares
  float* rowf = ssq + (size_t)M * (dim / 32);                                          // [M] ScaleNorm factors
  void* split_ws = (char*)workspace + p.xn_bytes + p.big_bytes + p.attn_bytes + p.norm_bytes;
  // a GEMM of a small batch may run split over K: hand it the workspace when the launcher says it would use one
  auto allow_split = [&](tribe_gemm_desc& g) {
    if (p.split_bytes == 0) return;
    g.stream_k = 1;
    const int64_t need = tribe_gemm_stream_k_workspace_bytes(&g);
    if (need > 0 && (size_t)need <= p.split_bytes) { g.stream_k_ws = split_ws; g.stream_k_ws_bytes = (int64_t)p.split_bytes; }
    else g.stream_k = 0;
  };
  int64_t n_part = dim / 64;   // slots per row the LAST producer wrote (one per wave column group of its tile: tribe_gemm_sumsq_slots)
  const float scale = 1.0f / sqrtf((float)d->dim_head);
  // With whole 256-column tiles the pre-norms are folded into the GEMMs: the GEMM that writes x also leaves bf16(x) in `xn` and
  // the per-row partial sums of squares; a [M]-sized kernel turns them into the ScaleNorm factors; the next GEMM reads the raw
  // bf16(x) and scales its accumulator rows (W . (x s) = s (W . x)).  Only the first norm (x comes from the projector) and the
  // final one (its consumer, the voxel head, has x on the column side) stay stand-alone launches: 2 instead of 2 * depth + 1.
  const bool fuse = p.fuse_norm;
  bool have_factors = false;   // rowf / xn describe the current x

  for (int l = 0; l < d->depth; ++l) {
    const tribe_encoder_layer& L = d->layers_host[l];
    TRIBE_REQUIRE(L.attn_norm_g && L.w_qkv && L.w_out && L.ff_norm_g && L.w_ff1 && L.b_ff1 && L.w_ff2 && L.b_ff2,
                  "tribe_encoder_fwd: layer %d has a null parameter", l);
    // ---- attention block: x = to_out(attn(rotary(qkv(norm(x))))) + x * residual_scale ----
    tribe_gemm_desc g = gemm_zero();
    if (have_factors) {
      rc = tribe_rownorm_scale_fwd(ssq, M, n_part, L.attn_norm_g, d->norm_gain_scale, d->norm_eps, rowf, stream);
      g.row_scale = rowf;
    } else {
      rc = tribe_scalenorm_fwd(x, M, dim, L.attn_norm_g, d->norm_gain_scale, d->norm_eps, xn, TRIBE_BF16, stream);
    }
    if (rc) return rc;
    g.M = M; g.N = 3 * inner; g.K = dim;
    g.A = xn; g.lda = dim; g.B = L.w_qkv; g.ldb = dim;
    g.C = qkv; g.ldc = 3 * inner; g.c_dtype = TRIBE_BF16;
    g.role = TRIBE_ROLE_QKV;
    allow_split(g);
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
    // q heads and k heads are adjacent in the fused row.  The DH = 384 attention kernel rotates Q while it loads its Q fragments
    // (interleaved pairs; bit-identical arithmetic), so only the k heads go through the stand-alone pass: half its 805 MB.
    const bool q_in_attn = g_attn_mode == 0 && d->rot_dim > 0 && d->rotary_interleaved == 1 && d->rot_dim % 16 == 0 &&
                           tribe_internal_attention_rotates_q(d->dim_head);
    if (d->rot_dim > 0) {
      rc = q_in_attn ? tribe_rotary_fwd(qkv + inner, M, d->T, 3 * inner, d->heads, d->dim_head, d->rot_dim, d->cos_tab, d->sin_tab,
                                        d->rotary_interleaved, stream)
                     : tribe_rotary_fwd(qkv, M, d->T, 3 * inner, 2 * d->heads, d->dim_head, d->rot_dim, d->cos_tab, d->sin_tab,
                                        d->rotary_interleaved, stream);
      if (rc) return rc;
    }
    rc = q_in_attn ? tribe_internal_attention_fused_qrot(qkv, d->B, d->T, d->heads, d->dim_head, scale, ao, d->cos_tab, d->sin_tab,
                                                         d->rot_dim, (hipStream_t)stream)
                   : tribe_attention_fwd(qkv, d->B, d->T, d->heads, d->dim_head, scale, ao, attn_ws, p.attn_bytes, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = dim; g.K = inner;
    g.A = ao; g.lda = inner; g.B = L.w_out; g.ldb = inner;
    g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32;
    g.res = x; g.ldres = dim; g.res_scale = L.attn_res_scale;
    g.role = TRIBE_ROLE_OUT_PROJ;
    if (fuse) {
      g.c_bf16 = xn; g.ld_c_bf16 = dim; g.row_sumsq = ssq;
      n_part = tribe_gemm_sumsq_slots(&g);
      TRIBE_REQUIRE(n_part > 0 && n_part <= dim / 32, "tribe_encoder_fwd: unexpected row_sumsq slot count %lld", (long long)n_part);
      g.ld_row_sumsq = n_part;
    }
    allow_split(g);
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
    // ---- feed-forward block: x = W2 gelu(W1 norm(x) + b1) + b2 + x * residual_scale ----
    g = gemm_zero();
    if (fuse) {
      rc = tribe_rownorm_scale_fwd(ssq, M, n_part, L.ff_norm_g, d->norm_gain_scale, d->norm_eps, rowf, stream);
      g.row_scale = rowf;
    } else {
      rc = tribe_scalenorm_fwd(x, M, dim, L.ff_norm_g, d->norm_gain_scale, d->norm_eps, xn, TRIBE_BF16, stream);
    }
    if (rc) return rc;
    g.M = M; g.N = d->ff_inner; g.K = dim;
    g.A = xn; g.lda = dim; g.B = L.w_ff1; g.ldb = dim;
    g.C = hbuf; g.ldc = d->ff_inner; g.c_dtype = TRIBE_BF16;
    g.bias = L.b_ff1; g.bias_mode = TRIBE_BIAS_COL; g.act = TRIBE_ACT_GELU;
    g.role = TRIBE_ROLE_FF1;
    allow_split(g);
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = dim; g.K = d->ff_inner;
    g.A = hbuf; g.lda = d->ff_inner; g.B = L.w_ff2; g.ldb = d->ff_inner;
    g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32;
    g.bias = L.b_ff2; g.bias_mode = TRIBE_BIAS_COL;
    g.res = x; g.ldres = dim; g.res_scale = L.ff_res_scale;
    g.role = TRIBE_ROLE_FF2;
    have_factors = fuse && l + 1 < d->depth;   // the next layer's QKV takes the raw bf16(x) + factors
    if (have_factors) {
      g.c_bf16 = xn; g.ld_c_bf16 = dim; g.row_sumsq = ssq;
      n_part = tribe_gemm_sumsq_slots(&g);
      TRIBE_REQUIRE(n_part > 0 && n_part <= dim / 32, "tribe_encoder_fwd: unexpected row_sumsq slot count %lld", (long long)n_part);
      g.ld_row_sumsq = n_part;
    }
    allow_split(g);
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
  }
  return tribe_scalenorm_fwd(x, M, dim, d->final_norm_g, d->norm_gain_scale, d->norm_eps, y, y_dtype, stream);
}

extern "C" int tribe_voxel_head_fwd(const uint16_t* x, int64_t B, int64_t T, int64_t C_pad, const uint16_t* w_packed, int64_t S,
                                    int64_t V, int64_t V_pad, const float* bias, const int64_t* subjects, float* y, void* stream) {
  TRIBE_REQUIRE(x && w_packed && subjects && y, "tribe_voxel_head_fwd: null pointer");
  TRIBE_REQUIRE(B > 0 && T > 0 && S > 0 && V > 0 && V_pad >= V && C_pad > 0, "tribe_voxel_head_fwd: bad shape");
  // y[b] (V x T) = W_s^T (V x C) . x_b^T (C x T): the T-contiguous output needs no transposed store,
  // and the per-subject weight is selected by the gather index instead of being materialised per sample
  // (the reference's index_select copies B x 12.3 MB, common.py:61).
  tribe_gemm_desc d = gemm_zero();
  d.M = V; d.N = T; d.K = C_pad; d.batch1 = B;
  d.A = w_packed; d.lda = C_pad; d.sA1 = V_pad * C_pad; d.gather1 = subjects; d.gather_a = 1;
  d.B = x; d.ldb = C_pad; d.sB1 = T * C_pad;
  d.C = y; d.ldc = T; d.sC1 = V * T; d.c_dtype = TRIBE_F32;
  if (bias) { d.bias = bias; d.bias_mode = TRIBE_BIAS_ROW; d.gather_bias = 1; d.sBias1 = V; }
  d.role = TRIBE_ROLE_VOXEL_HEAD;
  return tribe_gemm_bf16(&d, stream);
}
