// Frozen feature extractors as sequences of the path's own kernels:
// host-side orchestration only (no allocation, no sync), one C entry point per architecture.
//
// tribe_llama_fwd: transformers LlamaModel forward with output_hidden_states=True
// (data_utils/features/text.py:236-240 -> modeling_llama.py LlamaDecoderLayer: RMSNorm -> GQA attention with
// rotary (rotate_half) and a causal mask -> residual -> RMSNorm -> SwiGLU MLP -> residual), fused with the
// reference's per-word pooling of every hidden state (text.py:245-254: strip right padding, mean of the last
// len(word) positions) so that only [n_states, B, dim] floats leave the GPU instead of every hidden state.
#include <string.h>

#include "common.h"

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

struct LlamaPlan {
  int64_t M, qkv_w, q_w;
  size_t x_b, xn_b, qkv_b, ao_b, act_b, fin_b, q8_b;
};
inline LlamaPlan llama_plan(const tribe_llama_desc* d) {
  LlamaPlan p;
  p.M = d->B * d->T;
  p.q_w = (int64_t)d->heads_q * d->dim_head;
  p.qkv_w = (int64_t)(d->heads_q + 2 * d->heads_kv) * d->dim_head;
  p.x_b = align256((size_t)p.M * d->dim * 4);
  p.xn_b = align256((size_t)p.M * d->dim * 2);
  p.qkv_b = align256((size_t)p.M * p.qkv_w * 2);
  p.ao_b = align256((size_t)p.M * p.q_w * 2);
  p.act_b = align256((size_t)p.M * d->inter * 2);
  p.fin_b = align256((size_t)p.M * d->dim * 4);
  const int64_t widest = d->inter > p.q_w ? (d->inter > d->dim ? d->inter : d->dim) : (p.q_w > d->dim ? p.q_w : d->dim);
  p.q8_b = d->fp8_host ? align256((size_t)p.M * widest) : 0;   // one e4m3 staging buffer: every quantised input is consumed at once
  return p;
}

// one Linear of an extractor layer: bf16 GEMM, or quantise the bf16 input with its static scale and run the e4m3 GEMM.
// `amax` (calibration, bf16 path only) and the fp8 fields come from the layer's descriptor; tribe_llama_fp8_layer and
// tribe_vit_fp8_layer share this layout: four weight pointers, four weight scales, four input scales.
struct Fp8Linear4 {
  const uint8_t* w[4];
  float w_scale[4];
  float in_scale[4];
};
static_assert(sizeof(Fp8Linear4) == sizeof(tribe_llama_fp8_layer) && sizeof(Fp8Linear4) == sizeof(tribe_vit_fp8_layer) &&
                  sizeof(Fp8Linear4) == sizeof(tribe_conformer_fp8_layer),
              "fp8 layer layouts");

// fp8 route of the pre-norms: the norm writes the e4m3 operand of the Linear that follows straight into the staging buffer (one pass
// over x instead of norm -> bf16 -> quantise).  *done tells the caller whether it did; otherwise the bf16 norm must run.
inline int extractor_norm_fp8(const void* fp8_layers, int layer, int which, const float* x, int64_t rows, int64_t dim, const float* w,
                              const float* b, int layernorm, float eps, uint8_t* q8, void* stream, bool* done) {
  *done = false;
  if (!fp8_layers) return 0;
  const float in_scale = ((const Fp8Linear4*)fp8_layers)[layer].in_scale[which];
  if (!(in_scale > 0.f)) return 0;   // extractor_linear reports the missing scale
  *done = true;
  return tribe_norm_quantize_fp8_fwd(x, rows, dim, w, b, layernorm, eps, 1.0f / in_scale, q8, stream);
}

inline int extractor_linear(const char* who, const void* fp8_layers, float* amax_out, int layer, int which, tribe_gemm_desc& g,
                            const void* w_bf16, uint8_t* q8, void* stream, bool a_is_q8 = false) {
  if (amax_out) {
    int rc = tribe_absmax_fwd(g.A, TRIBE_BF16, g.M, g.K, g.lda, amax_out + (int64_t)layer * 4 + which, 1, stream);
    if (rc) return rc;
  }
  if (!fp8_layers) {
    g.B = w_bf16;
    return tribe_gemm_bf16(&g, stream);
  }
  const Fp8Linear4& F = ((const Fp8Linear4*)fp8_layers)[layer];
  TRIBE_REQUIRE(F.w[which] && F.in_scale[which] > 0.f && F.w_scale[which] > 0.f, "%s: layer %d fp8 weight %d or its scales missing", who, layer,
                which);
  if (!a_is_q8) {
    int rc = tribe_quantize_fp8_fwd(g.A, TRIBE_BF16, g.M, g.K, g.lda, 1.0f / F.in_scale[which], q8, g.K, stream);
    if (rc) return rc;
  }
  g.A = q8;
  g.B = F.w[which];
  g.alpha *= F.in_scale[which] * F.w_scale[which];   // on top of the caller's alpha (Conformer half-step FFN: 0.5)
  return tribe_gemm_fp8(&g, stream);
}
}  // namespace

extern "C" size_t tribe_llama_workspace_bytes(const tribe_llama_desc* d) {
  if (!d || d->B <= 0 || d->T <= 0) return 0;
  const LlamaPlan p = llama_plan(d);
  return p.x_b + p.xn_b + p.qkv_b + p.ao_b + p.act_b + p.fin_b + p.q8_b;
}

extern "C" int tribe_llama_fwd(const tribe_llama_desc* d, float* states, void* workspace, size_t workspace_bytes, void* stream) {
  TRIBE_REQUIRE(d && states && workspace, "tribe_llama_fwd: null pointer");
  TRIBE_REQUIRE(d->B > 0 && d->T > 0 && d->dim > 0 && d->depth >= 0 && d->heads_q > 0 && d->heads_kv > 0 && d->inter > 0,
                "tribe_llama_fwd: bad shape");
  TRIBE_REQUIRE(d->heads_q % d->heads_kv == 0, "tribe_llama_fwd: heads_q=%d not a multiple of heads_kv=%d", d->heads_q, d->heads_kv);
  TRIBE_REQUIRE(d->dim % 64 == 0 && d->inter % 64 == 0 && (d->heads_q * d->dim_head) % 64 == 0,
                "tribe_llama_fwd: dim, inter and heads_q*dim_head must be multiples of 64");
  TRIBE_REQUIRE(d->embed && d->ids && d->final_norm_w && d->cos_tab && d->sin_tab && (d->depth == 0 || d->layers_host),
                "tribe_llama_fwd: missing parameter pointer");
  TRIBE_REQUIRE(((uintptr_t)workspace % 256) == 0, "tribe_llama_fwd: workspace must be 256-byte aligned");
  TRIBE_REQUIRE(workspace_bytes >= tribe_llama_workspace_bytes(d), "tribe_llama_fwd: workspace too small");
  const LlamaPlan p = llama_plan(d);
  char* w = (char*)workspace;
  float* x = (float*)w; w += p.x_b;
  uint16_t* xn = (uint16_t*)w; w += p.xn_b;
  uint16_t* qkv = (uint16_t*)w; w += p.qkv_b;
  uint16_t* ao = (uint16_t*)w; w += p.ao_b;
  uint16_t* act = (uint16_t*)w; w += p.act_b;
  float* fin = (float*)w; w += p.fin_b;
  uint8_t* q8 = (uint8_t*)w;
  const int64_t M = p.M, dim = d->dim, BD = d->B * dim;
  TRIBE_REQUIRE(!d->fp8_host || (d->dim % 128 == 0 && p.q_w % 128 == 0 && d->inter % 128 == 0),
                "tribe_llama_fwd: the fp8 path needs dim, heads_q * dim_head and inter to be multiples of 128");
  TRIBE_REQUIRE(!(d->fp8_host && d->amax_out), "tribe_llama_fwd: calibrate (amax_out) on the bf16 path, not together with fp8_host");

  int rc = tribe_embedding_fwd(d->embed, d->embed_dtype, d->ids, M, dim, d->vocab, x, stream);
  if (rc) return rc;
  rc = tribe_segment_mean_fwd(x, d->B, d->T, dim, d->pool_start, d->pool_len, states, dim, stream);
  if (rc) return rc;

  for (int l = 0; l < d->depth; ++l) {
    const tribe_llama_layer& L = d->layers_host[l];
    TRIBE_REQUIRE(L.input_norm_w && L.w_qkv && L.w_o && L.post_norm_w && L.w_gate_up && L.w_down,
                  "tribe_llama_fwd: layer %d has a null parameter", l);
    bool q_in = false;
    rc = extractor_norm_fp8(d->fp8_host, l, 0, x, M, dim, L.input_norm_w, nullptr, 0, d->rms_eps, q8, stream, &q_in);
    if (!rc && !q_in) rc = tribe_rmsnorm_fwd(x, M, dim, L.input_norm_w, d->rms_eps, xn, TRIBE_BF16, stream);
    if (rc) return rc;
    tribe_gemm_desc g = gemm_zero();
    g.M = M; g.N = p.qkv_w; g.K = dim;
    g.A = xn; g.lda = dim; g.ldb = dim;
    g.C = qkv; g.ldc = p.qkv_w; g.c_dtype = TRIBE_BF16; g.role = TRIBE_ROLE_QKV;
    rc = extractor_linear("tribe_llama_fwd", d->fp8_host, d->amax_out, l, 0, g, L.w_qkv, q8, stream, q_in);
    if (rc) return rc;
    // rotate_half rotary over the full head dim on the q heads and the k heads (adjacent in the fused row)
    rc = tribe_rotary_fwd(qkv, M, d->T, p.qkv_w, d->heads_q + d->heads_kv, d->dim_head, d->dim_head, d->cos_tab, d->sin_tab, 0, stream);
    if (rc) return rc;
    tribe_attention_desc a;
    a.q = qkv; a.k = qkv + p.q_w; a.v = qkv + p.q_w + (int64_t)d->heads_kv * d->dim_head;
    a.ld_q = a.ld_k = a.ld_v = p.qkv_w;
    a.out = ao; a.ld_out = p.q_w;
    a.B = d->B; a.T = d->T; a.heads_q = d->heads_q; a.heads_kv = d->heads_kv; a.dim_head = d->dim_head;
    a.causal = 1;  // right padding + causal mask: real tokens never see pad keys, pad rows are never pooled
    a.scale = 1.0f / sqrtf((float)d->dim_head);
    a.rel_qe = nullptr; a.ld_rel_qe = 0; a.rel_stride_h = 0; a.rel_left = a.rel_right = 0; a.lse = nullptr;
    rc = tribe_attention_fwd_ex(&a, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = dim; g.K = p.q_w;
    g.A = ao; g.lda = p.q_w; g.ldb = p.q_w;
    g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32; g.res = x; g.ldres = dim; g.role = TRIBE_ROLE_OUT_PROJ;
    rc = extractor_linear("tribe_llama_fwd", d->fp8_host, d->amax_out, l, 1, g, L.w_o, q8, stream);
    if (rc) return rc;
    rc = extractor_norm_fp8(d->fp8_host, l, 2, x, M, dim, L.post_norm_w, nullptr, 0, d->rms_eps, q8, stream, &q_in);
    if (!rc && !q_in) rc = tribe_rmsnorm_fwd(x, M, dim, L.post_norm_w, d->rms_eps, xn, TRIBE_BF16, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = 2 * (int64_t)d->inter; g.K = dim;
    g.A = xn; g.lda = dim; g.ldb = dim;
    g.C = act; g.ldc = d->inter; g.c_dtype = TRIBE_BF16; g.act = TRIBE_ACT_SWIGLU; g.role = TRIBE_ROLE_FF1;
    rc = extractor_linear("tribe_llama_fwd", d->fp8_host, d->amax_out, l, 2, g, L.w_gate_up, q8, stream, q_in);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = dim; g.K = d->inter;
    g.A = act; g.lda = d->inter; g.ldb = d->inter;
    g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32; g.res = x; g.ldres = dim; g.role = TRIBE_ROLE_FF2;
    rc = extractor_linear("tribe_llama_fwd", d->fp8_host, d->amax_out, l, 3, g, L.w_down, q8, stream);
    if (rc) return rc;
    if (l + 1 < d->depth) {
      rc = tribe_segment_mean_fwd(x, d->B, d->T, dim, d->pool_start, d->pool_len, states + (int64_t)(l + 1) * BD, dim, stream);
      if (rc) return rc;
    }
  }
  // the last hidden state is the output of the final RMSNorm (LlamaModel.norm)
  if (d->depth > 0) {
    rc = tribe_rmsnorm_fwd(x, M, dim, d->final_norm_w, d->rms_eps, fin, TRIBE_F32, stream);
    if (rc) return rc;
    rc = tribe_segment_mean_fwd(fin, d->B, d->T, dim, d->pool_start, d->pool_len, states + (int64_t)d->depth * BD, dim, stream);
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// tribe_vjepa2_fwd: transformers VJEPA2Model encoder (data_utils/features/video.py:239-274 -> modeling_vjepa2.py
// VJEPA2Encoder: Conv3d tubelet patch embedding -> depth x [LayerNorm -> q/k/v Linear(+bias) -> 3-D rotary ->
// bidirectional attention -> proj + residual -> LayerNorm -> fc1 + GELU -> fc2 + residual]); every hidden state
// (embeddings + each layer output, NOT the final LayerNorm) is averaged over tokens (video.py:228).
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct VitPlan {
  int64_t tokens, M;
  size_t x_b, xn_b, qkv_b, ao_b, act_b, col_b, q8_b;
};
inline VitPlan vit_plan(const tribe_vjepa2_desc* d) {
  VitPlan p;
  p.tokens = (int64_t)(d->frames / d->tubelet) * (d->height / d->patch) * (d->width / d->patch);
  p.M = d->B * p.tokens;
  p.x_b = align256((size_t)p.M * d->dim * 4);
  p.xn_b = align256((size_t)p.M * d->dim * 2);
  p.qkv_b = align256((size_t)p.M * 3 * d->dim * 2);
  p.ao_b = align256((size_t)p.M * d->dim * 2);
  p.act_b = align256((size_t)p.M * d->mlp * 2);
  p.col_b = align256((size_t)p.M * d->K_pad * 2);
  p.q8_b = d->fp8_host ? align256((size_t)p.M * (d->mlp > d->dim ? d->mlp : d->dim)) : 0;
  return p;
}
}  // namespace

extern "C" size_t tribe_vjepa2_workspace_bytes(const tribe_vjepa2_desc* d) {
  if (!d || d->B <= 0 || d->tubelet <= 0 || d->patch <= 0) return 0;
  const VitPlan p = vit_plan(d);
  const size_t tail = p.act_b > p.col_b ? p.act_b : p.col_b;  // the im2col buffer is dead once the embedding GEMM ran
  return p.x_b + p.xn_b + p.qkv_b + p.ao_b + tail + p.q8_b;
}

extern "C" int tribe_vjepa2_fwd(const tribe_vjepa2_desc* d, float* states, void* workspace, size_t workspace_bytes, void* stream) {
  TRIBE_REQUIRE(d && states && workspace, "tribe_vjepa2_fwd: null pointer");
  TRIBE_REQUIRE(d->B > 0 && d->dim > 0 && d->depth >= 0 && d->heads > 0 && d->mlp > 0, "tribe_vjepa2_fwd: bad shape");
  TRIBE_REQUIRE(d->heads * d->dim_head == d->dim && d->dim % 64 == 0 && d->mlp % 64 == 0 && d->K_pad % 64 == 0,
                "tribe_vjepa2_fwd: dim = heads * dim_head, and dim / mlp / K_pad must be multiples of 64");
  TRIBE_REQUIRE(d->pixels && d->w_patch && d->cos_tab && d->sin_tab && (d->depth == 0 || d->layers_host),
                "tribe_vjepa2_fwd: missing parameter pointer");
  TRIBE_REQUIRE(((uintptr_t)workspace % 256) == 0 && workspace_bytes >= tribe_vjepa2_workspace_bytes(d),
                "tribe_vjepa2_fwd: workspace too small or misaligned");
  const VitPlan p = vit_plan(d);
  char* w = (char*)workspace;
  float* x = (float*)w; w += p.x_b;
  uint16_t* xn = (uint16_t*)w; w += p.xn_b;
  uint16_t* qkv = (uint16_t*)w; w += p.qkv_b;
  uint16_t* ao = (uint16_t*)w; w += p.ao_b;
  uint16_t* act = (uint16_t*)w;
  uint16_t* col = (uint16_t*)w;
  uint8_t* q8 = (uint8_t*)(w + (p.act_b > p.col_b ? p.act_b : p.col_b));
  const int64_t M = p.M, dim = d->dim, BD = d->B * dim;
  TRIBE_REQUIRE(!d->fp8_host || (d->dim % 128 == 0 && d->mlp % 128 == 0), "tribe_vjepa2_fwd: the fp8 path needs dim and mlp to be multiples of 128");
  TRIBE_REQUIRE(!(d->fp8_host && d->amax_out), "tribe_vjepa2_fwd: calibrate (amax_out) on the bf16 path, not together with fp8_host");

  int rc = tribe_im2col3d_fwd(d->pixels, d->B, d->frames, d->chans, d->height, d->width, d->tubelet, d->patch, col, d->K_pad, stream);
  if (rc) return rc;
  tribe_gemm_desc g = gemm_zero();
  g.M = M; g.N = dim; g.K = d->K_pad;
  g.A = col; g.lda = d->K_pad; g.B = d->w_patch; g.ldb = d->K_pad;
  g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32; g.role = TRIBE_ROLE_PROJECTOR;
  if (d->b_patch) { g.bias = d->b_patch; g.bias_mode = TRIBE_BIAS_COL; }
  rc = tribe_gemm_bf16(&g, stream);
  if (rc) return rc;
  rc = tribe_segment_mean_fwd(x, d->B, p.tokens, dim, nullptr, nullptr, states, dim, stream);
  if (rc) return rc;

  for (int l = 0; l < d->depth; ++l) {
    const tribe_vit_layer& L = d->layers_host[l];
    TRIBE_REQUIRE(L.norm1_w && L.w_qkv && L.w_proj && L.norm2_w && L.w_fc1 && L.w_fc2, "tribe_vjepa2_fwd: layer %d has a null parameter", l);
    bool q_in = false;
    rc = extractor_norm_fp8(d->fp8_host, l, 0, x, M, dim, L.norm1_w, L.norm1_b, 1, d->ln_eps, q8, stream, &q_in);
    if (!rc && !q_in) rc = tribe_layernorm_fwd(x, M, dim, L.norm1_w, L.norm1_b, d->ln_eps, xn, TRIBE_BF16, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = 3 * dim; g.K = dim;
    g.A = xn; g.lda = dim; g.ldb = dim;
    g.C = qkv; g.ldc = 3 * dim; g.c_dtype = TRIBE_BF16; g.role = TRIBE_ROLE_QKV;
    if (L.b_qkv) { g.bias = L.b_qkv; g.bias_mode = TRIBE_BIAS_COL; }
    rc = extractor_linear("tribe_vjepa2_fwd", d->fp8_host, d->amax_out, l, 0, g, L.w_qkv, q8, stream, q_in);
    if (rc) return rc;
    rc = tribe_rotary_fwd(qkv, M, p.tokens, 3 * dim, 2 * d->heads, d->dim_head, d->dim_head, d->cos_tab, d->sin_tab, 2, stream);
    if (rc) return rc;
    tribe_attention_desc a;
    a.q = qkv; a.k = qkv + dim; a.v = qkv + 2 * dim;
    a.ld_q = a.ld_k = a.ld_v = 3 * dim;
    a.out = ao; a.ld_out = dim;
    a.B = d->B; a.T = p.tokens; a.heads_q = d->heads; a.heads_kv = d->heads; a.dim_head = d->dim_head; a.causal = 0;
    a.scale = 1.0f / sqrtf((float)d->dim_head);
    a.rel_qe = nullptr; a.ld_rel_qe = 0; a.rel_stride_h = 0; a.rel_left = a.rel_right = 0; a.lse = nullptr;
    rc = tribe_attention_fwd_ex(&a, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = dim; g.K = dim;
    g.A = ao; g.lda = dim; g.ldb = dim;
    g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32; g.res = x; g.ldres = dim; g.role = TRIBE_ROLE_OUT_PROJ;
    if (L.b_proj) { g.bias = L.b_proj; g.bias_mode = TRIBE_BIAS_COL; }
    rc = extractor_linear("tribe_vjepa2_fwd", d->fp8_host, d->amax_out, l, 1, g, L.w_proj, q8, stream);
    if (rc) return rc;
    rc = extractor_norm_fp8(d->fp8_host, l, 2, x, M, dim, L.norm2_w, L.norm2_b, 1, d->ln_eps, q8, stream, &q_in);
    if (!rc && !q_in) rc = tribe_layernorm_fwd(x, M, dim, L.norm2_w, L.norm2_b, d->ln_eps, xn, TRIBE_BF16, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = d->mlp; g.K = dim;
    g.A = xn; g.lda = dim; g.ldb = dim;
    g.C = act; g.ldc = d->mlp; g.c_dtype = TRIBE_BF16; g.act = TRIBE_ACT_GELU; g.role = TRIBE_ROLE_FF1;
    if (L.b_fc1) { g.bias = L.b_fc1; g.bias_mode = TRIBE_BIAS_COL; }
    rc = extractor_linear("tribe_vjepa2_fwd", d->fp8_host, d->amax_out, l, 2, g, L.w_fc1, q8, stream, q_in);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = dim; g.K = d->mlp;
    g.A = act; g.lda = d->mlp; g.ldb = d->mlp;
    g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32; g.res = x; g.ldres = dim; g.role = TRIBE_ROLE_FF2;
    if (L.b_fc2) { g.bias = L.b_fc2; g.bias_mode = TRIBE_BIAS_COL; }
    rc = extractor_linear("tribe_vjepa2_fwd", d->fp8_host, d->amax_out, l, 3, g, L.w_fc2, q8, stream);
    if (rc) return rc;
    rc = tribe_segment_mean_fwd(x, d->B, p.tokens, dim, nullptr, nullptr, states + (int64_t)(l + 1) * BD, dim, stream);
    if (rc) return rc;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// tribe_w2vbert_fwd: transformers Wav2Vec2BertModel (data_utils/features/audio.py:253-263 -> modeling_wav2vec2_bert.py):
// feature projection (LayerNorm + Linear), then depth x conformer block [half-step FFN (swish) -> self-attention with
// "relative_key" position bias -> convolution module (pointwise + GLU, causal depthwise k=31, LayerNorm, swish,
// pointwise) -> half-step FFN -> LayerNorm]; every hidden state is resampled along time by row gather
// (F.interpolate nearest, audio.py:163-171).
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct W2vPlan {
  int64_t M, qe_ld;
  size_t x_b, xn_b, wide_b, qe_b, ao_b, glu_b, feat_b, featp_b, q8_b;
};
inline W2vPlan w2v_plan(const tribe_w2vbert_desc* d) {
  W2vPlan p;
  p.M = d->B * d->T;
  const int npos = d->rel_left + d->rel_right + 1;
  p.qe_ld = (int64_t)d->heads * ((npos + 7) / 8 * 8);
  p.x_b = align256((size_t)p.M * d->dim * 4);
  p.xn_b = align256((size_t)p.M * d->dim * 2);
  const int64_t wide = d->inter > 3 * d->dim ? d->inter : 3 * d->dim;
  p.wide_b = align256((size_t)p.M * wide * 2);
  p.q8_b = d->fp8_host ? align256((size_t)p.M * (d->inter > d->dim ? d->inter : d->dim)) : 0;   // e4m3 staging of one GEMM input
  p.qe_b = align256((size_t)p.M * p.qe_ld * 4);
  p.ao_b = align256((size_t)p.M * d->dim * 2);
  p.glu_b = align256((size_t)p.M * d->dim * 2);
  p.feat_b = align256((size_t)p.M * d->feat_dim * 4);
  p.featp_b = align256((size_t)p.M * d->feat_pad * 2);
  return p;
}
}  // namespace

extern "C" size_t tribe_w2vbert_workspace_bytes(const tribe_w2vbert_desc* d) {
  if (!d || d->B <= 0 || d->T <= 0) return 0;
  const W2vPlan p = w2v_plan(d);
  return p.x_b + p.xn_b + p.wide_b + p.qe_b + p.ao_b + p.glu_b + p.feat_b + p.featp_b + p.q8_b;
}

extern "C" int tribe_w2vbert_fwd(const tribe_w2vbert_desc* d, float* states, void* workspace, size_t workspace_bytes, void* stream) {
  TRIBE_REQUIRE(d && states && workspace, "tribe_w2vbert_fwd: null pointer");
  TRIBE_REQUIRE(d->B > 0 && d->T > 0 && d->dim > 0 && d->depth >= 0 && d->heads > 0 && d->inter > 0 && d->n_out > 0, "tribe_w2vbert_fwd: bad shape");
  TRIBE_REQUIRE(d->heads * d->dim_head == d->dim && d->dim_head == 64, "tribe_w2vbert_fwd: head size must be 64 (relative_key attention kernel)");
  TRIBE_REQUIRE(d->dim % 64 == 0 && d->inter % 64 == 0 && d->feat_pad % 64 == 0 && d->feat_pad >= d->feat_dim && d->feat_dim % 4 == 0,
                "tribe_w2vbert_fwd: dim / inter / feat_pad must be multiples of 64");
  TRIBE_REQUIRE(d->features && d->fp_ln_w && d->fp_ln_b && d->w_fp && d->out_index && (d->depth == 0 || d->layers_host),
                "tribe_w2vbert_fwd: missing parameter pointer");
  TRIBE_REQUIRE(((uintptr_t)workspace % 256) == 0 && workspace_bytes >= tribe_w2vbert_workspace_bytes(d),
                "tribe_w2vbert_fwd: workspace too small or misaligned");
  const W2vPlan p = w2v_plan(d);
  char* w = (char*)workspace;
  float* x = (float*)w; w += p.x_b;
  uint16_t* xn = (uint16_t*)w; w += p.xn_b;
  uint16_t* wide = (uint16_t*)w; w += p.wide_b;   // FFN hidden  |  qkv
  float* qe = (float*)w; w += p.qe_b;
  uint16_t* ao = (uint16_t*)w; w += p.ao_b;
  uint16_t* glu = (uint16_t*)w; w += p.glu_b;
  float* feat = (float*)w; w += p.feat_b;
  uint16_t* featp = (uint16_t*)w; w += p.featp_b;
  uint8_t* q8 = (uint8_t*)w;
  TRIBE_REQUIRE(!d->fp8_host || (d->dim % 128 == 0 && d->inter % 128 == 0), "tribe_w2vbert_fwd: the fp8 path needs dim and inter to be multiples of 128");
  TRIBE_REQUIRE(!(d->fp8_host && d->amax_out), "tribe_w2vbert_fwd: calibrate (amax_out) on the bf16 path, not together with fp8_host");
  const int64_t M = p.M, dim = d->dim;
  const int64_t state_sz = d->B * d->n_out * dim;
  const int npos = d->rel_left + d->rel_right + 1;
  const int qe_stride_h = (npos + 7) / 8 * 8;

  // feature projection: LayerNorm(feat_dim) -> Linear
  int rc = tribe_layernorm_fwd(d->features, M, d->feat_dim, d->fp_ln_w, d->fp_ln_b, d->ln_eps, feat, TRIBE_F32, stream);
  if (rc) return rc;
  rc = tribe_pack_weight_bf16(feat, M, d->feat_dim, d->feat_dim, featp, M, d->feat_pad, stream);  // cast + zero-pad K
  if (rc) return rc;
  tribe_gemm_desc g = gemm_zero();
  g.M = M; g.N = dim; g.K = d->feat_pad;
  g.A = featp; g.lda = d->feat_pad; g.B = d->w_fp; g.ldb = d->feat_pad;
  g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32; g.role = TRIBE_ROLE_PROJECTOR;
  if (d->b_fp) { g.bias = d->b_fp; g.bias_mode = TRIBE_BIAS_COL; }
  rc = tribe_gemm_bf16(&g, stream);
  if (rc) return rc;
  rc = tribe_gather_rows_fwd(x, d->B, d->T, dim, d->out_index, d->n_out, states, stream);
  if (rc) return rc;

  // half-step feed-forward x += 0.5 * (swish(LN(x) W_in^T + b_in) W_out^T + b_out); `which` = 0 (ffn1) or 2 (ffn2) selects the pair of e4m3
  // weights / scales of the layer when fp8_host is set (the LayerNorm then writes the e4m3 operand directly)
  auto ffn = [&](int l, int which, const float* ln_w, const float* ln_b, const uint16_t* w_in, const float* b_in, const uint16_t* w_out,
                 const float* b_out_half) -> int {
    bool q_in = false;
    int r = extractor_norm_fp8(d->fp8_host, l, which, x, M, dim, ln_w, ln_b, 1, d->ln_eps, q8, stream, &q_in);
    if (!r && !q_in) r = tribe_layernorm_fwd(x, M, dim, ln_w, ln_b, d->ln_eps, xn, TRIBE_BF16, stream);
    if (r) return r;
    tribe_gemm_desc q = gemm_zero();
    q.M = M; q.N = d->inter; q.K = dim;
    q.A = xn; q.lda = dim; q.ldb = dim;
    q.C = wide; q.ldc = d->inter; q.c_dtype = TRIBE_BF16; q.act = TRIBE_ACT_SILU; q.role = TRIBE_ROLE_FF1;
    if (b_in) { q.bias = b_in; q.bias_mode = TRIBE_BIAS_COL; }
    r = extractor_linear("tribe_w2vbert_fwd", d->fp8_host, d->amax_out, l, which, q, w_in, q8, stream, q_in);
    if (r) return r;
    q = gemm_zero();  // x = 0.5 * (h W^T + b) + x
    q.M = M; q.N = dim; q.K = d->inter;
    q.A = wide; q.lda = d->inter; q.ldb = d->inter;
    q.C = x; q.ldc = dim; q.c_dtype = TRIBE_F32; q.alpha = 0.5f; q.res = x; q.ldres = dim; q.role = TRIBE_ROLE_FF2;
    if (b_out_half) { q.bias = b_out_half; q.bias_mode = TRIBE_BIAS_COL; }
    return extractor_linear("tribe_w2vbert_fwd", d->fp8_host, d->amax_out, l, which + 1, q, w_out, q8, stream);
  };

  for (int l = 0; l < d->depth; ++l) {
    const tribe_conformer_layer& L = d->layers_host[l];
    TRIBE_REQUIRE(L.ffn1_ln_w && L.w_ffn1_in && L.w_ffn1_out && L.attn_ln_w && L.w_qkv && L.dist_emb && L.w_attn_out && L.conv_ln_w &&
                      L.w_pw1 && L.w_dw_kc && L.dw_ln_w && L.w_pw2 && L.ffn2_ln_w && L.w_ffn2_in && L.w_ffn2_out && L.final_ln_w,
                  "tribe_w2vbert_fwd: layer %d has a null parameter", l);
    // 1. half-step feed-forward
    rc = ffn(l, 0, L.ffn1_ln_w, L.ffn1_ln_b, L.w_ffn1_in, L.b_ffn1_in, L.w_ffn1_out, L.b_ffn1_out_half);
    if (rc) return rc;
    // 2. self-attention with relative_key bias
    rc = tribe_layernorm_fwd(x, M, dim, L.attn_ln_w, L.attn_ln_b, d->ln_eps, xn, TRIBE_BF16, stream);
    if (rc) return rc;
    uint16_t* qkv = wide;
    g = gemm_zero();
    g.M = M; g.N = 3 * dim; g.K = dim;
    g.A = xn; g.lda = dim; g.B = L.w_qkv; g.ldb = dim;
    g.C = qkv; g.ldc = 3 * dim; g.c_dtype = TRIBE_BF16; g.role = TRIBE_ROLE_QKV;
    if (L.b_qkv) { g.bias = L.b_qkv; g.bias_mode = TRIBE_BIAS_COL; }
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
    g = gemm_zero();  // qe[row][h][p] = q[row][h][:] . dist_emb[p][:]
    g.M = M; g.N = npos; g.K = d->dim_head; g.batch0 = d->heads;
    g.A = qkv; g.lda = 3 * dim; g.sA0 = d->dim_head;
    g.B = L.dist_emb; g.ldb = d->dim_head; g.sB0 = 0;
    g.C = qe; g.ldc = p.qe_ld; g.sC0 = qe_stride_h; g.c_dtype = TRIBE_F32; g.role = TRIBE_ROLE_ATTN_SCORES;
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
    tribe_attention_desc a;
    a.q = qkv; a.k = qkv + dim; a.v = qkv + 2 * dim;
    a.ld_q = a.ld_k = a.ld_v = 3 * dim;
    a.out = ao; a.ld_out = dim;
    a.B = d->B; a.T = d->T; a.heads_q = d->heads; a.heads_kv = d->heads; a.dim_head = d->dim_head; a.causal = 0;
    a.scale = 1.0f / sqrtf((float)d->dim_head);
    a.rel_qe = qe; a.ld_rel_qe = p.qe_ld; a.rel_stride_h = qe_stride_h; a.rel_left = d->rel_left; a.rel_right = d->rel_right; a.lse = nullptr;
    rc = tribe_attention_fwd_ex(&a, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = dim; g.K = dim;
    g.A = ao; g.lda = dim; g.B = L.w_attn_out; g.ldb = dim;
    g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32; g.res = x; g.ldres = dim; g.role = TRIBE_ROLE_OUT_PROJ;
    if (L.b_attn_out) { g.bias = L.b_attn_out; g.bias_mode = TRIBE_BIAS_COL; }
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
    // 3. convolution module
    rc = tribe_layernorm_fwd(x, M, dim, L.conv_ln_w, L.conv_ln_b, d->ln_eps, xn, TRIBE_BF16, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = 2 * dim; g.K = dim;
    g.A = xn; g.lda = dim; g.B = L.w_pw1; g.ldb = dim;
    g.C = glu; g.ldc = dim; g.c_dtype = TRIBE_BF16; g.act = TRIBE_ACT_GLU; g.role = TRIBE_ROLE_GENERIC;
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
    rc = tribe_dwconv_ln_swish_fwd(glu, d->B, d->T, (int32_t)dim, d->conv_kernel, L.w_dw_kc, L.dw_ln_w, L.dw_ln_b, d->ln_eps, xn, stream);
    if (rc) return rc;
    g = gemm_zero();
    g.M = M; g.N = dim; g.K = dim;
    g.A = xn; g.lda = dim; g.B = L.w_pw2; g.ldb = dim;
    g.C = x; g.ldc = dim; g.c_dtype = TRIBE_F32; g.res = x; g.ldres = dim; g.role = TRIBE_ROLE_GENERIC;
    rc = tribe_gemm_bf16(&g, stream);
    if (rc) return rc;
    // 4. half-step feed-forward, then the block's final LayerNorm (in place on the residual stream)
    rc = ffn(l, 2, L.ffn2_ln_w, L.ffn2_ln_b, L.w_ffn2_in, L.b_ffn2_in, L.w_ffn2_out, L.b_ffn2_out_half);
    if (rc) return rc;
    rc = tribe_layernorm_fwd(x, M, dim, L.final_ln_w, L.final_ln_b, d->ln_eps, x, TRIBE_F32, stream);
    if (rc) return rc;
    rc = tribe_gather_rows_fwd(x, d->B, d->T, dim, d->out_index, d->n_out, states + (int64_t)(l + 1) * state_sz, stream);
    if (rc) return rc;
  }
  return 0;
}
