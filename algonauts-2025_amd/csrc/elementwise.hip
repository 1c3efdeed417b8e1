// HBM-bound kernels of the TRIBE encode path: packing / layout changes, ScaleNorm,
// rotary, softmax, V transpose, adaptive average pool.  All are pure streaming
// kernels (roofline: HBM bandwidth); loads/stores are 8-16 B per lane where the
// layout allows.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------
// f32 [rows, cols] -> bf16 [rows_pad, cols_pad] (zero padded); 8 outputs per thread
// ---------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t ld_src,
                                   unsigned short* __restrict__ dst, int64_t rows_pad, int64_t cols_pad) {
  const int64_t chunks = cols_pad >> 3;
  const int64_t total = rows_pad * chunks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / chunks, c = (i - r * chunks) << 3;
    u16x8_t o;
    if (r < rows && c + 8 <= cols && (ld_src & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
      const float4 v0 = *(const float4*)(src + r * ld_src + c);
      const float4 v1 = *(const float4*)(src + r * ld_src + c + 4);
      o[0] = f32_to_bf16(v0.x); o[1] = f32_to_bf16(v0.y); o[2] = f32_to_bf16(v0.z); o[3] = f32_to_bf16(v0.w);
      o[4] = f32_to_bf16(v1.x); o[5] = f32_to_bf16(v1.y); o[6] = f32_to_bf16(v1.z); o[7] = f32_to_bf16(v1.w);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (r < rows && c + k < cols) ? f32_to_bf16(src[r * ld_src + c + k]) : 0;
    }
    *(u16x8_t*)(dst + r * cols_pad + c) = o;
  }
}

// the unpadded, contiguous case (every weight of the encoder: cols % 64 == 0) is a flat f32 -> bf16 cast: 32 elements per thread and
// trip, eight 16-byte nontemporal loads in flight, no index arithmetic (the general kernel above ran at 2.6 TB/s over the 5.7 GB of
// weights a training step re-packs)
__global__ __launch_bounds__(256) void cast_flat_kernel(const float* __restrict__ src, int64_t n32, unsigned short* __restrict__ dst) {
  // a workgroup converts 8192 consecutive elements per trip: load k of lane l is float4 number 256 k + l of the span, so every
  // wave-instruction reads (and every store writes) one contiguous run
  const int64_t spans = (n32 * 32 + 8191) / 8192, n4 = n32 * 8;
  for (int64_t sp = blockIdx.x; sp < spans; sp += gridDim.x) {
    const int64_t base4 = sp * 2048 + threadIdx.x;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (base4 + 256 * k < n4) v[k] = load_nt_f4(src + (base4 + 256 * k) * 4);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (base4 + 256 * k >= n4) continue;
      u16x4_t o;
      o[0] = f32_to_bf16(v[k].x); o[1] = f32_to_bf16(v[k].y); o[2] = f32_to_bf16(v[k].z); o[3] = f32_to_bf16(v[k].w);
      *(u16x4_t*)(dst + (base4 + 256 * k) * 4) = o;
    }
  }
}

// ---------------------------------------------------------------------------------
// generic tiled transpose+cast:  out[z][j][i] (bf16, ld_out, zero padded to i_pad x j_pad)
//   = reduce_l in[z][l][i][j]   with in element strides (s_z, s_l, s_i, 1), j contiguous.
// 64x64 tile through LDS: reads coalesced along j, writes coalesced along i.
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float ld_as_f32(const T* p);
template <>
__device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_as_f32<double>(const double* p) { return (float)*p; }
template <>
__device__ __forceinline__ float ld_as_f32<unsigned short>(const unsigned short* p) { return bf16_to_f32(*p); }

template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const T* __restrict__ in, int64_t s_z, int64_t s_l, int64_t s_i,
                                                             int64_t L, float l_scale, int64_t I, int64_t J,
                                                             unsigned short* __restrict__ out, int64_t so_z, int64_t ld_out,
                                                             int64_t I_pad, int64_t J_pad) {
  __shared__ float tile[64][65];
  const int64_t z = blockIdx.z;
  const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const T* src = in + z * s_z;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int64_t i = i0 + r, j = j0 + tx;
    float v = 0.f;
    if (i < I && j < J) {
      const T* p = src + i * s_i + j;
      for (int64_t l = 0; l < L; ++l) v += ld_as_f32<T>(p + l * s_l);
      v *= l_scale;
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  unsigned short* dst = out + z * so_z;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int64_t j = j0 + r, i = i0 + tx;
    if (j < J_pad && i < I_pad) dst[j * ld_out + i] = f32_to_bf16(tile[tx][r]);
  }
}

// ---------------------------------------------------------------------------------
// ScaleNorm: one wave per row;  y = x * (gain / max(||x||, eps))
// ---------------------------------------------------------------------------------
template <int OUT_BF16>
__global__ __launch_bounds__(256) void scalenorm_kernel(const float* __restrict__ x, int64_t rows, int64_t dim,
                                                        const float* __restrict__ g, float gain_scale, float eps,
                                                        void* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + row * dim);
  const int64_t n4 = dim >> 2;
  float ss = 0.f;
  for (int64_t i = lane; i < n4; i += 64) {
    const float4 v = xr[i];
    ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  ss = wave_sum(ss);
  const float scale = (g[0] * gain_scale) / fmaxf(sqrtf(ss), eps);
  for (int64_t i = lane; i < n4; i += 64) {
    const float4 v = xr[i];  // second touch is an L2 hit (12 KiB row)
    if (OUT_BF16) {
      u16x4_t o;
      o[0] = f32_to_bf16(v.x * scale); o[1] = f32_to_bf16(v.y * scale);
      o[2] = f32_to_bf16(v.z * scale); o[3] = f32_to_bf16(v.w * scale);
      ((u16x4_t*)((unsigned short*)y + row * dim))[i] = o;
    } else {
      ((float4*)((float*)y + row * dim))[i] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }
  }
}

// Same, with the row held in registers between the reduction and the scaling (dim = 256 * NV, NV float4 per lane): one
// HBM read of x instead of two -- the counters showed the second touch of the two-pass kernel mostly missing L2
// (1.61 GB fetched for a 0.81 GB input) -- and all NV loads of a row in flight at once.
template <int OUT_BF16, int NV>
__global__ __launch_bounds__(256) void scalenorm_reg_kernel(const float* __restrict__ x, int64_t rows, const float* __restrict__ g,
                                                            float gain_scale, float eps, void* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  constexpr int64_t dim = 256 * NV;
  const float4* xr = (const float4*)(x + row * dim);
  float4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = xr[lane + 64 * k];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) ss += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
  ss = wave_sum(ss);
  const float scale = (g[0] * gain_scale) / fmaxf(sqrtf(ss), eps);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (OUT_BF16) {
      u16x4_t o;
      o[0] = f32_to_bf16(v[k].x * scale); o[1] = f32_to_bf16(v[k].y * scale);
      o[2] = f32_to_bf16(v[k].z * scale); o[3] = f32_to_bf16(v[k].w * scale);
      ((u16x4_t*)((unsigned short*)y + row * dim))[lane + 64 * k] = o;
    } else {
      ((float4*)((float*)y + row * dim))[lane + 64 * k] = make_float4(v[k].x * scale, v[k].y * scale, v[k].z * scale, v[k].w * scale);
    }
  }
}

// ---------------------------------------------------------------------------------
// partial rotary embedding, in place on the q and k sections of a fused qkv buffer
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rotary_kernel(unsigned short* __restrict__ x, int64_t rows, int64_t T, int64_t row_stride,
                                                     int n_heads, int dim_head, int rot_dim, const float* __restrict__ cos_tab,
                                                     const float* __restrict__ sin_tab, int interleaved) {
  const int half = rot_dim >> 1;
  const int items_per_head = half >> 2;  // each item rotates 4 pairs (8 elements)
  const int64_t total = rows * n_heads * items_per_head;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int it = (int)(t % items_per_head); t /= items_per_head;
    const int h = (int)(t % n_heads);
    const int64_t row = t / n_heads;
    const int64_t pos = row % T;
    unsigned short* base = x + row * row_stride + (int64_t)h * dim_head;
    if (interleaved == 2) {
      // per-ELEMENT tables [T, rot_dim] with interleaved pairs: out[2p] = a C[2p] - b S[2p], out[2p+1] = b C[2p+1] + a S[2p+1]
      // (VJEPA2 rotate_queries_or_keys tiles its frequencies, so the two elements of a pair see different angles)
      const float4 c0 = *(const float4*)(cos_tab + pos * rot_dim + it * 8), c1 = *(const float4*)(cos_tab + pos * rot_dim + it * 8 + 4);
      const float4 s0 = *(const float4*)(sin_tab + pos * rot_dim + it * 8), s1 = *(const float4*)(sin_tab + pos * rot_dim + it * 8 + 4);
      const float ce[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, se[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      u16x8_t v = *(u16x8_t*)(base + it * 8);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float a = bf16_to_f32(v[2 * p]), b = bf16_to_f32(v[2 * p + 1]);
        v[2 * p] = f32_to_bf16(a * ce[2 * p] - b * se[2 * p]);
        v[2 * p + 1] = f32_to_bf16(b * ce[2 * p + 1] + a * se[2 * p + 1]);
      }
      *(u16x8_t*)(base + it * 8) = v;
      continue;
    }
    const float4 c = *(const float4*)(cos_tab + pos * half + it * 4);
    const float4 s = *(const float4*)(sin_tab + pos * half + it * 4);
    const float cs[4] = {c.x, c.y, c.z, c.w}, sn[4] = {s.x, s.y, s.z, s.w};
    if (interleaved) {
      u16x8_t v = *(u16x8_t*)(base + it * 8);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float a = bf16_to_f32(v[2 * p]), b = bf16_to_f32(v[2 * p + 1]);
        v[2 * p] = f32_to_bf16(a * cs[p] - b * sn[p]);
        v[2 * p + 1] = f32_to_bf16(b * cs[p] + a * sn[p]);
      }
      *(u16x8_t*)(base + it * 8) = v;
    } else {
      u16x4_t lo = *(u16x4_t*)(base + it * 4), hi = *(u16x4_t*)(base + half + it * 4);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float a = bf16_to_f32(lo[p]), b = bf16_to_f32(hi[p]);
        lo[p] = f32_to_bf16(a * cs[p] - b * sn[p]);
        hi[p] = f32_to_bf16(b * cs[p] + a * sn[p]);
      }
      *(u16x4_t*)(base + it * 4) = lo;
      *(u16x4_t*)(base + half + it * 4) = hi;
    }
  }
}

// ---------------------------------------------------------------------------------
// RMSNorm / LayerNorm: one wave per row (f32 in, f32 statistics), bf16 or f32 out; OUT_BF16 == 2: e4m3 bytes of the
// bf16-rounded result times q_inv_scale, saturating (= the bf16 norm followed by tribe_quantize_fp8_fwd, bit for bit, in one pass)
// ---------------------------------------------------------------------------------
template <int OUT_BF16, int LAYERNORM>
__global__ __launch_bounds__(256) void rowstat_norm_kernel(const float* __restrict__ x, int64_t rows, int64_t dim,
                                                           const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                           void* __restrict__ y, float q_inv_scale = 0.f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + row * dim);
  const int64_t n4 = dim >> 2;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t i = lane; i < n4; i += 64) {
    const float4 v = xr[i];
    s1 += v.x + v.y + v.z + v.w;
    s2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  float mean = 0.f, rstd;
  if (LAYERNORM) {
    mean = s1 / (float)dim;
    float var = 0.f;  // second pass for the centred variance (row is L2-resident)
    for (int64_t i = lane; i < n4; i += 64) {
      const float4 v = xr[i];
      const float a = v.x - mean, c = v.y - mean, d = v.z - mean, e = v.w - mean;
      var += a * a + c * c + d * d + e * e;
    }
    var = wave_sum(var) / (float)dim;
    rstd = rsqrtf(var + eps);
  } else {
    rstd = rsqrtf(s2 / (float)dim + eps);
  }
  for (int64_t i = lane; i < n4; i += 64) {
    const float4 v = xr[i];
    const float4 g = ((const float4*)w)[i];
    float o0 = (v.x - mean) * rstd * g.x, o1 = (v.y - mean) * rstd * g.y, o2 = (v.z - mean) * rstd * g.z, o3 = (v.w - mean) * rstd * g.w;
    if (LAYERNORM && b) {
      const float4 bb = ((const float4*)b)[i];
      o0 += bb.x; o1 += bb.y; o2 += bb.z; o3 += bb.w;
    }
    if (OUT_BF16 == 2) {
      float q[4] = {o0, o1, o2, o3};
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = fminf(fmaxf(bf16_to_f32(f32_to_bf16(q[j])) * q_inv_scale, -448.f), 448.f);
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], pk, true);
      ((int*)((unsigned char*)y + row * dim))[i] = pk;
    } else if (OUT_BF16) {
      u16x4_t o;
      o[0] = f32_to_bf16(o0); o[1] = f32_to_bf16(o1); o[2] = f32_to_bf16(o2); o[3] = f32_to_bf16(o3);
      ((u16x4_t*)((unsigned short*)y + row * dim))[i] = o;
    } else {
      ((float4*)((float*)y + row * dim))[i] = make_float4(o0, o1, o2, o3);
    }
  }
}

// The same norm with the row held in registers (NV float4 per lane, dim <= 256 * NV): x is read from memory ONCE instead of three
// times (statistics, centred variance, output); same per-lane accumulation order, so the results are bit-identical to the kernel above.
template <int OUT_BF16, int LAYERNORM, int NV>
__global__ __launch_bounds__(256) void rowstat_norm_reg_kernel(const float* __restrict__ x, int64_t rows, int64_t dim,
                                                               const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                               void* __restrict__ y, float q_inv_scale = 0.f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4* xr = (const float4*)(x + row * dim);
  const int n4 = (int)(dim >> 2);
  float4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = lane + 64 * j;
    v[j] = i < n4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (lane + 64 * j < n4) {
      s1 += v[j].x + v[j].y + v[j].z + v[j].w;
      s2 += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  float mean = 0.f, rstd;
  if (LAYERNORM) {
    mean = s1 / (float)dim;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (lane + 64 * j < n4) {
        const float a = v[j].x - mean, c = v[j].y - mean, d = v[j].z - mean, e = v[j].w - mean;
        var += a * a + c * c + d * d + e * e;
      }
    }
    var = wave_sum(var) / (float)dim;
    rstd = rsqrtf(var + eps);
  } else {
    rstd = rsqrtf(s2 / (float)dim + eps);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = lane + 64 * j;
    if (i >= n4) continue;
    const float4 g = ((const float4*)w)[i];
    float o0 = (v[j].x - mean) * rstd * g.x, o1 = (v[j].y - mean) * rstd * g.y, o2 = (v[j].z - mean) * rstd * g.z, o3 = (v[j].w - mean) * rstd * g.w;
    if (LAYERNORM && b) {
      const float4 bb = ((const float4*)b)[i];
      o0 += bb.x; o1 += bb.y; o2 += bb.z; o3 += bb.w;
    }
    if (OUT_BF16 == 2) {
      float q[4] = {o0, o1, o2, o3};
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = fminf(fmaxf(bf16_to_f32(f32_to_bf16(q[k])) * q_inv_scale, -448.f), 448.f);
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], pk, true);
      ((int*)((unsigned char*)y + row * dim))[i] = pk;
    } else if (OUT_BF16) {
      u16x4_t o;
      o[0] = f32_to_bf16(o0); o[1] = f32_to_bf16(o1); o[2] = f32_to_bf16(o2); o[3] = f32_to_bf16(o3);
      ((u16x4_t*)((unsigned short*)y + row * dim))[i] = o;
    } else {
      ((float4*)((float*)y + row * dim))[i] = make_float4(o0, o1, o2, o3);
    }
  }
}

// launch the register-resident form when the row fits 12 float4 per lane (dim <= 3072) and is 16-byte aligned, else the three-pass walk
template <int OUT_BF16, int LAYERNORM>
void launch_rowstat_norm(const float* x, int64_t rows, int64_t dim, const float* w, const float* b, float eps, void* y, float q_inv_scale,
                         hipStream_t s) {
  dim3 grid((unsigned)((rows + 3) / 4));
  const int nv = (int)((dim / 4 + 63) / 64);
  // the register-resident kernel stores float4 / u16x4 / packed fp8x4 per lane: y needs 16 / 8 / 4-byte alignment (row pitch = dim elements)
  const uintptr_t y_align = OUT_BF16 == 0 ? 16 : (OUT_BF16 == 1 ? 8 : 4);
  const bool aligned = ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && (!b || ((uintptr_t)b % 16) == 0) && ((uintptr_t)y % y_align) == 0 &&
                       dim % 4 == 0;
#define TRIBE_RSN(NV) hipLaunchKernelGGL((rowstat_norm_reg_kernel<OUT_BF16, LAYERNORM, NV>), grid, dim3(256), 0, s, x, rows, dim, w, b, eps, y, q_inv_scale)
  if (aligned && nv <= 4) TRIBE_RSN(4);
  else if (aligned && nv <= 6) TRIBE_RSN(6);
  else if (aligned && nv <= 8) TRIBE_RSN(8);
  else if (aligned && nv <= 12) TRIBE_RSN(12);
  else hipLaunchKernelGGL((rowstat_norm_kernel<OUT_BF16, LAYERNORM>), grid, dim3(256), 0, s, x, rows, dim, w, b, eps, y, q_inv_scale);
#undef TRIBE_RSN
}

// Conv3d(stride == kernel) patch embedding as a GEMM: unfold pixels [B, F, C, H, W] into rows of
// K = C * tub * p * p (k = ((c * tub + dt) * p + dy) * p + dx, the Conv3d weight's own flattening), 8 outputs per thread
__global__ __launch_bounds__(256) void im2col3d_kernel(const float* __restrict__ pix, int64_t B, int F, int Cc, int H, int W, int tub,
                                                       int p, unsigned short* __restrict__ out, int64_t K_pad) {
  const int gh = H / p, gw = W / p, gf = F / tub;
  const int64_t tokens = (int64_t)gf * gh * gw;
  const int64_t K = (int64_t)Cc * tub * p * p;
  const int64_t chunks = K_pad >> 3;
  const int64_t total = B * tokens * chunks;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / chunks;
    const int64_t k0 = (idx - row * chunks) << 3;
    const int64_t b = row / tokens;
    int64_t tkn = row - b * tokens;
    const int ft = (int)(tkn / (gh * gw)); tkn -= (int64_t)ft * gh * gw;
    const int py = (int)(tkn / gw), px = (int)(tkn - (int64_t)py * gw);
    u16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t k = k0 + e;
      float v = 0.f;
      if (k < K) {
        int64_t r = k;
        const int dx = (int)(r % p); r /= p;
        const int dy = (int)(r % p); r /= p;
        const int dt = (int)(r % tub);
        const int c = (int)(r / tub);
        v = pix[(((b * F + (int64_t)ft * tub + dt) * Cc + c) * H + (int64_t)py * p + dy) * W + (int64_t)px * p + dx];
      }
      o[e] = f32_to_bf16(v);
    }
    *(u16x8_t*)(out + row * K_pad + k0) = o;
  }
}

// causal depthwise Conv1d (left padding K-1) over time + LayerNorm over channels + swish, one workgroup per (b, t) row
// (Wav2Vec2BertConvolutionModule, modeling_wav2vec2_bert.py:214-222).  x bf16 [B*T, C]; w f32 [K, C] (tap-major so that
// a thread's 4 channels are one float4); output bf16 [B*T, C].  The 31 input rows of a window are L2 hits.
template <int MAXG>
__global__ __launch_bounds__(256) void dwconv_ln_swish_kernel(const unsigned short* __restrict__ x, int64_t T, int C, int K,
                                                              const float* __restrict__ w, const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, float eps,
                                                              unsigned short* __restrict__ y) {
  __shared__ float red[8];
  const int64_t row = blockIdx.x;  // b * T + t
  const int64_t t = row % T;
  const int groups = C >> 2;
  float acc[MAXG][4];
  float s1 = 0.f;
#pragma unroll
  for (int gi = 0; gi < MAXG; ++gi) {
    const int cg = threadIdx.x + gi * 256;
    acc[gi][0] = acc[gi][1] = acc[gi][2] = acc[gi][3] = 0.f;
    if (cg < groups) {
      for (int k = 0; k < K; ++k) {
        const int64_t tt = t - (K - 1) + k;
        if (tt < 0) continue;
        const u16x4_t xv = *(const u16x4_t*)(x + (row - t + tt) * C + cg * 4);
        const float4 wv = *(const float4*)(w + (int64_t)k * C + cg * 4);
        acc[gi][0] += bf16_to_f32(xv[0]) * wv.x; acc[gi][1] += bf16_to_f32(xv[1]) * wv.y;
        acc[gi][2] += bf16_to_f32(xv[2]) * wv.z; acc[gi][3] += bf16_to_f32(xv[3]) * wv.w;
      }
      s1 += acc[gi][0] + acc[gi][1] + acc[gi][2] + acc[gi][3];
    }
  }
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  const float mean = block_sum(s1) / (float)C;
  float s2 = 0.f;
#pragma unroll
  for (int gi = 0; gi < MAXG; ++gi)
    if (threadIdx.x + gi * 256 < groups)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = acc[gi][e] - mean; s2 += d * d; }
  const float rstd = rsqrtf(block_sum(s2) / (float)C + eps);
#pragma unroll
  for (int gi = 0; gi < MAXG; ++gi) {
    const int cg = threadIdx.x + gi * 256;
    if (cg < groups) {
      const float4 g = *(const float4*)(ln_w + cg * 4), b = *(const float4*)(ln_b + cg * 4);
      const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
      u16x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = (acc[gi][e] - mean) * rstd * gg[e] + bb[e];
        o[e] = f32_to_bf16(v / (1.0f + __expf(-v)));  // swish
      }
      *(u16x4_t*)(y + row * C + cg * 4) = o;
    }
  }
}

// The same module for TT consecutive time steps per workgroup (K <= 31 taps, C <= 1024): every input row is loaded ONCE and feeds all
// the outputs whose causal window contains it (the one-row kernel above re-reads 31 rows per output: 1.5 GB of L2 traffic for 49 MB of
// input at 8 x 3000 frames), the taps sit in registers, and the LayerNorm statistics of the TT rows are reduced together.
template <int TT, int KMAX>
__global__ __launch_bounds__(256) void dwconv_ln_swish_tile_kernel(const unsigned short* __restrict__ x, int64_t T, int C, int K,
                                                                   const float* __restrict__ w, const float* __restrict__ ln_w,
                                                                   const float* __restrict__ ln_b, float eps, unsigned short* __restrict__ y) {
  __shared__ float red[4][TT];
  const int tiles = (int)((T + TT - 1) / TT);
  const int64_t b = blockIdx.x / tiles;
  const int t0 = (int)(blockIdx.x % tiles) * TT;
  const int cg = threadIdx.x, groups = C >> 2;
  const bool live = cg < groups;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the K taps of this thread's 4 channels, RIGHT-aligned in KMAX slots (slot KMAX - 1 - d = the tap that joins input t - d to output t,
  // w[K - 1 - d]; slots of d >= K are zero), so that every slot index below is a compile-time constant
  float4 wk[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i)
    wk[i] = (live && i >= KMAX - K) ? *(const float4*)(w + (int64_t)(i - (KMAX - K)) * C + cg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc[TT];
#pragma unroll
  for (int j = 0; j < TT; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned short* xb = x + b * T * C + cg * 4;
  // input row t0 + r, r = -(KMAX - 1) .. TT - 1, reaches the outputs t0 + j with 0 <= j - r <= KMAX - 1
#pragma unroll
  for (int r = -(KMAX - 1); r < TT; ++r) {
    const int tin = t0 + r;
    if (tin < 0 || tin >= T || !live) continue;          // causal left padding / past the end of the sequence
    const u16x4_t xv = *(const u16x4_t*)(xb + (int64_t)tin * C);
    const float x0 = bf16_to_f32(xv[0]), x1 = bf16_to_f32(xv[1]), x2 = bf16_to_f32(xv[2]), x3 = bf16_to_f32(xv[3]);
#pragma unroll
    for (int j = 0; j < TT; ++j) {
      const int d = j - r;                                  // compile-time after unrolling
      if (d < 0 || d > KMAX - 1) continue;
      const float4 ww = wk[KMAX - 1 - d];
      acc[j].x += x0 * ww.x; acc[j].y += x1 * ww.y; acc[j].z += x2 * ww.z; acc[j].w += x3 * ww.w;
    }
  }
  // LayerNorm over the C channels of each of the TT rows: wave sums, then the 4 waves through LDS
  auto rows_sum = [&](float (&v)[TT]) {
#pragma unroll
    for (int j = 0; j < TT; ++j) v[j] = wave_sum(v[j]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < TT; ++j) red[wave][j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TT; ++j) v[j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
  };
  float s[TT];
#pragma unroll
  for (int j = 0; j < TT; ++j) s[j] = live ? (acc[j].x + acc[j].y) + (acc[j].z + acc[j].w) : 0.f;
  rows_sum(s);
  float mean[TT];
#pragma unroll
  for (int j = 0; j < TT; ++j) {
    mean[j] = s[j] / (float)C;
    const float a0 = acc[j].x - mean[j], a1 = acc[j].y - mean[j], a2 = acc[j].z - mean[j], a3 = acc[j].w - mean[j];
    s[j] = live ? (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3) : 0.f;
  }
  rows_sum(s);
  if (!live) return;
  const float4 g = *(const float4*)(ln_w + cg * 4), bb = *(const float4*)(ln_b + cg * 4);
#pragma unroll
  for (int j = 0; j < TT; ++j) {
    if (t0 + j >= T) continue;
    const float rstd = rsqrtf(s[j] / (float)C + eps);
    const float v0 = (acc[j].x - mean[j]) * rstd * g.x + bb.x, v1 = (acc[j].y - mean[j]) * rstd * g.y + bb.y;
    const float v2 = (acc[j].z - mean[j]) * rstd * g.z + bb.z, v3 = (acc[j].w - mean[j]) * rstd * g.w + bb.w;
    u16x4_t o;
    o[0] = f32_to_bf16(v0 / (1.0f + __expf(-v0))); o[1] = f32_to_bf16(v1 / (1.0f + __expf(-v1)));
    o[2] = f32_to_bf16(v2 / (1.0f + __expf(-v2))); o[3] = f32_to_bf16(v3 / (1.0f + __expf(-v3)));
    *(u16x4_t*)(y + (b * T + t0 + j) * C + cg * 4) = o;
  }
}

// out[b][i][:] = x[b*T + idx[i]][:]   (F.interpolate(mode="nearest") along time = a row gather, audio.py:163-171)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int64_t T, int64_t dim,
                                                          const int64_t* __restrict__ idx, int64_t n, float* __restrict__ out) {
  const int64_t b = blockIdx.y, i = blockIdx.x;
  int64_t t = idx[i];
  t = t < 0 ? 0 : (t >= T ? T - 1 : t);
  const float4* src = (const float4*)(x + (b * T + t) * dim);
  float4* dst = (float4*)(out + (b * n + i) * dim);
  for (int64_t c = threadIdx.x; c < (dim >> 2); c += blockDim.x) dst[c] = src[c];
}

// embedding gather: one wave per token row
template <typename T>
__global__ __launch_bounds__(256) void embedding_kernel(const T* __restrict__ table, const int64_t* __restrict__ ids, int64_t n,
                                                        int64_t dim, int64_t vocab, float* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  int64_t id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // ids are validated on the host; clamp keeps a bad id from faulting
  const T* src = table + id * dim;
  for (int64_t i = lane; i < dim; i += 64) x[row * dim + i] = ld_as_f32<T>(src + i);
}

// segment mean over time: grid (dim/256, B, slices); every block sums one slice of the window and adds its share with
// one f32 atomic per column (out is zeroed by a memset node first).  A [8192, 1408] state is reduced by 704 blocks
// instead of the 6 a one-block-per-column-group walk would give.
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ x, int64_t T, int64_t dim,
                                                           const int64_t* __restrict__ start, const int64_t* __restrict__ len,
                                                           float* __restrict__ out, int64_t ld_out) {
  const int64_t b = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= dim) return;
  int64_t s = start ? start[b] : 0, n = len ? len[b] : T;
  if (s < 0) s = 0;
  if (s + n > T) n = T - s;
  if (n <= 0) return;
  const int64_t per = (n + gridDim.z - 1) / gridDim.z;
  const int64_t t0 = (int64_t)blockIdx.z * per;
  const int64_t t1 = (t0 + per < n) ? t0 + per : n;
  if (t0 >= t1) return;
  float acc = 0.f;
  const float* p = x + (b * T + s) * dim + c;
  for (int64_t t = t0; t < t1; ++t) acc += p[t * dim];
  atomicAdd(out + b * ld_out + c, acc / (float)n);
}

// the same means, four columns per lane and eight rows in flight (16-byte nontemporal loads: a state is read once): the one-column walk
// above keeps one 4-byte load per lane in flight and read a [8192, 1408] state at 2.1 TB/s (41 of them per ViT-g clip)
__global__ __launch_bounds__(256) void segment_mean4_kernel(const float* __restrict__ x, int64_t T, int64_t dim, const int64_t* __restrict__ start,
                                                            const int64_t* __restrict__ len, float* __restrict__ out, int64_t ld_out) {
  constexpr int U = 8;
  const int64_t b = blockIdx.y;
  const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= dim) return;
  int64_t s = start ? start[b] : 0, n = len ? len[b] : T;
  if (s < 0) s = 0;
  if (s + n > T) n = T - s;
  if (n <= 0) return;
  const int64_t per = (n + gridDim.z - 1) / gridDim.z;
  const int64_t t0 = (int64_t)blockIdx.z * per;
  const int64_t t1 = (t0 + per < n) ? t0 + per : n;
  if (t0 >= t1) return;
  const float* p = x + (b * T + s) * dim + c;
  float4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t t = t0;
  for (; t + U <= t1; t += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load_nt_f4(p + (t + u) * dim);
#pragma unroll
    for (int u = 0; u < U; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
  }
  for (; t < t1; ++t) {
    const float4 v = load_nt_f4(p + t * dim);
    acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
  }
#pragma unroll
  for (int u = 1; u < U; ++u) { acc[0].x += acc[u].x; acc[0].y += acc[u].y; acc[0].z += acc[u].z; acc[0].w += acc[u].w; }
  const float inv = 1.0f / (float)n;
  float* o = out + b * ld_out + c;
  atomicAdd(o, acc[0].x * inv); atomicAdd(o + 1, acc[0].y * inv); atomicAdd(o + 2, acc[0].z * inv); atomicAdd(o + 3, acc[0].w * inv);
}

// ---------------------------------------------------------------------------------
// row softmax: S f32 [R, T] (ld_s) -> P bf16 [R, T_pad] (ld_p), zero padded; one wave per row
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ S, int64_t R, int64_t T, int64_t ld_s,
                                                      unsigned short* __restrict__ P, int64_t T_pad, int64_t ld_p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* s = S + row * ld_s;
  float m = -INFINITY;
  for (int64_t i = lane; i < T; i += 64) m = fmaxf(m, s[i]);
  m = wave_max(m);
  float sum = 0.f;
  for (int64_t i = lane; i < T; i += 64) sum += __expf(s[i] - m);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  unsigned short* p = P + row * ld_p;
  for (int64_t i = lane; i < T_pad; i += 64) p[i] = (i < T) ? f32_to_bf16(__expf(s[i] - m) * inv) : (unsigned short)0;
}

// the same softmax with the row in registers (T = 256 NV, no padding): S is read once with 16-byte loads, exp is evaluated once per
// score (the walk above evaluates it twice and moves 4 / 2 bytes per lane and instruction), P leaves as 8-byte stores
template <int NV>
__global__ __launch_bounds__(256) void softmax_reg_kernel(const float* __restrict__ S, int64_t R, int64_t ld_s, unsigned short* __restrict__ P,
                                                          int64_t ld_p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* s = S + row * ld_s;
  float4 v[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) v[j] = load_nt_f4(s + (lane + 64 * j) * 4);
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < NV; ++j) m = fmaxf(m, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
  m = wave_max(m);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j].x = __expf(v[j].x - m); v[j].y = __expf(v[j].y - m); v[j].z = __expf(v[j].z - m); v[j].w = __expf(v[j].w - m);
    sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  unsigned short* p = P + row * ld_p;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    u16x4_t o;
    o[0] = f32_to_bf16(v[j].x * inv); o[1] = f32_to_bf16(v[j].y * inv); o[2] = f32_to_bf16(v[j].z * inv); o[3] = f32_to_bf16(v[j].w * inv);
    *(u16x4_t*)(p + (lane + 64 * j) * 4) = o;
  }
}

// ---------------------------------------------------------------------------------
// adaptive average pool along the last axis: x f32 [rows, T_in] -> y [rows, T_out]
// ---------------------------------------------------------------------------------
__global__ void adaptive_pool_kernel(const float* __restrict__ x, int64_t rows, int64_t T_in, float* __restrict__ y,
                                     int64_t T_out) {
  const int64_t total = rows * T_out;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / T_out, i = idx - r * T_out;
    const int64_t a = (i * T_in) / T_out;
    const int64_t b = ((i + 1) * T_in + T_out - 1) / T_out;
    const float* p = x + r * T_in;
    float acc = 0.f;
    for (int64_t t = a; t < b; ++t) acc += p[t];
    y[idx] = acc / (float)(b - a);
  }
}

// x[m][col0 + n] (=|+=) rowadd[m % T][col0 + n] + gadd[idx[m / T]][col0 + n]   (zero-projector block)
__global__ void fill_embed_kernel(float* __restrict__ x, int64_t BT, int64_t T, int64_t N_out, int64_t hidden, int64_t col0,
                                  const float* __restrict__ pos, const float* __restrict__ subj,
                                  const int64_t* __restrict__ sid) {
  const int64_t total = BT * N_out;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / N_out, n = col0 + (idx - m * N_out);
    float v = 0.f;
    if (pos) v += pos[(m % T) * hidden + n];
    if (subj) v += subj[sid[m / T] * hidden + n];
    x[m * hidden + n] = v;
  }
}

inline unsigned grid_for(int64_t total, int block) {
  int64_t b = (total + block - 1) / block;
  if (b > 256 * 8) b = 256 * 8;  // cap and grid-stride (guide: memory-bound grid sizing)
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int tribe_pack_weight_bf16(const float* src, int64_t rows, int64_t cols, int64_t ld_src, uint16_t* dst,
                                      int64_t rows_pad, int64_t cols_pad, void* stream) {
  TRIBE_REQUIRE(src && dst, "tribe_pack_weight_bf16: null pointer");
  TRIBE_REQUIRE(rows > 0 && cols > 0 && rows_pad >= rows && cols_pad >= cols && ld_src >= cols,
                "tribe_pack_weight_bf16: bad shape rows=%lld cols=%lld rows_pad=%lld cols_pad=%lld ld=%lld", (long long)rows,
                (long long)cols, (long long)rows_pad, (long long)cols_pad, (long long)ld_src);
  TRIBE_REQUIRE(cols_pad % 8 == 0 && ((uintptr_t)dst % 16) == 0, "tribe_pack_weight_bf16: cols_pad %% 8 and 16-byte dst required");
  if (rows == rows_pad && cols == cols_pad && ld_src == cols && (rows * cols) % 32 == 0 && ((uintptr_t)src % 16) == 0) {
    const int64_t n32 = rows * cols / 32;
    hipLaunchKernelGGL(cast_flat_kernel, dim3(grid_for((n32 * 32 + 8191) / 8192 * 256, 256)), dim3(256), 0, (hipStream_t)stream, src, n32, dst);
    TRIBE_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(rows_pad * (cols_pad / 8), 256)), dim3(256), 0, (hipStream_t)stream, src,
                     rows, cols, ld_src, dst, rows_pad, cols_pad);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_pack_subject_weights(const float* w, int64_t S, int64_t C, int64_t V, uint16_t* dst, int64_t V_pad,
                                          int64_t C_pad, void* stream) {
  TRIBE_REQUIRE(w && dst, "tribe_pack_subject_weights: null pointer");
  TRIBE_REQUIRE(S > 0 && C > 0 && V > 0 && V_pad >= V && C_pad >= C && C_pad % 8 == 0,
                "tribe_pack_subject_weights: bad shape S=%lld C=%lld V=%lld V_pad=%lld C_pad=%lld", (long long)S, (long long)C,
                (long long)V, (long long)V_pad, (long long)C_pad);
  // in[s][c][v] -> out[s][v][c]:  I = C (i index), J = V (contiguous in the source)
  dim3 grid((unsigned)((C_pad + 63) / 64), (unsigned)((V_pad + 63) / 64), (unsigned)S);
  hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, w, C * V, (int64_t)0, V, (int64_t)1,
                     1.0f, C, V, dst, V_pad * C_pad, C_pad, C_pad, V_pad);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_pack_features(const void* feat, int32_t dtype, int64_t B, int64_t L, int64_t D, int64_t T,
                                   int32_t layer_mean, uint16_t* dst, int64_t K_pad, void* stream) {
  TRIBE_REQUIRE(feat && dst, "tribe_pack_features: null pointer");
  TRIBE_REQUIRE(B > 0 && L > 0 && D > 0 && T > 0, "tribe_pack_features: bad shape B=%lld L=%lld D=%lld T=%lld", (long long)B,
                (long long)L, (long long)D, (long long)T);
  const int64_t K = layer_mean ? D : L * D;
  TRIBE_REQUIRE(K_pad >= K && K_pad % 8 == 0, "tribe_pack_features: K_pad=%lld must be >= %lld and a multiple of 8",
                (long long)K_pad, (long long)K);
  TRIBE_REQUIRE(B < 65536, "tribe_pack_features: batch too large for one launch");
  if (!layer_mean && (dtype == TRIBE_BF16 || dtype == TRIBE_F32))
    // plain "b (l d) t -> b t (l d)" + cast: the vectorised transpose of backward.hip (4 elements per access), 2x this file's
    // element-wise kernel -- which stays for the layer mean and for f64 inputs
    return tribe_transpose_bf16(feat, dtype, B, L * D, T, L * D * T, T, dst, T * K_pad, K_pad, stream);
  // in[b][k][t] (k = l*D + d for "cat"; reduce over l for "mean") -> out[b][t][k]
  const int64_t s_z = L * D * T, s_i = T;
  const int64_t Lr = layer_mean ? L : 1, s_l = layer_mean ? D * T : 0;
  const float l_scale = layer_mean ? 1.0f / (float)L : 1.0f;
  dim3 grid((unsigned)((K_pad + 63) / 64), (unsigned)((T + 63) / 64), (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
#define TRIBE_PF(TYPE)                                                                                                   \
  hipLaunchKernelGGL(transpose_cast_kernel<TYPE>, grid, dim3(256), 0, s, (const TYPE*)feat, s_z, s_l, s_i, Lr, l_scale, K, T, \
                     dst, T * K_pad, K_pad, K_pad, T)
  if (dtype == TRIBE_F32) TRIBE_PF(float);
  else if (dtype == TRIBE_F64) TRIBE_PF(double);
  else if (dtype == TRIBE_BF16) TRIBE_PF(unsigned short);
  else TRIBE_REQUIRE(false, "tribe_pack_features: unsupported dtype %d", dtype);
#undef TRIBE_PF
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_scalenorm_fwd(const float* x, int64_t rows, int64_t dim, const float* g, float gain_scale, float eps,
                                   void* y, int32_t y_dtype, void* stream) {
  TRIBE_REQUIRE(x && g && y, "tribe_scalenorm_fwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && dim > 0 && dim % 4 == 0, "tribe_scalenorm_fwd: rows=%lld dim=%lld (dim %% 4 required)",
                (long long)rows, (long long)dim);
  TRIBE_REQUIRE(y_dtype == TRIBE_F32 || y_dtype == TRIBE_BF16, "tribe_scalenorm_fwd: y_dtype must be f32 or bf16");
  dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0;
#define TRIBE_SN_REG(NV)                                                                                                  \
  do {                                                                                                                    \
    if (y_dtype == TRIBE_BF16) hipLaunchKernelGGL((scalenorm_reg_kernel<1, NV>), grid, dim3(256), 0, s, x, rows, g, gain_scale, eps, y); \
    else hipLaunchKernelGGL((scalenorm_reg_kernel<0, NV>), grid, dim3(256), 0, s, x, rows, g, gain_scale, eps, y);          \
  } while (0)
  if (aligned && dim == 3072) TRIBE_SN_REG(12);        // the TRIBE encoder width
  else if (aligned && dim == 1024) TRIBE_SN_REG(4);
  else if (aligned && dim == 768) TRIBE_SN_REG(3);     // the small test models
  else if (y_dtype == TRIBE_BF16)
    hipLaunchKernelGGL(scalenorm_kernel<1>, grid, dim3(256), 0, s, x, rows, dim, g, gain_scale, eps, y);
  else
    hipLaunchKernelGGL(scalenorm_kernel<0>, grid, dim3(256), 0, s, x, rows, dim, g, gain_scale, eps, y);
#undef TRIBE_SN_REG
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_rotary_fwd(uint16_t* x, int64_t rows, int64_t T, int64_t row_stride, int32_t n_heads, int32_t dim_head,
                                int32_t rot_dim, const float* cos_tab, const float* sin_tab, int32_t interleaved, void* stream) {
  if (rot_dim == 0) return 0;
  TRIBE_REQUIRE(x && cos_tab && sin_tab, "tribe_rotary_fwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && T > 0 && n_heads > 0 && dim_head > 0, "tribe_rotary_fwd: bad shape");
  TRIBE_REQUIRE(rot_dim > 0 && rot_dim <= dim_head && rot_dim % 8 == 0 && dim_head % 8 == 0,
                "tribe_rotary_fwd: rot_dim=%d must be a multiple of 8 and <= dim_head=%d (dim_head %% 8 == 0)", rot_dim, dim_head);
  TRIBE_REQUIRE(row_stride >= (int64_t)n_heads * dim_head && row_stride % 8 == 0, "tribe_rotary_fwd: bad row stride");
  TRIBE_REQUIRE(interleaved >= 0 && interleaved <= 2, "tribe_rotary_fwd: interleaved must be 0, 1 or 2");
  const int64_t total = rows * n_heads * (rot_dim / 8);
  hipLaunchKernelGGL(rotary_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, T, row_stride, n_heads,
                     dim_head, rot_dim, cos_tab, sin_tab, interleaved);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_rmsnorm_fwd(const float* x, int64_t rows, int64_t dim, const float* w, float eps, void* y, int32_t y_dtype,
                                 void* stream) {
  TRIBE_REQUIRE(x && w && y, "tribe_rmsnorm_fwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && dim > 0 && dim % 4 == 0, "tribe_rmsnorm_fwd: rows=%lld dim=%lld (dim %% 4 required)", (long long)rows,
                (long long)dim);
  TRIBE_REQUIRE(y_dtype == TRIBE_F32 || y_dtype == TRIBE_BF16, "tribe_rmsnorm_fwd: y_dtype must be f32 or bf16");
  if (y_dtype == TRIBE_BF16) launch_rowstat_norm<1, 0>(x, rows, dim, w, nullptr, eps, y, 0.f, (hipStream_t)stream);
  else launch_rowstat_norm<0, 0>(x, rows, dim, w, nullptr, eps, y, 0.f, (hipStream_t)stream);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_norm_quantize_fp8_fwd(const float* x, int64_t rows, int64_t dim, const float* w, const float* b, int32_t layernorm,
                                           float eps, float inv_scale, uint8_t* y8, void* stream) {
  TRIBE_REQUIRE(x && w && y8, "tribe_norm_quantize_fp8_fwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && dim > 0 && dim % 16 == 0, "tribe_norm_quantize_fp8_fwd: rows=%lld dim=%lld (dim %% 16 required: the e4m3 GEMM's K)",
                (long long)rows, (long long)dim);
  TRIBE_REQUIRE(inv_scale > 0.f && ((uintptr_t)y8 % 4) == 0 && (layernorm || !b),
                "tribe_norm_quantize_fp8_fwd: inv_scale must be positive, y8 4-byte aligned, and RMSNorm takes no bias");
  if (layernorm) launch_rowstat_norm<2, 1>(x, rows, dim, w, b, eps, (void*)y8, inv_scale, (hipStream_t)stream);
  else launch_rowstat_norm<2, 0>(x, rows, dim, w, nullptr, eps, (void*)y8, inv_scale, (hipStream_t)stream);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_layernorm_fwd(const float* x, int64_t rows, int64_t dim, const float* w, const float* b, float eps, void* y,
                                   int32_t y_dtype, void* stream) {
  TRIBE_REQUIRE(x && w && y, "tribe_layernorm_fwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && dim > 0 && dim % 4 == 0, "tribe_layernorm_fwd: rows=%lld dim=%lld (dim %% 4 required)", (long long)rows,
                (long long)dim);
  TRIBE_REQUIRE(y_dtype == TRIBE_F32 || y_dtype == TRIBE_BF16, "tribe_layernorm_fwd: y_dtype must be f32 or bf16");
  if (y_dtype == TRIBE_BF16) launch_rowstat_norm<1, 1>(x, rows, dim, w, b, eps, y, 0.f, (hipStream_t)stream);
  else launch_rowstat_norm<0, 1>(x, rows, dim, w, b, eps, y, 0.f, (hipStream_t)stream);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_embedding_fwd(const void* table, int32_t table_dtype, const int64_t* ids, int64_t n, int64_t dim, int64_t vocab,
                                   float* x, void* stream) {
  TRIBE_REQUIRE(table && ids && x, "tribe_embedding_fwd: null pointer");
  TRIBE_REQUIRE(n > 0 && dim > 0 && vocab > 0, "tribe_embedding_fwd: bad shape");
  dim3 grid((unsigned)((n + 3) / 4));
  if (table_dtype == TRIBE_F32)
    hipLaunchKernelGGL(embedding_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)table, ids, n, dim, vocab, x);
  else if (table_dtype == TRIBE_BF16)
    hipLaunchKernelGGL(embedding_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)table, ids, n,
                       dim, vocab, x);
  else
    TRIBE_REQUIRE(false, "tribe_embedding_fwd: table dtype must be f32 or bf16");
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_im2col3d_fwd(const float* pixels, int64_t B, int32_t frames, int32_t chans, int32_t height, int32_t width,
                                  int32_t tubelet, int32_t patch, uint16_t* out, int64_t K_pad, void* stream) {
  TRIBE_REQUIRE(pixels && out, "tribe_im2col3d_fwd: null pointer");
  TRIBE_REQUIRE(B > 0 && frames > 0 && chans > 0 && tubelet > 0 && patch > 0 && frames % tubelet == 0 && height % patch == 0 &&
                    width % patch == 0,
                "tribe_im2col3d_fwd: frames / height / width must be multiples of the tubelet / patch size");
  const int64_t K = (int64_t)chans * tubelet * patch * patch;
  TRIBE_REQUIRE(K_pad >= K && K_pad % 8 == 0, "tribe_im2col3d_fwd: K_pad=%lld must be >= %lld and a multiple of 8", (long long)K_pad,
                (long long)K);
  const int64_t tokens = (int64_t)(frames / tubelet) * (height / patch) * (width / patch);
  hipLaunchKernelGGL(im2col3d_kernel, dim3(grid_for(B * tokens * (K_pad / 8), 256)), dim3(256), 0, (hipStream_t)stream, pixels, B, frames,
                     chans, height, width, tubelet, patch, out, K_pad);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_dwconv_ln_swish_fwd(const uint16_t* x, int64_t B, int64_t T, int32_t C, int32_t K, const float* w_kc,
                                         const float* ln_w, const float* ln_b, float eps, uint16_t* y, void* stream) {
  TRIBE_REQUIRE(x && w_kc && ln_w && ln_b && y, "tribe_dwconv_ln_swish_fwd: null pointer");
  TRIBE_REQUIRE(B > 0 && T > 0 && C > 0 && K > 0 && C % 4 == 0 && C <= 4096, "tribe_dwconv_ln_swish_fwd: C=%d must be a multiple of 4, <= 4096", C);
  TRIBE_REQUIRE(B * T < (1ll << 31), "tribe_dwconv_ln_swish_fwd: too many rows");
  dim3 grid((unsigned)(B * T));
  if (C <= 1024 && K <= 31 && ((uintptr_t)w_kc % 16) == 0 && ((uintptr_t)ln_w % 16) == 0 && ((uintptr_t)ln_b % 16) == 0 && ((uintptr_t)x % 8) == 0) {
    constexpr int TT = 8;
    const int64_t tiles = (T + TT - 1) / TT;
    hipLaunchKernelGGL((dwconv_ln_swish_tile_kernel<TT, 31>), dim3((unsigned)(B * tiles)), dim3(256), 0, (hipStream_t)stream, x, T, C, K, w_kc,
                       ln_w, ln_b, eps, y);
  } else if (C <= 1024)
    hipLaunchKernelGGL(dwconv_ln_swish_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, T, C, K, w_kc, ln_w, ln_b, eps, y);
  else
    hipLaunchKernelGGL(dwconv_ln_swish_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, T, C, K, w_kc, ln_w, ln_b, eps, y);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_gather_rows_fwd(const float* x, int64_t B, int64_t T, int64_t dim, const int64_t* idx, int64_t n, float* out,
                                     void* stream) {
  TRIBE_REQUIRE(x && idx && out, "tribe_gather_rows_fwd: null pointer");
  TRIBE_REQUIRE(B > 0 && T > 0 && dim > 0 && dim % 4 == 0 && n > 0 && B < 65536, "tribe_gather_rows_fwd: bad shape");
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, T, dim, idx, n, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_segment_mean_fwd(const float* x, int64_t B, int64_t T, int64_t dim, const int64_t* start, const int64_t* len,
                                      float* out, int64_t ld_out, void* stream) {
  TRIBE_REQUIRE(x && out, "tribe_segment_mean_fwd: null pointer");
  TRIBE_REQUIRE(B > 0 && T > 0 && dim > 0 && ld_out >= dim && B < 65536, "tribe_segment_mean_fwd: bad shape");
  int64_t slices = (T + 63) / 64;
  if (slices > 128) slices = 128;
  hipError_t e = hipMemset2DAsync(out, (size_t)ld_out * sizeof(float), 0, (size_t)dim * sizeof(float), (size_t)B, (hipStream_t)stream);
  if (e != hipSuccess) { tribe_set_error("tribe_segment_mean_fwd: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  if (dim % 4 == 0 && ((uintptr_t)x % 16) == 0) {
    // >= 64 rows per lane where the segment allows it (atomics on the dim sums serialise per address), enough slices to fill the chip
    const int64_t col_blocks = (dim / 4 + 255) / 256;
    int64_t slices4 = (2048 + col_blocks * B - 1) / (col_blocks * B);
    if (slices4 > (T + 63) / 64) slices4 = (T + 63) / 64;
    if (slices4 < 1) slices4 = 1;
    dim3 grid4((unsigned)col_blocks, (unsigned)B, (unsigned)slices4);
    hipLaunchKernelGGL(segment_mean4_kernel, grid4, dim3(256), 0, (hipStream_t)stream, x, T, dim, start, len, out, ld_out);
    TRIBE_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((unsigned)((dim + 255) / 256), (unsigned)B, (unsigned)slices);
  hipLaunchKernelGGL(segment_mean_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, T, dim, start, len, out, ld_out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_adaptive_avg_pool_fwd(const float* x, int64_t rows, int64_t T_in, float* y, int64_t T_out, void* stream) {
  TRIBE_REQUIRE(x && y, "tribe_adaptive_avg_pool_fwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && T_in > 0 && T_out > 0, "tribe_adaptive_avg_pool_fwd: bad shape rows=%lld T_in=%lld T_out=%lld",
                (long long)rows, (long long)T_in, (long long)T_out);
  hipLaunchKernelGGL(adaptive_pool_kernel, dim3(grid_for(rows * T_out, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, T_in,
                     y, T_out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_projector_zero_fwd(int64_t BT, int64_t T, int64_t N_out, float* x, int64_t hidden, int64_t col0,
                                        const float* pos_embed, const float* subj_embed, const int64_t* subject_id,
                                        void* stream) {
  TRIBE_REQUIRE(x, "tribe_projector_zero_fwd: null pointer");
  TRIBE_REQUIRE(BT > 0 && T > 0 && BT % T == 0 && N_out > 0 && col0 >= 0 && col0 + N_out <= hidden,
                "tribe_projector_zero_fwd: bad shape");
  TRIBE_REQUIRE(!subj_embed || subject_id, "tribe_projector_zero_fwd: subject_embed without subject_id");
  hipLaunchKernelGGL(fill_embed_kernel, dim3(grid_for(BT * N_out, 256)), dim3(256), 0, (hipStream_t)stream, x, BT, T, N_out,
                     hidden, col0, pos_embed, subj_embed, subject_id);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

// ---- internal helpers used by encoder.hip (same shared object, not part of the public ABI) ----
int tribe_internal_softmax(const float* S, int64_t R, int64_t T, int64_t ld_s, uint16_t* P, int64_t T_pad, int64_t ld_p,
                           hipStream_t stream) {
  const dim3 grid((unsigned)((R + 3) / 4));
  const bool reg = T_pad == T && T % 256 == 0 && T <= 2048 && ld_s % 4 == 0 && ld_p % 4 == 0 && ((uintptr_t)S % 16) == 0 && ((uintptr_t)P % 8) == 0;
  if (reg && T == 256) hipLaunchKernelGGL(softmax_reg_kernel<1>, grid, dim3(256), 0, stream, S, R, ld_s, P, ld_p);
  else if (reg && T == 512) hipLaunchKernelGGL(softmax_reg_kernel<2>, grid, dim3(256), 0, stream, S, R, ld_s, P, ld_p);
  else if (reg && T == 1024) hipLaunchKernelGGL(softmax_reg_kernel<4>, grid, dim3(256), 0, stream, S, R, ld_s, P, ld_p);
  else if (reg && T == 2048) hipLaunchKernelGGL(softmax_reg_kernel<8>, grid, dim3(256), 0, stream, S, R, ld_s, P, ld_p);
  else hipLaunchKernelGGL(softmax_kernel, grid, dim3(256), 0, stream, S, R, T, ld_s, P, T_pad, ld_p);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

// V section of qkv [B, T, 3*inner] -> Vt [B*heads, dim_head, T_pad]
int tribe_internal_transpose_v(const uint16_t* qkv, int64_t B, int64_t T, int heads, int dim_head, uint16_t* vt, int64_t T_pad,
                               hipStream_t stream) {
  const int64_t inner = (int64_t)heads * dim_head;
  // per z = (b, h): in[t][d] with row stride 3*inner -> out[d][t];   I = T (i index), J = dim_head (contiguous)
  for (int64_t b = 0; b < B; ++b) {
    dim3 grid((unsigned)((T_pad + 63) / 64), (unsigned)((dim_head + 63) / 64), (unsigned)heads);
    hipLaunchKernelGGL(transpose_cast_kernel<unsigned short>, grid, dim3(256), 0, stream, qkv + b * T * 3 * inner + 2 * inner,
                       (int64_t)dim_head, (int64_t)0, 3 * inner, (int64_t)1, 1.0f, T, (int64_t)dim_head,
                       vt + b * heads * dim_head * T_pad, dim_head * T_pad, T_pad, T_pad, (int64_t)dim_head);
  }
  TRIBE_LAUNCH_CHECK();
  return 0;
}
