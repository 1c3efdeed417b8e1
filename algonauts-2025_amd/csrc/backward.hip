// Backward building blocks of the TRIBE path (pl_module.training_step -> loss.backward()): streaming kernels only;
// the dense gradients are MFMA GEMMs (gemm.hip) fed with transposed bf16 operands produced here.  Roofline: HBM.
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<unsigned short>(const unsigned short* p) { return bf16_to_f32(*p); }

// out[z][c][r] = in[z][r][c] (bf16), r zero-padded to R_pad; 64x64 tiles through LDS.
// VEC = 4: every thread moves 4 consecutive elements per access (8-byte bf16 / 16-byte f32 loads along c, 8-byte
// stores along r) -- the element-wise version (VEC = 1, kept for unaligned views) ran at a third of the bandwidth and
// was 12 % of the training step (the wgrad operands dY^T, X^T and the attention-backward operands all pass through here).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int64_t R, int64_t C, int64_t s_z, int64_t Z0, int64_t s_z0,
                                                        int64_t s_r, unsigned short* __restrict__ out, int64_t so_z, int64_t R_pad) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][68];   // 136-byte rows: 8-byte aligned vector writes
  const int64_t z = blockIdx.z;
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
  const T* src = in + (z / Z0) * s_z + (z % Z0) * s_z0;   // two batch levels: z = z1 * Z0 + z0
  unsigned short* dst = out + z * so_z;
  if (VEC == 4) {
    const int q = threadIdx.x & 15, p = threadIdx.x >> 4;   // 16 groups of 4 elements x 16 lines per pass
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = p + 16 * it;
      const int64_t r = r0 + i, c = c0 + 4 * q;
      u16x4_t v = {0, 0, 0, 0};
      if (r < R) {
        if (c + 4 <= C) {
          if (sizeof(T) == 2) {
            v = *(const u16x4_t*)((const unsigned short*)src + r * s_r + c);
          } else {
            const float4 f = *(const float4*)((const float*)src + r * s_r + c);
            v[0] = f32_to_bf16(f.x); v[1] = f32_to_bf16(f.y); v[2] = f32_to_bf16(f.z); v[3] = f32_to_bf16(f.w);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (c + k < C) v[k] = f32_to_bf16(ldf<T>(src + r * s_r + c + k));
        }
      }
      *(u16x4_t*)&tile[i][4 * q] = v;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = p + 16 * it;                 // column of the input tile = row of the output
      const int64_t c = c0 + i, r = r0 + 4 * q;
      if (c < C && r < R_pad) {
        u16x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = tile[4 * q + k][i];
        *(u16x4_t*)(dst + c * R_pad + r) = o;    // R_pad is a multiple of 4 on this path, so the group never straddles it
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? f32_to_bf16(ldf<T>(src + r * s_r + c)) : (unsigned short)0;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (c < C && r < R_pad) dst[c * R_pad + r] = tile[tx][i];
  }
}

// host side: the vector path needs 4-element alignment of every row start on both sides
inline bool transpose_vec_ok(const void* in, int esz, int64_t s_z, int64_t s_z0, int64_t s_r, const void* out, int64_t so_z, int64_t R_pad) {
  const int64_t a = 4 * esz;
  return ((uintptr_t)in % a) == 0 && (s_z * esz) % a == 0 && (s_z0 * esz) % a == 0 && (s_r * esz) % a == 0 && ((uintptr_t)out % 8) == 0 &&
         so_z % 4 == 0 && R_pad % 4 == 0;
}

// column sums: grid (N/256, slices); atomics into out
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ a, const float* __restrict__ b, int64_t M, int64_t N,
                                                     int64_t ld, float* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int64_t per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t m0 = (int64_t)blockIdx.y * per, m1 = (m0 + per < M) ? m0 + per : M;
  float acc = 0.f;
  for (int64_t m = m0; m < m1; ++m) {
    const float v = ldf<T>(a + m * ld + n);
    acc += b ? v * b[m * ld + n] : v;
  }
  if (m0 < m1) atomicAdd(out + n, acc);
}

// the same sums, four columns per lane (16-byte f32 / 8-byte bf16 loads, read once: nontemporal) and eight rows in flight per lane: the
// one-column kernel above kept one 4-byte load per lane in flight and ran at 2.0-2.4 TB/s (bias gradients: 3.9 ms of the B = 16 step)
template <typename T>
__device__ __forceinline__ float4 ld4f(const T* p);
template <>
__device__ __forceinline__ float4 ld4f<float>(const float* p) { return load_nt_f4(p); }
template <>
__device__ __forceinline__ float4 ld4f<unsigned short>(const unsigned short* p) {
  const u16x4_t v = __builtin_nontemporal_load((const u16x4_t*)p);
  return make_float4(bf16_to_f32(v[0]), bf16_to_f32(v[1]), bf16_to_f32(v[2]), bf16_to_f32(v[3]));
}
template <typename T>
__global__ __launch_bounds__(256) void colsum4_kernel(const T* __restrict__ a, const float* __restrict__ b, int64_t M, int64_t N, int64_t ld,
                                                      float* __restrict__ out) {
  constexpr int U = 8;
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (n >= N) return;
  const int64_t per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t m0 = (int64_t)blockIdx.y * per, m1 = (m0 + per < M) ? m0 + per : M;
  float4 acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t m = m0;
  for (; m + U <= m1; m += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4f<T>(a + (m + u) * ld + n);
    if (b) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4 w = load_nt_f4(b + (m + u) * ld + n);
        v[u].x *= w.x; v[u].y *= w.y; v[u].z *= w.z; v[u].w *= w.w;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
  }
  for (; m < m1; ++m) {
    float4 v = ld4f<T>(a + m * ld + n);
    if (b) { const float4 w = load_nt_f4(b + m * ld + n); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
    acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
  }
#pragma unroll
  for (int u = 1; u < U; ++u) { acc[0].x += acc[u].x; acc[0].y += acc[u].y; acc[0].z += acc[u].z; acc[0].w += acc[u].w; }
  if (m0 < m1) {
    atomicAdd(out + n, acc[0].x); atomicAdd(out + n + 1, acc[0].y); atomicAdd(out + n + 2, acc[0].z); atomicAdd(out + n + 3, acc[0].w);
  }
}

// One pass over an incoming f32 gradient dy [M, N] for everything a Linear / FeedForward backward needs from it: the bias gradient
// sum_m dy, the res_scale gradient sum_m dy * res, and the bf16 copy the dgrad / wgrad GEMMs read (three separate passes before).
__global__ __launch_bounds__(256) void colsum_cast_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t M, int64_t N, int64_t ld,
                                                          float* __restrict__ sum_a, float* __restrict__ sum_ab, unsigned short* __restrict__ a_bf16,
                                                          int64_t ld_bf16) {
  constexpr int U = 8;
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (n >= N) return;
  const int64_t per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t m0 = (int64_t)blockIdx.y * per, m1 = (m0 + per < M) ? m0 + per : M;
  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
  auto take = [&](int64_t m, const float4 v, const float4 w) {
    sa.x += v.x; sa.y += v.y; sa.z += v.z; sa.w += v.w;
    if (b) { sb.x += v.x * w.x; sb.y += v.y * w.y; sb.z += v.z * w.z; sb.w += v.w * w.w; }
    if (a_bf16) {
      u16x4_t o;
      o[0] = f32_to_bf16(v.x); o[1] = f32_to_bf16(v.y); o[2] = f32_to_bf16(v.z); o[3] = f32_to_bf16(v.w);
      *(u16x4_t*)(a_bf16 + m * ld_bf16 + n) = o;
    }
  };
  int64_t m = m0;
  for (; m + U <= m1; m += U) {
    float4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load_nt_f4(a + (m + u) * ld + n);
    if (b) {
#pragma unroll
      for (int u = 0; u < U; ++u) w[u] = load_nt_f4(b + (m + u) * ld + n);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) take(m + u, v[u], b ? w[u] : v[u]);
  }
  for (; m < m1; ++m) {
    const float4 v = load_nt_f4(a + m * ld + n);
    take(m, v, b ? load_nt_f4(b + m * ld + n) : v);
  }
  if (m0 < m1) {
    if (sum_a) { atomicAdd(sum_a + n, sa.x); atomicAdd(sum_a + n + 1, sa.y); atomicAdd(sum_a + n + 2, sa.z); atomicAdd(sum_a + n + 3, sa.w); }
    if (sum_ab) { atomicAdd(sum_ab + n, sb.x); atomicAdd(sum_ab + n + 1, sb.y); atomicAdd(sum_ab + n + 2, sb.z); atomicAdd(sum_ab + n + 3, sb.w); }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void scalenorm_bwd_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                            const float* __restrict__ g, float gain_scale, float eps, int64_t rows,
                                                            int64_t dim, const float* __restrict__ dres, const float* __restrict__ rs,
                                                            float* __restrict__ dx, float* __restrict__ dg) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * dim;
  const T* dyr = dy + row * dim;
  float ss = 0.f, dot = 0.f;
  for (int64_t i = lane; i < dim; i += 64) {
    const float xv = xr[i];
    ss += xv * xv;
    dot += xv * ldf<T>(dyr + i);
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float norm = sqrtf(ss);
  const float s = g[0] * gain_scale;
  // y = x * s / max(norm, eps): below eps the norm is a constant and the projection term vanishes
  const bool clamped = norm < eps;
  const float inv = 1.0f / fmaxf(norm, eps);
  const float proj = clamped ? 0.f : dot * inv * inv;  // <xhat, dy> / norm
  for (int64_t i = lane; i < dim; i += 64) {
    float v = s * inv * (ldf<T>(dyr + i) - xr[i] * proj);
    if (dres) v += dres[row * dim + i] * (rs ? rs[i] : 1.0f);
    dx[row * dim + i] = v;
  }
  if (dg && lane == 0) atomicAdd(dg, gain_scale * dot * inv);
}

// Same with the row of x and dy held in registers (dim = 256 * NV): one pass over HBM, vector accesses, and ONE atomic
// per workgroup for the gain gradient instead of one per row.
template <typename T, int NV>
__global__ __launch_bounds__(256) void scalenorm_bwd_reg_kernel(const float* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ g,
                                                                float gain_scale, float eps, int64_t rows, const float* __restrict__ dres,
                                                                const float* __restrict__ rs, float* __restrict__ dx, float* __restrict__ dg) {
  __shared__ float dg_part[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wv;
  constexpr int64_t dim = 256 * NV;
  float contrib = 0.f;
  if (row < rows) {
    const float4* xr = (const float4*)(x + row * dim);
    float4 xv[NV], dv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) xv[k] = xr[lane + 64 * k];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (sizeof(T) == 2) {
        const u16x4_t h = ((const u16x4_t*)((const unsigned short*)dy + row * dim))[lane + 64 * k];
        dv[k] = make_float4(bf16_to_f32(h[0]), bf16_to_f32(h[1]), bf16_to_f32(h[2]), bf16_to_f32(h[3]));
      } else {
        dv[k] = ((const float4*)((const float*)dy + row * dim))[lane + 64 * k];
      }
    }
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      ss += xv[k].x * xv[k].x + xv[k].y * xv[k].y + xv[k].z * xv[k].z + xv[k].w * xv[k].w;
      dot += xv[k].x * dv[k].x + xv[k].y * dv[k].y + xv[k].z * dv[k].z + xv[k].w * dv[k].w;
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float norm = sqrtf(ss);
    const float s = g[0] * gain_scale;
    const bool clamped = norm < eps;
    const float inv = 1.0f / fmaxf(norm, eps);
    const float proj = clamped ? 0.f : dot * inv * inv;
    const float k0 = s * inv;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float4 v = make_float4(k0 * (dv[k].x - xv[k].x * proj), k0 * (dv[k].y - xv[k].y * proj), k0 * (dv[k].z - xv[k].z * proj),
                             k0 * (dv[k].w - xv[k].w * proj));
      if (dres) {
        const float4 r = ((const float4*)(dres + row * dim))[lane + 64 * k];
        if (rs) {
          const float4 q = ((const float4*)rs)[lane + 64 * k];
          v.x += r.x * q.x; v.y += r.y * q.y; v.z += r.z * q.z; v.w += r.w * q.w;
        } else {
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
      }
      ((float4*)(dx + row * dim))[lane + 64 * k] = v;
    }
    contrib = gain_scale * dot * inv;
  }
  if (dg) {
    if (lane == 0) dg_part[wv] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dg, dg_part[0] + dg_part[1] + dg_part[2] + dg_part[3]);
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const unsigned short* __restrict__ P, const float* __restrict__ dP,
                                                          int64_t rows, int64_t T, int64_t T_pad, int64_t ld_p, int64_t ld_dp,
                                                          float scale, unsigned short* __restrict__ dS, int64_t ld_ds) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const unsigned short* p = P + row * ld_p;
  const float* dp = dP + row * ld_dp;
  float delta = 0.f;
  for (int64_t i = lane; i < T; i += 64) delta += bf16_to_f32(p[i]) * dp[i];
  delta = wave_sum(delta);
  unsigned short* ds = dS + row * ld_ds;
  for (int64_t i = lane; i < T_pad; i += 64)
    ds[i] = (i < T) ? f32_to_bf16(bf16_to_f32(p[i]) * (dp[i] - delta) * scale) : (unsigned short)0;
}

// the same product with the row in registers (T = 256 NV, no padding): P and dP are read once, 8- / 16-byte accesses
template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_reg_kernel(const unsigned short* __restrict__ P, const float* __restrict__ dP, int64_t rows,
                                                              int64_t ld_p, int64_t ld_dp, float scale, unsigned short* __restrict__ dS, int64_t ld_ds) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const unsigned short* p = P + row * ld_p;
  const float* dp = dP + row * ld_dp;
  float4 pv[NV], dv[NV];
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const u16x4_t q = __builtin_nontemporal_load((const u16x4_t*)(p + (lane + 64 * j) * 4));
    pv[j] = make_float4(bf16_to_f32(q[0]), bf16_to_f32(q[1]), bf16_to_f32(q[2]), bf16_to_f32(q[3]));
    dv[j] = load_nt_f4(dp + (lane + 64 * j) * 4);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) delta += (pv[j].x * dv[j].x + pv[j].y * dv[j].y) + (pv[j].z * dv[j].z + pv[j].w * dv[j].w);
  delta = wave_sum(delta);
  unsigned short* ds = dS + row * ld_ds;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    u16x4_t o;
    o[0] = f32_to_bf16(pv[j].x * (dv[j].x - delta) * scale); o[1] = f32_to_bf16(pv[j].y * (dv[j].y - delta) * scale);
    o[2] = f32_to_bf16(pv[j].z * (dv[j].z - delta) * scale); o[3] = f32_to_bf16(pv[j].w * (dv[j].w - delta) * scale);
    *(u16x4_t*)(ds + (lane + 64 * j) * 4) = o;
  }
}

// dpred = gs * 2 (p - t) / n.  HBM-bound: 12 bytes per element; float4 streams, two pairs in flight per lane.
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                      const float* __restrict__ gs, float* __restrict__ dp, int vec) {
  const float k = gs[0] * 2.0f / (float)n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = vec ? (n >> 2) : 0;
  const float4* p4 = (const float4*)p;
  const float4* t4 = (const float4*)t;
  float4* d4 = (float4*)dp;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 a0 = load_nt_f4(p4 + i), b0 = load_nt_f4(t4 + i);
    const float4 a1 = load_nt_f4(p4 + i + stride), b1 = load_nt_f4(t4 + i + stride);
    d4[i] = make_float4(k * (a0.x - b0.x), k * (a0.y - b0.y), k * (a0.z - b0.z), k * (a0.w - b0.w));
    d4[i + stride] = make_float4(k * (a1.x - b1.x), k * (a1.y - b1.y), k * (a1.z - b1.z), k * (a1.w - b1.w));
  }
  for (; i < n4; i += stride) {
    const float4 a = p4[i], b = t4[i];
    d4[i] = make_float4(k * (a.x - b.x), k * (a.y - b.y), k * (a.z - b.z), k * (a.w - b.w));
  }
  for (int64_t j = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) dp[j] = k * (p[j] - t[j]);
}

__global__ void pool_bwd_kernel(const float* __restrict__ dy, int64_t rows, int64_t T_in, int64_t T_out, float* __restrict__ dx) {
  const int64_t total = rows * T_in;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / T_in, t = idx - r * T_in;
    // windows i with floor(i T_in / T_out) <= t < ceil((i+1) T_in / T_out)
    int64_t i_lo = (t * T_out) / T_in - 1;  // windows before this one end at or before t
    if (i_lo < 0) i_lo = 0;
    float acc = 0.f;
    for (int64_t i = i_lo; i < T_out; ++i) {
      const int64_t a = (i * T_in) / T_out, b = ((i + 1) * T_in + T_out - 1) / T_out;
      if (a > t) break;
      if (t < b) acc += dy[r * T_out + i] / (float)(b - a);
    }
    dx[idx] = acc;
  }
}

__global__ __launch_bounds__(256) void rowsum_scatter_kernel(const float* __restrict__ x, int64_t B, int64_t V, int64_t T,
                                                             const int64_t* __restrict__ idx, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // row = b * V + v
  const int64_t b = row / V, v = row - b * V;
  if (b >= B) return;
  const float* p = x + row * T;
  float acc = 0.f;
  for (int64_t t = lane; t < T; t += 64) acc += p[t];
  acc = wave_sum(acc);
  if (lane == 0) atomicAdd(out + idx[b] * V + v, acc);
}

// dst[idx[b]][e] += src[b][e] for b = 0 .. B - 1 IN ORDER: one thread owns element e (4 floats) of every destination slab, so the sum
// of the samples of one subject has a fixed order (atomics would not) and nothing races
__global__ __launch_bounds__(256) void slab_scatter_sum_kernel(const float* __restrict__ src, int64_t B, int64_t n4, const int64_t* __restrict__ idx,
                                                               float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n4) return;
  for (int64_t b = 0; b < B; ++b) {
    const float4 v = load_nt_f4(src + (b * n4 + e) * 4);
    float4* d = (float4*)dst + idx[b] * n4 + e;
    float4 acc = *d;
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    *d = acc;
  }
}

__global__ void scale_cols_kernel(const float* __restrict__ x, const float* __restrict__ rs, int64_t M, int64_t N, float* __restrict__ y) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = rs ? x[i] * rs[i % N] : x[i];
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, int64_t n4, unsigned short* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = ((const float4*)x)[i];
    u16x4_t o;
    o[0] = f32_to_bf16(v.x); o[1] = f32_to_bf16(v.y); o[2] = f32_to_bf16(v.z); o[3] = f32_to_bf16(v.w);
    ((u16x4_t*)y)[i] = o;
  }
}

// d(1 - r)/dx_i = -( yc_i / den - cov * sy * xc_i / (sx * den^2) ),  den = sx * sy + 1e-8.  Workgroup (v, chunk): voxel v of a run
// of sequences, one row per wave at a time (HBM-bound, 12 bytes per element; float4 when the rows allow it).
__global__ __launch_bounds__(256) void pearson_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                               int64_t B, int64_t V, int64_t T, int64_t sb, int64_t sv, int64_t st,
                                                               const double* __restrict__ stats, float w, const float* __restrict__ gs,
                                                               float* __restrict__ dpred, int64_t rows_per_wg, int vec) {
  const int64_t v = blockIdx.x;
  const double* s = stats + v * 6;
  const double n = s[5];
  const float mx = (float)(s[0] / n), my = (float)(s[1] / n);
  const double cov = s[4] - s[0] * s[1] / n;
  double vx = s[2] - s[0] * s[0] / n, vy = s[3] - s[1] * s[1] / n;
  vx = vx > 0.0 ? vx : 0.0;
  vy = vy > 0.0 ? vy : 0.0;
  const float sx = sqrtf((float)vx), sy = sqrtf((float)vy);
  const float den = sx * sy + 1e-8f;
  const float k = gs[0] * w;
  const float a = -k / den;                                               // * yc_i
  const float c = (sx > 0.f) ? k * (float)cov * sy / (sx * den * den) : 0.f;  // * xc_i
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b0 = (int64_t)blockIdx.y * rows_per_wg;
  const int64_t b1 = (b0 + rows_per_wg < B) ? b0 + rows_per_wg : B;
  for (int64_t b = b0 + wave; b < b1; b += 4) {
    const float* x = pred + b * sb + v * sv;
    const float* y = truth + b * sb + v * sv;
    float* d = dpred + (b * V + v) * T;
    if (vec) {
      const int64_t T4 = T >> 2;
      for (int64_t i = lane; i < T4; i += 64) {
        const float4 xa = load_nt_f4((const float4*)x + i), ya = load_nt_f4((const float4*)y + i);
        ((float4*)d)[i] = make_float4(a * (ya.x - my) + c * (xa.x - mx), a * (ya.y - my) + c * (xa.y - mx),
                                      a * (ya.z - my) + c * (xa.z - mx), a * (ya.w - my) + c * (xa.w - mx));
      }
    } else {
      for (int64_t t = lane; t < T; t += 64) d[t] = a * (y[t * st] - my) + c * (x[t * st] - mx);
    }
  }
}

__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ S, int64_t N, int64_t ld, float* __restrict__ lse,
                                                       float* __restrict__ diag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* s = S + row * ld;
  float m = -INFINITY;
  for (int64_t i = lane; i < N; i += 64) m = fmaxf(m, s[i]);
  m = wave_max(m);
  float sum = 0.f;
  for (int64_t i = lane; i < N; i += 64) sum += __expf(s[i] - m);
  sum = wave_sum(sum);
  if (lane == 0) { lse[row] = m + logf(sum); if (diag) diag[row] = s[row]; }
}

__global__ __launch_bounds__(256) void infonce_dlogits_kernel(const float* __restrict__ S, int64_t N, int64_t ld,
                                                              const float* __restrict__ lse_r, const float* __restrict__ lse_c,
                                                              const float* __restrict__ gs, unsigned short* __restrict__ dL, int64_t N_pad) {
  const int64_t i = blockIdx.x;
  const float k = gs[0] * 0.5f / (float)N, lr = lse_r[i];
  for (int64_t j = threadIdx.x; j < N_pad; j += blockDim.x) {
    float v = 0.f;
    if (j < N) {
      const float s = S[i * ld + j];
      v = k * (__expf(s - lr) + __expf(s - lse_c[j]) - (i == j ? 2.0f : 0.0f));
    }
    dL[i * N_pad + j] = f32_to_bf16(v);
  }
}

inline unsigned grid_for(int64_t total, int block) {
  int64_t b = (total + block - 1) / block;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (unsigned)b;
}


// ---- Adam over a whole parameter group in ONE launch (torch.optim.Adam semantics, defaults.py:126-133) -----------------
// table[i] = {p, g, m, v, n}; work is cut into chunks of ADAM_CHUNK elements: chunk c belongs to tensor chunk_tensor[c] and
// starts at element chunk_start[c].  HBM-bound: 16 B read + 12 B written per parameter.
constexpr int ADAM_CHUNK = 16384;
__global__ __launch_bounds__(256) void adam_step_kernel(const tribe_adam_tensor* __restrict__ table, const int32_t* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_start, float lr, float beta1, float beta2, float eps,
                                                        float weight_decay, float bias_c1, float bias_c2_sqrt, int decoupled) {
  const tribe_adam_tensor t = table[chunk_tensor[blockIdx.x]];
  const int64_t i0 = chunk_start[blockIdx.x];
  const int64_t i1 = (i0 + ADAM_CHUNK < t.n) ? i0 + ADAM_CHUNK : t.n;
  const float step_size = lr / bias_c1;
  const bool vec = (((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0 && (i0 & 3) == 0;
  auto update = [&](float& p, float g, float& m, float& v) {
    if (weight_decay != 0.f) {
      if (decoupled) p *= 1.f - lr * weight_decay;           // AdamW
      else g = fmaf(weight_decay, p, g);                      // Adam: L2 term joins the gradient
    }
    m = fmaf(1.f - beta1, g - m, m);                          // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(1.f - beta2, g * g, beta2 * v);                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / bias_c2_sqrt + eps;
    p -= step_size * (m / denom);                             // param.addcdiv_(exp_avg, denom, value=-step_size)
  };
  if (vec) {
    const int64_t n4 = (i1 - i0) / 4;
    const bool shadow4 = t.p_bf16 && (((uintptr_t)t.p_bf16) & 7) == 0;
    for (int64_t q = threadIdx.x; q < n4; q += 256) {
      const int64_t i = i0 + 4 * q;
      float4 p = *(float4*)(t.p + i), m = *(float4*)(t.m + i), v = *(float4*)(t.v + i);
      const float4 g = *(const float4*)(t.g + i);
      update(p.x, g.x, m.x, v.x); update(p.y, g.y, m.y, v.y); update(p.z, g.z, m.z, v.z); update(p.w, g.w, m.w, v.w);
      *(float4*)(t.p + i) = p; *(float4*)(t.m + i) = m; *(float4*)(t.v + i) = v;
      if (shadow4) {   // the bf16 operand copy the next forward's GEMMs read: written here instead of by a cast pass over the f32 weights
        u16x4_t o;
        o[0] = f32_to_bf16(p.x); o[1] = f32_to_bf16(p.y); o[2] = f32_to_bf16(p.z); o[3] = f32_to_bf16(p.w);
        *(u16x4_t*)(t.p_bf16 + i) = o;
      } else if (t.p_bf16) {
        t.p_bf16[i] = f32_to_bf16(p.x); t.p_bf16[i + 1] = f32_to_bf16(p.y); t.p_bf16[i + 2] = f32_to_bf16(p.z); t.p_bf16[i + 3] = f32_to_bf16(p.w);
      }
    }
    for (int64_t i = i0 + 4 * n4 + threadIdx.x; i < i1; i += 256) {
      update(t.p[i], t.g[i], t.m[i], t.v[i]);
      if (t.p_bf16) t.p_bf16[i] = f32_to_bf16(t.p[i]);
    }
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
      update(t.p[i], t.g[i], t.m[i], t.v[i]);
      if (t.p_bf16) t.p_bf16[i] = f32_to_bf16(t.p[i]);
    }
  }
}

// ---- running average of the weights (Lightning StochasticWeightAveraging / torch.optim.swa_utils.AveragedModel's default avg_fn:
// avg += (p - avg) / (n_averaged + 1); main.py:365-373).  Same table / work list as Adam: t.p = the average (updated), t.g = the live
// parameter (read); m, v unused.  HBM-bound: 8 B read + 4 B written per parameter.  weight == 1 is a plain copy (first average).
__global__ __launch_bounds__(256) void swa_update_kernel(const tribe_adam_tensor* __restrict__ table, const int32_t* __restrict__ chunk_tensor,
                                                         const int64_t* __restrict__ chunk_start, float weight) {
  const tribe_adam_tensor t = table[chunk_tensor[blockIdx.x]];
  const int64_t i0 = chunk_start[blockIdx.x];
  const int64_t i1 = (i0 + ADAM_CHUNK < t.n) ? i0 + ADAM_CHUNK : t.n;
  const bool copy = weight == 1.f;
  auto blend = [&](float a, float p) { return copy ? p : a + (p - a) * weight; };
  const bool vec = (((uintptr_t)t.p | (uintptr_t)t.g) & 15) == 0 && (i0 & 3) == 0;
  int64_t done = i0;
  if (vec) {
    const int64_t n4 = (i1 - i0) / 4;
    for (int64_t q = threadIdx.x; q < n4; q += 256) {
      const int64_t i = i0 + 4 * q;
      float4 a = *(float4*)(t.p + i);
      const float4 p = *(const float4*)(t.g + i);
      a.x = blend(a.x, p.x); a.y = blend(a.y, p.y); a.z = blend(a.z, p.z); a.w = blend(a.w, p.w);
      *(float4*)(t.p + i) = a;
    }
    done = i0 + 4 * n4;
  }
  for (int64_t i = done + threadIdx.x; i < i1; i += 256) t.p[i] = blend(t.p[i], t.g[i]);
}

}  // namespace

int tribe_internal_softmax(const float* S, int64_t R, int64_t T, int64_t ld_s, uint16_t* P, int64_t T_pad, int64_t ld_p, hipStream_t stream);

extern "C" int tribe_transpose_bf16(const void* in, int32_t in_dtype, int64_t Z, int64_t R, int64_t C, int64_t s_z, int64_t s_r,
                                    uint16_t* out, int64_t so_z, int64_t R_pad, void* stream) {
  TRIBE_REQUIRE(in && out, "tribe_transpose_bf16: null pointer");
  TRIBE_REQUIRE(Z > 0 && R > 0 && C > 0 && R_pad >= R && s_r >= C && Z < 65536, "tribe_transpose_bf16: bad shape");
  dim3 grid((unsigned)((R_pad + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)Z);
  TRIBE_REQUIRE(in_dtype == TRIBE_F32 || in_dtype == TRIBE_BF16, "tribe_transpose_bf16: dtype must be f32 or bf16");
  const bool vec = transpose_vec_ok(in, in_dtype == TRIBE_F32 ? 4 : 2, s_z, 0, s_r, out, so_z, R_pad);
#define TRIBE_TR(TYPE, VEC)                                                                                                          \
  hipLaunchKernelGGL((transpose_kernel<TYPE, VEC>), grid, dim3(256), 0, (hipStream_t)stream, (const TYPE*)in, R, C, s_z, (int64_t)1, (int64_t)0, \
                     s_r, out, so_z, R_pad)
  if (in_dtype == TRIBE_F32) { if (vec) TRIBE_TR(float, 4); else TRIBE_TR(float, 1); }
  else { if (vec) TRIBE_TR(unsigned short, 4); else TRIBE_TR(unsigned short, 1); }
#undef TRIBE_TR
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_colsum_fwd(const void* a, int32_t a_dtype, const float* b, int64_t M, int64_t N, int64_t ld, float* out,
                                int32_t accumulate, void* stream) {
  TRIBE_REQUIRE(a && out, "tribe_colsum_fwd: null pointer");
  TRIBE_REQUIRE(M > 0 && N > 0 && ld >= N, "tribe_colsum_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s);
    if (e != hipSuccess) { tribe_set_error("tribe_colsum_fwd: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  }
  const int esz = a_dtype == TRIBE_F32 ? 4 : 2;
  if ((a_dtype == TRIBE_F32 || a_dtype == TRIBE_BF16) && N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)a % (4 * esz)) == 0 && (!b || ((uintptr_t)b % 16) == 0)) {
    // enough row slices for several waves per SIMD, but at least 32 rows per lane so that the 8-deep loop is what runs
    const int64_t col_blocks = (N / 4 + 255) / 256;
    int64_t slices4 = (8192 + col_blocks - 1) / col_blocks;
    if (slices4 > (M + 127) / 128) slices4 = (M + 127) / 128;   // >= 128 rows per lane: the atomics on the N sums are serialised per address
    if (slices4 < 1) slices4 = 1;
    if (slices4 > 65535) slices4 = 65535;
    dim3 grid4((unsigned)col_blocks, (unsigned)slices4);
    if (a_dtype == TRIBE_F32) hipLaunchKernelGGL(colsum4_kernel<float>, grid4, dim3(256), 0, s, (const float*)a, b, M, N, ld, out);
    else hipLaunchKernelGGL(colsum4_kernel<unsigned short>, grid4, dim3(256), 0, s, (const unsigned short*)a, b, M, N, ld, out);
    TRIBE_LAUNCH_CHECK();
    return 0;
  }
  int64_t slices = (M + 127) / 128;
  if (slices > 256) slices = 256;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)slices);
  if (a_dtype == TRIBE_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)a, b, M, N, ld, out);
  else if (a_dtype == TRIBE_BF16) hipLaunchKernelGGL(colsum_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short*)a, b, M, N, ld, out);
  else TRIBE_REQUIRE(false, "tribe_colsum_fwd: dtype must be f32 or bf16");
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_colsum_cast_fwd(const float* a, const float* b, int64_t M, int64_t N, int64_t ld, float* sum_a, float* sum_ab,
                                     uint16_t* a_bf16, int64_t ld_bf16, void* stream) {
  TRIBE_REQUIRE(a && (sum_a || sum_ab || a_bf16), "tribe_colsum_cast_fwd: nothing to compute");
  TRIBE_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ld >= N && ld % 4 == 0 && ((uintptr_t)a % 16) == 0 && (!b || ((uintptr_t)b % 16) == 0) && (!sum_ab || b),
                "tribe_colsum_cast_fwd: needs N %% 4 == 0, 16-byte aligned rows and b for sum_ab");
  TRIBE_REQUIRE(!a_bf16 || (ld_bf16 >= N && ld_bf16 % 4 == 0 && ((uintptr_t)a_bf16 % 8) == 0), "tribe_colsum_cast_fwd: bad bf16 output");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipSuccess;
  if (sum_a) e = hipMemsetAsync(sum_a, 0, (size_t)N * sizeof(float), s);
  if (e == hipSuccess && sum_ab) e = hipMemsetAsync(sum_ab, 0, (size_t)N * sizeof(float), s);
  if (e != hipSuccess) { tribe_set_error("tribe_colsum_cast_fwd: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  const int64_t col_blocks = (N / 4 + 255) / 256;
  int64_t slices = (8192 + col_blocks - 1) / col_blocks;
  if (slices > (M + 127) / 128) slices = (M + 127) / 128;
  if (slices < 1) slices = 1;
  if (slices > 65535) slices = 65535;
  hipLaunchKernelGGL(colsum_cast_kernel, dim3((unsigned)col_blocks, (unsigned)slices), dim3(256), 0, s, a, b, M, N, ld, sum_a, sum_ab, a_bf16, ld_bf16);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_scalenorm_bwd(const float* x, const void* dy, int32_t dy_dtype, const float* g, float gain_scale, float eps,
                                   int64_t rows, int64_t dim, const float* dres, const float* rs, float* dx, float* dg, void* stream) {
  TRIBE_REQUIRE(x && dy && g && dx, "tribe_scalenorm_bwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && dim > 0, "tribe_scalenorm_bwd: bad shape");
  dim3 grid((unsigned)((rows + 3) / 4));
  TRIBE_REQUIRE(dy_dtype == TRIBE_F32 || dy_dtype == TRIBE_BF16, "tribe_scalenorm_bwd: dy dtype must be f32 or bf16");
  hipStream_t s = (hipStream_t)stream;
  const bool aligned = (((uintptr_t)x | (uintptr_t)dx | (uintptr_t)(dres ? dres : x) | (uintptr_t)(rs ? rs : x)) % 16) == 0 && ((uintptr_t)dy % 16) == 0;
#define TRIBE_SNB(NV)                                                                                                             \
  do {                                                                                                                            \
    if (dy_dtype == TRIBE_F32)                                                                                                    \
      hipLaunchKernelGGL((scalenorm_bwd_reg_kernel<float, NV>), grid, dim3(256), 0, s, x, (const float*)dy, g, gain_scale, eps, rows, dres, rs, dx, dg); \
    else                                                                                                                          \
      hipLaunchKernelGGL((scalenorm_bwd_reg_kernel<unsigned short, NV>), grid, dim3(256), 0, s, x, (const unsigned short*)dy, g, gain_scale, eps, \
                         rows, dres, rs, dx, dg);                                                                                 \
  } while (0)
  if (aligned && dim == 3072) TRIBE_SNB(12);
  else if (aligned && dim == 768) TRIBE_SNB(3);
  else if (dy_dtype == TRIBE_F32)
    hipLaunchKernelGGL(scalenorm_bwd_kernel<float>, grid, dim3(256), 0, s, x, (const float*)dy, g, gain_scale, eps, rows, dim, dres, rs, dx, dg);
  else
    hipLaunchKernelGGL(scalenorm_bwd_kernel<unsigned short>, grid, dim3(256), 0, s, x, (const unsigned short*)dy, g, gain_scale, eps, rows,
                       dim, dres, rs, dx, dg);
#undef TRIBE_SNB
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_softmax_bwd(const uint16_t* P, const float* dP, int64_t rows, int64_t T, int64_t T_pad, int64_t ld_p, int64_t ld_dp,
                                 float scale, uint16_t* dS, int64_t ld_ds, void* stream) {
  TRIBE_REQUIRE(P && dP && dS, "tribe_softmax_bwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && T > 0 && T_pad >= T && ld_p >= T && ld_dp >= T && ld_ds >= T_pad, "tribe_softmax_bwd: bad shape");
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  const bool reg = T_pad == T && T % 256 == 0 && T <= 2048 && ld_p % 4 == 0 && ld_dp % 4 == 0 && ld_ds % 4 == 0 && ((uintptr_t)P % 8) == 0 &&
                   ((uintptr_t)dP % 16) == 0 && ((uintptr_t)dS % 8) == 0;
  if (reg && T == 256) hipLaunchKernelGGL(softmax_bwd_reg_kernel<1>, grid, dim3(256), 0, st, P, dP, rows, ld_p, ld_dp, scale, dS, ld_ds);
  else if (reg && T == 512) hipLaunchKernelGGL(softmax_bwd_reg_kernel<2>, grid, dim3(256), 0, st, P, dP, rows, ld_p, ld_dp, scale, dS, ld_ds);
  else if (reg && T == 1024) hipLaunchKernelGGL(softmax_bwd_reg_kernel<4>, grid, dim3(256), 0, st, P, dP, rows, ld_p, ld_dp, scale, dS, ld_ds);
  else if (reg && T == 2048) hipLaunchKernelGGL(softmax_bwd_reg_kernel<8>, grid, dim3(256), 0, st, P, dP, rows, ld_p, ld_dp, scale, dS, ld_ds);
  else hipLaunchKernelGGL(softmax_bwd_kernel, grid, dim3(256), 0, st, P, dP, rows, T, T_pad, ld_p, ld_dp, scale, dS, ld_ds);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

// D[b][h][t] = scale * sum_d a[b T + t][h dh + d] * b[b T + t][h dh + d]: one wave per (row, head), 8 bf16 (16 bytes) per lane and operand.
// The rowsum(dO * O) of the attention backward (= rowsum(P * dP), FlashAttention-2 eq. for D), written as the row bias the dS GEMM adds.
__global__ __launch_bounds__(256) void rowdot_heads_kernel(const unsigned short* __restrict__ a, int64_t ld_a, const unsigned short* __restrict__ b,
                                                           int64_t ld_b, int64_t rows, int T, int heads, int dim_head, float scale,
                                                           float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (row, head), head fastest
  if (item >= rows * heads) return;
  const int64_t row = item / heads;
  const int hd = (int)(item - row * heads);
  const unsigned short* pa = a + row * ld_a + (int64_t)hd * dim_head;
  const unsigned short* pb = b + row * ld_b + (int64_t)hd * dim_head;
  float acc = 0.f;
  for (int d = lane * 8; d < dim_head; d += 512) {
    const uint4 va = *(const uint4*)(pa + d), vb = *(const uint4*)(pb + d);
    const unsigned int wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc = fmaf(__uint_as_float(wa[k] << 16), __uint_as_float(wb[k] << 16), acc);
      acc = fmaf(__uint_as_float(wa[k] & 0xffff0000u), __uint_as_float(wb[k] & 0xffff0000u), acc);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) out[((row / T) * heads + hd) * (int64_t)T + row % T] = acc * scale;
}

extern "C" int tribe_rowdot_heads_bf16(const uint16_t* a, int64_t ld_a, const uint16_t* b, int64_t ld_b, int64_t B, int64_t T, int32_t heads,
                                       int32_t dim_head, float scale, float* out, void* stream) {
  TRIBE_REQUIRE(a && b && out, "tribe_rowdot_heads_bf16: null pointer");
  TRIBE_REQUIRE(B > 0 && T > 0 && T < (1ll << 31) && heads > 0 && dim_head > 0 && dim_head % 8 == 0, "tribe_rowdot_heads_bf16: bad shape");
  TRIBE_REQUIRE(ld_a >= (int64_t)heads * dim_head && ld_b >= (int64_t)heads * dim_head && ld_a % 8 == 0 && ld_b % 8 == 0 &&
                    ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0,
                "tribe_rowdot_heads_bf16: rows must be 16-byte aligned and at least heads * dim_head wide");
  const int64_t items = B * T * heads;
  TRIBE_REQUIRE((items + 3) / 4 < (1ll << 31), "tribe_rowdot_heads_bf16: grid too large");
  hipLaunchKernelGGL(rowdot_heads_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, ld_a, b, ld_b, B * T, (int)T,
                     (int)heads, (int)dim_head, scale, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_softmax_fwd(const float* S, int64_t rows, int64_t T, int64_t ld_s, uint16_t* P, int64_t T_pad, int64_t ld_p, void* stream) {
  TRIBE_REQUIRE(S && P, "tribe_softmax_fwd: null pointer");
  TRIBE_REQUIRE(rows > 0 && T > 0 && T_pad >= T && ld_s >= T && ld_p >= T_pad, "tribe_softmax_fwd: bad shape");
  return tribe_internal_softmax(S, rows, T, ld_s, P, T_pad, ld_p, (hipStream_t)stream);
}

extern "C" int tribe_mse_bwd(const float* pred, const float* truth, int64_t n, const float* gscale, float* dpred, void* stream) {
  TRIBE_REQUIRE(pred && truth && gscale && dpred && n > 0, "tribe_mse_bwd: bad argument");
  const int vec = ((uintptr_t)pred % 16) == 0 && ((uintptr_t)truth % 16) == 0 && ((uintptr_t)dpred % 16) == 0;
  int64_t nb = (n / 4 + 511) / 512;
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, pred, truth, n, gscale, dpred, vec);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_adaptive_avg_pool_bwd(const float* dy, int64_t rows, int64_t T_in, int64_t T_out, float* dx, void* stream) {
  TRIBE_REQUIRE(dy && dx && rows > 0 && T_in > 0 && T_out > 0, "tribe_adaptive_avg_pool_bwd: bad argument");
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(grid_for(rows * T_in, 256)), dim3(256), 0, (hipStream_t)stream, dy, rows, T_in, T_out, dx);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_rowsum_scatter(const float* x, int64_t B, int64_t V, int64_t T, const int64_t* idx, float* out, void* stream) {
  TRIBE_REQUIRE(x && idx && out && B > 0 && V > 0 && T > 0 && B < 65536, "tribe_rowsum_scatter: bad argument");
  hipLaunchKernelGGL(rowsum_scatter_kernel, dim3((unsigned)((B * V + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, B, V, T, idx, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_slab_scatter_sum(const float* src, int64_t B, int64_t n, const int64_t* idx, float* dst, void* stream) {
  TRIBE_REQUIRE(src && idx && dst && B > 0 && n > 0 && n % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                "tribe_slab_scatter_sum: needs n %% 4 == 0 and 16-byte aligned slabs");
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(slab_scatter_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, B, n4, idx, dst);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_pearson_loss_bwd(const float* pred, const float* truth, int64_t B, int64_t V, int64_t T, int64_t sb, int64_t sv,
                                      int64_t st, const double* stats, int32_t reduction_sum, const float* gscale, float* dpred,
                                      void* stream) {
  TRIBE_REQUIRE(pred && truth && stats && gscale && dpred && B > 0 && V > 0 && T > 0, "tribe_pearson_loss_bwd: bad argument");
  const int vec = st == 1 && T % 4 == 0 && sb % 4 == 0 && sv % 4 == 0 && ((uintptr_t)pred % 16) == 0 && ((uintptr_t)truth % 16) == 0 &&
                  ((uintptr_t)dpred % 16) == 0;
  int64_t chunks = (4096 + V - 1) / V;
  if (chunks > (B + 3) / 4) chunks = (B + 3) / 4;
  if (chunks < 1) chunks = 1;
  if (chunks > 65535) chunks = 65535;
  const int64_t rows_per_wg = (B + chunks - 1) / chunks;
  chunks = (B + rows_per_wg - 1) / rows_per_wg;
  hipLaunchKernelGGL(pearson_loss_bwd_kernel, dim3((unsigned)V, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, pred, truth, B, V, T, sb, sv,
                     st, stats, reduction_sum ? 1.0f : 1.0f / (float)V, gscale, dpred, rows_per_wg, vec);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_lse_rows_fwd(const float* S, int64_t N, int64_t ld, float* lse, float* diag, void* stream) {
  TRIBE_REQUIRE(S && lse && N > 0 && ld >= N, "tribe_lse_rows_fwd: bad argument");
  hipLaunchKernelGGL(lse_rows_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, N, ld, lse, diag);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_infonce_dlogits(const float* S, int64_t N, int64_t ld, const float* lse_r, const float* lse_c, const float* gscale,
                                     uint16_t* dL, int64_t N_pad, void* stream) {
  TRIBE_REQUIRE(S && lse_r && lse_c && gscale && dL && N > 0 && ld >= N && N_pad >= N && N < (1ll << 31), "tribe_infonce_dlogits: bad argument");
  hipLaunchKernelGGL(infonce_dlogits_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, S, N, ld, lse_r, lse_c, gscale, dL, N_pad);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_scale_cols_fwd(const float* x, const float* rs, int64_t M, int64_t N, float* y, void* stream) {
  TRIBE_REQUIRE(x && y && M > 0 && N > 0, "tribe_scale_cols_fwd: bad argument");
  hipLaunchKernelGGL(scale_cols_kernel, dim3(grid_for(M * N, 256)), dim3(256), 0, (hipStream_t)stream, x, rs, M, N, y);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_cast_bf16_fwd(const float* x, int64_t n, uint16_t* y, void* stream) {
  TRIBE_REQUIRE(x && y && n > 0 && n % 4 == 0, "tribe_cast_bf16_fwd: n must be a positive multiple of 4");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, n / 4, y);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t tribe_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int tribe_adam_step(const tribe_adam_tensor* table, const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int64_t step, int32_t decoupled, void* stream) {
  TRIBE_REQUIRE(table && chunk_tensor && chunk_start, "tribe_adam_step: null pointer");
  TRIBE_REQUIRE(n_chunks > 0 && n_chunks < (1ll << 31) && step >= 1, "tribe_adam_step: need chunks and a 1-based step count");
  TRIBE_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "tribe_adam_step: bad hyper-parameters");
  const double c1 = 1.0 - pow((double)beta1, (double)step), c2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_start, lr, beta1,
                     beta2, eps, weight_decay, (float)c1, (float)sqrt(c2), decoupled);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_transpose_bf16_b2(const void* in, int32_t in_dtype, int64_t Z1, int64_t Z0, int64_t R, int64_t C, int64_t s_z1, int64_t s_z0,
                                       int64_t s_r, uint16_t* out, int64_t so_z, int64_t R_pad, void* stream) {
  TRIBE_REQUIRE(in && out, "tribe_transpose_bf16_b2: null pointer");
  TRIBE_REQUIRE(Z1 > 0 && Z0 > 0 && R > 0 && C > 0 && R_pad >= R && s_r >= C && Z1 * Z0 < 65536, "tribe_transpose_bf16_b2: bad shape");
  dim3 grid((unsigned)((R_pad + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)(Z1 * Z0));
  TRIBE_REQUIRE(in_dtype == TRIBE_F32 || in_dtype == TRIBE_BF16, "tribe_transpose_bf16_b2: dtype must be f32 or bf16");
  const bool vec = transpose_vec_ok(in, in_dtype == TRIBE_F32 ? 4 : 2, s_z1, s_z0, s_r, out, so_z, R_pad);
#define TRIBE_TR(TYPE, VEC)                                                                                                          \
  hipLaunchKernelGGL((transpose_kernel<TYPE, VEC>), grid, dim3(256), 0, (hipStream_t)stream, (const TYPE*)in, R, C, s_z1, Z0, s_z0, s_r, out, \
                     so_z, R_pad)
  if (in_dtype == TRIBE_F32) { if (vec) TRIBE_TR(float, 4); else TRIBE_TR(float, 1); }
  else { if (vec) TRIBE_TR(unsigned short, 4); else TRIBE_TR(unsigned short, 1); }
#undef TRIBE_TR
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_swa_update(const tribe_adam_tensor* table, const int32_t* chunk_tensor, const int64_t* chunk_start, int64_t n_chunks,
                                float weight, void* stream) {
  TRIBE_REQUIRE(table && chunk_tensor && chunk_start, "tribe_swa_update: null pointer");
  TRIBE_REQUIRE(n_chunks > 0 && n_chunks < (1ll << 31), "tribe_swa_update: need 1 .. 2^31 chunks");
  TRIBE_REQUIRE(weight > 0.f && weight <= 1.f, "tribe_swa_update: weight = 1 / (n_averaged + 1) must lie in (0, 1]");
  hipLaunchKernelGGL(swa_update_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_start, weight);
  TRIBE_LAUNCH_CHECK();
  return 0;
}
