// Segment assembly from HBM-resident extractor outputs: the byte-moving step right
// before the projector GEMMs.  The reference does this on the host, per segment and per feature, with numpy
// (`_aggregate_layers`: features/text.py:129-149; `TimedArray.overlap` / `+=`: base.py:130-211), then copies fp32
// [B, L, D, T] tensors to the device and transposes them there (model.py:146-155).  Here the cache lives in HBM and
// one launch per modality writes the bf16 [B*T, C_pad] rows the projector GEMM reads.
//
// All three kernels are HBM-bound copies / short reductions: roofline = HBM bytes (read source + write output).
#include "common.h"

namespace {

// ---- layer aggregation: out[b][g][i] = mean_{s in [lo[g], hi[g])} in[b][s][i] -------------------------------------
// Sequential fp32 adds in layer order and one fp32 divide by the count: the order numpy uses for `.mean(0)` over the
// leading axis, so the result is bit-identical to the reference's group_mean (groups of one = plain index select).
template <int VEC>
__global__ __launch_bounds__(256) void group_mean_kernel(const float* __restrict__ in, int64_t batch, int64_t n_states, int64_t plane,
                                                         const int32_t* __restrict__ lo, const int32_t* __restrict__ hi,
                                                         int32_t n_groups, float* __restrict__ out) {
  const int64_t pv = plane / VEC;
  const int64_t total = batch * n_groups * pv;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx % pv;
    const int64_t g = (idx / pv) % n_groups;
    const int64_t b = idx / (pv * n_groups);
    const int s0 = lo[g], s1 = hi[g];
    const float* src = in + (b * n_states + s0) * plane + i * VEC;
    float acc[VEC];
    if (VEC == 4) {
      const float4 v = *(const float4*)src;
      acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w;
    } else {
      acc[0] = src[0];
    }
    for (int s = s0 + 1; s < s1; ++s) {
      src += plane;
      if (VEC == 4) {
        const float4 v = *(const float4*)src;
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      } else {
        acc[0] += src[0];
      }
    }
    const float cnt = (float)(s1 - s0);
    float* dst = out + (b * n_groups + g) * plane + i * VEC;
    if (VEC == 4) {
      *(float4*)dst = make_float4(acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt);
    } else {
      dst[0] = acc[0] / cnt;
    }
  }
}

// ---- sampled features: sum of time slices -> packed bf16 rows ------------------------------------------------------
// One workgroup builds a 64-channel x 64-step tile of segment b: every thread owns 16 (channel, step) cells, walks the
// segment's pieces in order (sum order of `out += piece`) with loads coalesced along time, then the tile goes through
// LDS so that the stores are coalesced along channels (128 B of bf16 per output row).
__global__ __launch_bounds__(256) void segment_gather_packed_kernel(const tribe_feature_piece* __restrict__ pieces,
                                                                    const int32_t* __restrict__ seg_ptr, int64_t C, int64_t T,
                                                                    unsigned short* __restrict__ out, int64_t C_pad) {
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;
  const int64_t b = blockIdx.z;
  const int64_t t0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int64_t t = t0 + tx;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int p0 = seg_ptr[b], p1 = seg_ptr[b + 1];
  for (int p = p0; p < p1; ++p) {
    const tribe_feature_piece pc = pieces[p];
    if (pc.dst_first >= t0 + 64 || pc.dst_first + pc.dst_count <= t0) continue;  // uniform: piece misses this tile
    const int64_t rel = t - pc.dst_first;
    if (rel < 0 || rel >= pc.dst_count || t >= T) continue;
    const float* src = pc.src + pc.src_first + (pc.src_count == 1 ? 0 : rel);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int64_t c = c0 + ty + 4 * i;
      if (c < C) acc[i] += src[c * pc.ld];
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[ty + 4 * i][tx] = acc[i];
  __syncthreads();
  const int tl = tid >> 2, cq = (tid & 3) * 16;
  const int64_t row = t0 + tl;
  if (row >= T) return;
  unsigned short* dst = out + (b * T + row) * C_pad + c0 + cq;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (c0 + cq + h * 8 >= C_pad) break;
    u16x8_t o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cl = cq + h * 8 + k;
      o[k] = (c0 + cl < C) ? f32_to_bf16(tile[cl][tl]) : (unsigned short)0;
    }
    *(u16x8_t*)(dst + h * 8) = o;
  }
}

// ---- sampled features, reference layout: out f32 [B, C, T] (targets such as fMRI; no transpose) --------------------
__global__ __launch_bounds__(256) void segment_gather_rows_kernel(const tribe_feature_piece* __restrict__ pieces,
                                                                  const int32_t* __restrict__ seg_ptr, int64_t C, int64_t T,
                                                                  float* __restrict__ out) {
  const int64_t b = blockIdx.z, c = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float acc = 0.f;
  const int p0 = seg_ptr[b], p1 = seg_ptr[b + 1];
  for (int p = p0; p < p1; ++p) {
    const tribe_feature_piece pc = pieces[p];
    const int64_t rel = t - pc.dst_first;
    if (rel < 0 || rel >= pc.dst_count) continue;
    acc += pc.src[c * pc.ld + pc.src_first + (pc.src_count == 1 ? 0 : rel)];
  }
  out[(b * C + c) * T + t] = acc;
}

// ---- word features: out[row] = sum of the table rows listed for (segment, step) `row`, as bf16 ---------------------
// CSR lists keep the reference's order (event order inside the segment), so the fp32 sum is the reference's sum.
template <int VEC, bool F32OUT>
__global__ __launch_bounds__(256) void word_bag_kernel(const float* __restrict__ table, int64_t C, const int32_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ word_idx, void* __restrict__ out_v, int64_t C_pad) {
  const int64_t row = blockIdx.x;
  const int w0 = row_ptr[row], w1 = row_ptr[row + 1];
  unsigned short* dst = (unsigned short*)out_v + row * C_pad;
  float* dst32 = (float*)out_v + row * C_pad;
  for (int64_t q = threadIdx.x; q < C_pad / VEC; q += 256) {
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    if (q * VEC < C) {
      for (int w = w0; w < w1; ++w) {
        const float* src = table + (int64_t)word_idx[w] * C + q * VEC;
        if (VEC == 4) {
          const float4 v = *(const float4*)src;
          acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
        } else {
          acc[0] += src[0];
        }
      }
    }
    if (F32OUT) {
      if (VEC == 4) *(float4*)(dst32 + q * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      else dst32[q] = acc[0];
    } else if (VEC == 4) {
      u16x4_t o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = f32_to_bf16(acc[k]);
      *(u16x4_t*)(dst + q * 4) = o;
    } else {
      dst[q] = f32_to_bf16(acc[0]);
    }
  }
}


// ---- predictions for the submission writer: out[z][c][r] = in[z][r][c] (f32), 64x64 tiles through LDS ----------------
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, int64_t R, int64_t C, float* __restrict__ out) {
  __shared__ float tile[64][65];
  const int64_t z = blockIdx.z;
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = in + z * R * C;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    if (r < R && c < C) tile[i][tx] = src[r * C + c];
  }
  __syncthreads();
  float* dst = out + z * R * C;
#pragma unroll 4
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (c < C && r < R) dst[c * R + r] = tile[tx][i];
  }
}

// ---- ensemble averaging (grids/average_submissions.py:107-125) -----------------------------------------------------
// out[i] = sum_n preds[n][i] * w, sequential in n with a separately rounded product and sum (no FMA contraction), which
// is what numpy's `np.sum(preds * weights, axis=0)` computes:
//   PER_COLUMN = 1: w = wcol[n * V + i % V], all f32 (per-voxel softmax weights, :96-99);
//   PER_COLUMN = 0: w = wsub[n] in f64, product and sum in f64, f64 output (score weights, :100-103).
template <int PER_COLUMN>
__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* __restrict__ preds, int64_t N, int64_t M, int64_t V,
                                                           const float* __restrict__ wcol, const double* __restrict__ wsub,
                                                           void* __restrict__ out) {
#pragma clang fp contract(off)   // hipcc contracts a * b + c into an FMA by default; numpy rounds the product first
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    if (PER_COLUMN) {
      const int64_t v = i % V;
      float acc = preds[i] * wcol[v];
      for (int64_t n = 1; n < N; ++n) {
        const float prod = preds[n * M + i] * wcol[n * V + v];
        acc = acc + prod;
      }
      ((float*)out)[i] = acc;
    } else {
      double acc = (double)preds[i] * wsub[0];
      for (int64_t n = 1; n < N; ++n) {
        const double prod = (double)preds[n * M + i] * wsub[n];
        acc = acc + prod;
      }
      ((double*)out)[i] = acc;
    }
  }
}

// ---- correlation matrix of N prediction sets (np.corrcoef over rows, average_submissions.py:38-53) ------------------
// Pass 1: row sums -> means (f64).  Pass 2: every workgroup takes a K slice, accumulates the N x N products of the
// centred rows in f64 (thread (i, j) owns one pair) and adds its partial to the output with f64 atomics.
__global__ __launch_bounds__(256) void row_sum_f64_kernel(const float* __restrict__ x, int64_t K, double* __restrict__ sums) {
  const int64_t n = blockIdx.y;
  const int64_t per = (K + gridDim.x - 1) / gridDim.x;
  const int64_t k0 = (int64_t)blockIdx.x * per, k1 = (k0 + per < K) ? k0 + per : K;
  double acc = 0.0;
  for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) acc += (double)x[n * K + k];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(&sums[n], acc);
}

constexpr int CORR_MAX_N = 64, CORR_CHUNK = 64;
__global__ __launch_bounds__(256) void corr_accumulate_kernel(const float* __restrict__ x, int64_t N, int64_t K, const double* __restrict__ sums,
                                                              double* __restrict__ cov) {
  __shared__ double tile[CORR_MAX_N][CORR_CHUNK + 1];
  const int64_t per = ((K + gridDim.x - 1) / gridDim.x + CORR_CHUNK - 1) / CORR_CHUNK * CORR_CHUNK;
  const int64_t k0 = (int64_t)blockIdx.x * per, k1 = (k0 + per < K) ? k0 + per : K;
  const int n_pairs = (int)(N * N);
  double acc[16];                                  // pairs p = threadIdx.x + 256 * q, q < 16  (N <= 64 -> <= 4096 pairs)
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  for (int64_t kc = k0; kc < k1; kc += CORR_CHUNK) {
    for (int e = threadIdx.x; e < N * CORR_CHUNK; e += 256) {
      const int n = e / CORR_CHUNK, kk = e % CORR_CHUNK;
      const int64_t k = kc + kk;
      tile[n][kk] = (k < k1) ? (double)x[n * K + k] - sums[n] / (double)K : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int p = threadIdx.x + 256 * q;
      if (p < n_pairs) {
        const int i = p / (int)N, j = p % (int)N;
        double a = 0.0;
        for (int kk = 0; kk < CORR_CHUNK; ++kk) a += tile[i][kk] * tile[j][kk];
        acc[q] += a;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int p = threadIdx.x + 256 * q;
    if (p < n_pairs) atomicAdd(&cov[p], acc[q]);
  }
}

__global__ void corr_finalize_kernel(const double* __restrict__ cov, int64_t N, double* __restrict__ corr) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N * N) return;
  const int i = p / (int)N, j = p % (int)N;
  corr[p] = cov[p] / sqrt(cov[i * N + i] * cov[j * N + j]);
}

inline unsigned grid_1d(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  if (g > 65536 * 16) g = 65536 * 16;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int tribe_group_mean_fwd(const float* states, int64_t batch, int64_t n_states, int64_t plane, const int32_t* lo,
                                    const int32_t* hi, int32_t n_groups, float* out, void* stream) {
  TRIBE_REQUIRE(states && lo && hi && out, "tribe_group_mean_fwd: null pointer");
  TRIBE_REQUIRE(batch > 0 && n_states > 0 && plane > 0 && n_groups > 0, "tribe_group_mean_fwd: bad shape batch=%lld n_states=%lld plane=%lld groups=%d",
                (long long)batch, (long long)n_states, (long long)plane, n_groups);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = plane % 4 == 0 && ((uintptr_t)states % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (vec)
    hipLaunchKernelGGL(group_mean_kernel<4>, dim3(grid_1d(batch * n_groups * (plane / 4), 256)), dim3(256), 0, s, states, batch, n_states,
                       plane, lo, hi, n_groups, out);
  else
    hipLaunchKernelGGL(group_mean_kernel<1>, dim3(grid_1d(batch * n_groups * plane, 256)), dim3(256), 0, s, states, batch, n_states, plane,
                       lo, hi, n_groups, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_segment_gather_fwd(const tribe_feature_piece* pieces, const int32_t* seg_ptr, int64_t B, int64_t C, int64_t T,
                                        void* out, int32_t out_dtype, int64_t C_pad, void* stream) {
  TRIBE_REQUIRE(pieces && seg_ptr && out, "tribe_segment_gather_fwd: null pointer");
  TRIBE_REQUIRE(B > 0 && B < 65536 && C > 0 && T > 0, "tribe_segment_gather_fwd: bad shape B=%lld C=%lld T=%lld", (long long)B, (long long)C,
                (long long)T);
  hipStream_t s = (hipStream_t)stream;
  if (out_dtype == TRIBE_BF16) {
    TRIBE_REQUIRE(C_pad >= C && C_pad % 8 == 0 && ((uintptr_t)out % 16) == 0,
                  "tribe_segment_gather_fwd: C_pad=%lld must be >= C=%lld, a multiple of 8, and out 16-byte aligned", (long long)C_pad,
                  (long long)C);
    TRIBE_REQUIRE((T + 63) / 64 < 65536, "tribe_segment_gather_fwd: T too large for one launch");
    dim3 grid((unsigned)((C_pad + 63) / 64), (unsigned)((T + 63) / 64), (unsigned)B);
    hipLaunchKernelGGL(segment_gather_packed_kernel, grid, dim3(256), 0, s, pieces, seg_ptr, C, T, (unsigned short*)out, C_pad);
  } else if (out_dtype == TRIBE_F32) {
    TRIBE_REQUIRE(C < 65536, "tribe_segment_gather_fwd: C too large for the [B, C, T] layout launch");
    dim3 grid((unsigned)((T + 255) / 256), (unsigned)C, (unsigned)B);
    hipLaunchKernelGGL(segment_gather_rows_kernel, grid, dim3(256), 0, s, pieces, seg_ptr, C, T, (float*)out);
  } else {
    TRIBE_REQUIRE(false, "tribe_segment_gather_fwd: out_dtype must be bf16 (packed [B,T,C_pad]) or f32 ([B,C,T])");
  }
  TRIBE_LAUNCH_CHECK();
  return 0;
}

static int word_bag_launch(const char* who, const float* table, int64_t n_words, int64_t C, const int32_t* row_ptr, const int32_t* word_idx,
                           int64_t rows, void* out, int64_t C_pad, bool f32out, void* stream) {
  TRIBE_REQUIRE(table && row_ptr && word_idx && out, "%s: null pointer", who);
  TRIBE_REQUIRE(n_words > 0 && C > 0 && rows > 0 && rows < 2147483647LL, "%s: bad shape words=%lld C=%lld rows=%lld", who,
                (long long)n_words, (long long)C, (long long)rows);
  TRIBE_REQUIRE(C_pad >= C && (f32out || C_pad % 8 == 0), "%s: C_pad=%lld must be >= C=%lld (and a multiple of 8 for bf16 rows)", who,
                (long long)C_pad, (long long)C);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = C % 4 == 0 && C_pad % 4 == 0 && ((uintptr_t)table % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (vec && f32out)
    hipLaunchKernelGGL((word_bag_kernel<4, true>), dim3((unsigned)rows), dim3(256), 0, s, table, C, row_ptr, word_idx, out, C_pad);
  else if (vec)
    hipLaunchKernelGGL((word_bag_kernel<4, false>), dim3((unsigned)rows), dim3(256), 0, s, table, C, row_ptr, word_idx, out, C_pad);
  else if (f32out)
    hipLaunchKernelGGL((word_bag_kernel<1, true>), dim3((unsigned)rows), dim3(256), 0, s, table, C, row_ptr, word_idx, out, C_pad);
  else
    hipLaunchKernelGGL((word_bag_kernel<1, false>), dim3((unsigned)rows), dim3(256), 0, s, table, C, row_ptr, word_idx, out, C_pad);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_word_bag_fwd(const float* table, int64_t n_words, int64_t C, const int32_t* row_ptr, const int32_t* word_idx,
                                  int64_t rows, uint16_t* out, int64_t C_pad, void* stream) {
  return word_bag_launch("tribe_word_bag_fwd", table, n_words, C, row_ptr, word_idx, rows, out, C_pad, false, stream);
}

extern "C" int tribe_word_bag_f32_fwd(const float* table, int64_t n_words, int64_t C, const int32_t* row_ptr, const int32_t* word_idx,
                                      int64_t rows, float* out, void* stream) {
  return word_bag_launch("tribe_word_bag_f32_fwd", table, n_words, C, row_ptr, word_idx, rows, out, C, true, stream);
}

extern "C" int tribe_transpose_f32_fwd(const float* in, int64_t Z, int64_t R, int64_t C, float* out, void* stream) {
  TRIBE_REQUIRE(in && out && in != out, "tribe_transpose_f32_fwd: null or aliased pointer");
  TRIBE_REQUIRE(Z > 0 && Z < 65536 && R > 0 && C > 0 && (R + 63) / 64 < 65536, "tribe_transpose_f32_fwd: bad shape Z=%lld R=%lld C=%lld", (long long)Z,
                (long long)R, (long long)C);
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64), (unsigned)Z);
  hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, R, C, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_weighted_sum_fwd(const float* preds, int64_t N, int64_t M, int64_t V, const float* w_column, const double* w_set,
                                      void* out, void* stream) {
  TRIBE_REQUIRE(preds && out, "tribe_weighted_sum_fwd: null pointer");
  TRIBE_REQUIRE((w_column != nullptr) != (w_set != nullptr), "tribe_weighted_sum_fwd: give exactly one of w_column (f32 [N, V]) / w_set (f64 [N])");
  TRIBE_REQUIRE(N > 0 && M > 0 && V > 0 && M % V == 0, "tribe_weighted_sum_fwd: bad shape N=%lld M=%lld V=%lld", (long long)N, (long long)M, (long long)V);
  hipStream_t s = (hipStream_t)stream;
  if (w_column)
    hipLaunchKernelGGL(weighted_sum_kernel<1>, dim3(grid_1d(M, 256)), dim3(256), 0, s, preds, N, M, V, w_column, w_set, out);
  else
    hipLaunchKernelGGL(weighted_sum_kernel<0>, dim3(grid_1d(M, 256)), dim3(256), 0, s, preds, N, M, V, w_column, w_set, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t tribe_corr_matrix_workspace_bytes(int64_t N) { return (size_t)(N + N * N) * sizeof(double); }

extern "C" int tribe_corr_matrix_fwd(const float* x, int64_t N, int64_t K, double* corr, void* workspace, size_t workspace_bytes, void* stream) {
  TRIBE_REQUIRE(x && corr && workspace, "tribe_corr_matrix_fwd: null pointer");
  TRIBE_REQUIRE(N > 0 && N <= CORR_MAX_N && K > 1, "tribe_corr_matrix_fwd: N=%lld must be in [1, %d], K=%lld > 1", (long long)N, CORR_MAX_N, (long long)K);
  TRIBE_REQUIRE(workspace_bytes >= tribe_corr_matrix_workspace_bytes(N), "tribe_corr_matrix_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  double* sums = (double*)workspace;
  double* cov = sums + N;
  hipError_t e = hipMemsetAsync(workspace, 0, tribe_corr_matrix_workspace_bytes(N), s);
  if (e != hipSuccess) { tribe_set_error("tribe_corr_matrix_fwd: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  int64_t slices = (K + 65535) / 65536;
  if (slices > 1024) slices = 1024;
  hipLaunchKernelGGL(row_sum_f64_kernel, dim3((unsigned)slices, (unsigned)N), dim3(256), 0, s, x, K, sums);
  int64_t blocks = (K + 8191) / 8192;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(corr_accumulate_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, N, K, sums, cov);
  hipLaunchKernelGGL(corr_finalize_kernel, dim3((unsigned)((N * N + 255) / 256)), dim3(256), 0, s, cov, N, corr);
  TRIBE_LAUNCH_CHECK();
  return 0;
}
