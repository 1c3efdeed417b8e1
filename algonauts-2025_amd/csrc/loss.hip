// Loss / metric reductions on [B, V, T'] predictions and targets (T' contiguous).
// pl_module.py:54-56 flattens to [(B T'), V] before the loss; every reduction here is
// a per-voxel (or global) sum, so the flatten is never materialised: one workgroup owns
// one voxel and walks its B rows of T' contiguous floats (coalesced, HBM-bound:
// 2 * B*V*T' * 4 bytes per call).  Sums are carried in f64 so that the
// cov = Sxy - Sx*Sy/n form stays exact to f32 output precision.
#include "common.h"

namespace {

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                          double* __restrict__ partial) {
  __shared__ double sh[4];
  float acc = 0.f;
  double dacc = 0.0;
  int cnt = 0;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = ((const float4*)p)[i], b = ((const float4*)t)[i];
    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
    acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    if (++cnt == 64) { dacc += (double)acc; acc = 0.f; cnt = 0; }
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = p[i] - t[i];
    acc += d * d;
  }
  dacc += (double)acc;
  const double tot = block_sum_d(dacc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void mse_final_kernel(const double* __restrict__ partial, int nparts, int64_t n,
                                                        float* __restrict__ out) {
  __shared__ double sh[4];
  double v = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) v += partial[i];
  const double tot = block_sum_d(v, sh);
  if (threadIdx.x == 0) out[0] = (float)(tot / (double)n);
}

// one workgroup per voxel v; for each row b: 5 sums over t, added into dst[g(b)][v][0..5]
__global__ __launch_bounds__(256) void pearson_stats_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                            int64_t B, int64_t V, int64_t T, int64_t sb, int64_t sv, int64_t st,
                                                            const int64_t* __restrict__ group, int64_t n_groups,
                                                            double* __restrict__ stats) {
  __shared__ double sh[4];
  const int64_t v = blockIdx.x;
  for (int64_t b = 0; b < B; ++b) {
    const float* x = pred + b * sb + v * sv;
    const float* y = truth + b * sb + v * sv;
    double s[5] = {0, 0, 0, 0, 0};
    for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
      const double a = (double)x[t * st], c = (double)y[t * st];
      s[0] += a; s[1] += c; s[2] += a * a; s[3] += c * c; s[4] += a * c;
    }
    double r[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = block_sum_d(s[k], sh);
    if (threadIdx.x == 0) {
      int64_t g = group ? group[b] : 0;
      if (g >= 0 && g < n_groups) {
        double* dst = stats + (g * V + v) * 6;
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[k] += r[k];
        dst[5] += (double)T;
      }
    }
  }
}

__global__ void pearson_from_stats_kernel(const double* __restrict__ stats, int64_t n, float* __restrict__ r) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* s = stats + i * 6;
  const double cnt = s[5];
  if (cnt < 2.0) { r[i] = __builtin_nanf(""); return; }
  const double cov = s[4] - s[0] * s[1] / cnt;
  const double vx = s[2] - s[0] * s[0] / cnt, vy = s[3] - s[1] * s[1] / cnt;
  double v = cov / sqrt(vx * vy);
  if (v > 1.0) v = 1.0;
  if (v < -1.0) v = -1.0;
  r[i] = (float)v;  // 0/0 -> NaN for a constant column, as scipy / torchmetrics give
}

// PearsonLoss (losses.py:17-42): per voxel 1 - cov / (sqrt(Sxx_c) * sqrt(Syy_c) + 1e-8), then mean | sum over voxels
__global__ __launch_bounds__(256) void pearson_loss_final_kernel(const double* __restrict__ stats, int64_t V, int reduction_sum,
                                                                 float* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int64_t v = threadIdx.x; v < V; v += blockDim.x) {
    const double* s = stats + v * 6;
    const double cnt = s[5];
    const double cov = s[4] - s[0] * s[1] / cnt;
    double vx = s[2] - s[0] * s[0] / cnt, vy = s[3] - s[1] * s[1] / cnt;
    vx = vx > 0.0 ? vx : 0.0;
    vy = vy > 0.0 ? vy : 0.0;
    // the reference works in f32: mirror its rounding of the two square roots and of the eps add
    const float xs = sqrtf((float)vx), ys = sqrtf((float)vy);
    const float pcc = (float)cov / (xs * ys + 1e-8f);
    acc += (double)(1.0f - pcc);
  }
  const double tot = block_sum_d(acc, sh);
  if (threadIdx.x == 0) out[0] = (float)(reduction_sum ? tot : tot / (double)V);
}

}  // namespace

extern "C" size_t tribe_mse_workspace_bytes(int64_t n) {
  (void)n;
  return 1024 * sizeof(double);
}

extern "C" int tribe_mse_fwd(const float* pred, const float* truth, int64_t n, float* out, void* workspace,
                             size_t workspace_bytes, void* stream) {
  TRIBE_REQUIRE(pred && truth && out && workspace, "tribe_mse_fwd: null pointer");
  TRIBE_REQUIRE(n > 0, "tribe_mse_fwd: empty input");
  TRIBE_REQUIRE(workspace_bytes >= tribe_mse_workspace_bytes(n), "tribe_mse_fwd: workspace too small");
  TRIBE_REQUIRE(((uintptr_t)pred % 16) == 0 && ((uintptr_t)truth % 16) == 0, "tribe_mse_fwd: inputs must be 16-byte aligned");
  int64_t nb = (n / 4 + 255) / 256;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_partial_kernel, dim3((unsigned)nb), dim3(256), 0, s, pred, truth, n, (double*)workspace);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, (int)nb, n, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_pearson_stats_update(const float* pred, const float* truth, int64_t B, int64_t V, int64_t T, int64_t sb,
                                          int64_t sv, int64_t st, const int64_t* group, int64_t n_groups, double* stats,
                                          void* stream) {
  TRIBE_REQUIRE(pred && truth && stats, "tribe_pearson_stats_update: null pointer");
  TRIBE_REQUIRE(B > 0 && V > 0 && T > 0 && n_groups > 0, "tribe_pearson_stats_update: bad shape B=%lld V=%lld T=%lld groups=%lld",
                (long long)B, (long long)V, (long long)T, (long long)n_groups);
  hipLaunchKernelGGL(pearson_stats_kernel, dim3((unsigned)V), dim3(256), 0, (hipStream_t)stream, pred, truth, B, V, T, sb, sv, st,
                     group, n_groups, stats);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_pearson_from_stats(const double* stats, int64_t n_groups, int64_t V, float* r, void* stream) {
  TRIBE_REQUIRE(stats && r, "tribe_pearson_from_stats: null pointer");
  TRIBE_REQUIRE(n_groups > 0 && V > 0, "tribe_pearson_from_stats: bad shape");
  const int64_t n = n_groups * V;
  hipLaunchKernelGGL(pearson_from_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats, n, r);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t tribe_pearson_loss_workspace_bytes(int64_t V) { return (size_t)V * 6 * sizeof(double); }

extern "C" int tribe_pearson_loss_fwd(const float* pred, const float* truth, int64_t B, int64_t V, int64_t T, int64_t sb,
                                      int64_t sv, int64_t st, int32_t reduction_sum, float* out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  TRIBE_REQUIRE(pred && truth && out && workspace, "tribe_pearson_loss_fwd: null pointer");
  TRIBE_REQUIRE(B > 0 && V > 0 && T > 0, "tribe_pearson_loss_fwd: bad shape");
  TRIBE_REQUIRE(workspace_bytes >= tribe_pearson_loss_workspace_bytes(V), "tribe_pearson_loss_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(workspace, 0, tribe_pearson_loss_workspace_bytes(V), s);
  if (e != hipSuccess) { tribe_set_error("tribe_pearson_loss_fwd: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(pearson_stats_kernel, dim3((unsigned)V), dim3(256), 0, s, pred, truth, B, V, T, sb, sv, st,
                     (const int64_t*)nullptr, (int64_t)1, (double*)workspace);
  hipLaunchKernelGGL(pearson_loss_final_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, V, (int)reduction_sum, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}
