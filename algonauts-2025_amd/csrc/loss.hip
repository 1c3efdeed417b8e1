// Loss / metric reductions on [B, V, T'] predictions and targets (T' contiguous).
// pl_module.py:54-56 flattens to [(B T'), V] before the loss; every reduction here is
// a per-voxel (or global) sum, so the flatten is never materialised: a workgroup owns
// one voxel of a run of sequences and its waves walk whole rows of T' contiguous floats
// (coalesced float4 streams, HBM-bound: 2 * B*V*T' * 4 bytes per call, the roofline
// these kernels are measured against in scripts/loss_bench.py).  Sums are carried in f64
// so that the cov = Sxy - Sx*Sy/n form stays exact to f32 output precision.
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

// HBM-bound: 8 bytes read per element.  Four independent float4 pairs are requested before any is consumed (128 B in
// flight per lane; 2048 workgroups x 256 lanes cover the ~16 MB the chip needs in flight at 8 TB/s), f32 partial sums of
// at most 16 squared differences are folded into an f64 carry.
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                          double* __restrict__ partial) {
  __shared__ double sh[4];
  double dacc = 0.0;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float4* p4 = (const float4*)p;
  const float4* t4 = (const float4*)t;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = load_nt_f4(p4 + i + u * stride); b[u] = load_nt_f4(t4 + i + u * stride); }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float d0 = a[u].x - b[u].x, d1 = a[u].y - b[u].y, d2 = a[u].z - b[u].z, d3 = a[u].w - b[u].w;
      acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    dacc += (double)acc;
  }
  for (; i < n4; i += stride) {
    const float4 a = p4[i], b = t4[i];
    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
    dacc += (double)(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
  }
  float tail = 0.f;
  for (int64_t j = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const float d = p[j] - t[j];
    tail += d * d;
  }
  dacc += (double)tail;
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

// Strided fallback (any st): one workgroup per voxel v; for each row b: 5 sums over t, added into dst[g(b)][v][0..5]
__global__ __launch_bounds__(256) void pearson_stats_strided_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
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

// Contiguous rows (st == 1, 16-byte aligned, T % 4 == 0) -- the layout the voxel head writes.  HBM-bound: 8 bytes read per
// (row, voxel, t).  Workgroup (v, chunk) owns voxel v of `rows_per_wg` consecutive sequences; each of its four waves walks
// whole rows (T contiguous floats per tensor: float4 per lane, two rows-worth of requests in flight) and keeps the five f64
// sums IN REGISTERS across rows of the same group; only a change of group (grouped metric: one Pearson state per subject,
// metrics/base.py:39-91) or the end of the chunk costs a wave reduction and six f64 atomics.  No LDS, no barrier.
__device__ __forceinline__ void pearson_flush(double (&s)[5], double cnt, double* __restrict__ dst) {
#pragma unroll
  for (int k = 0; k < 5; ++k) s[k] = wave_sum_d(s[k]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) unsafeAtomicAdd(dst + k, s[k]);
    unsafeAtomicAdd(dst + 5, cnt);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) s[k] = 0.0;
}

__device__ __forceinline__ void pearson_acc4(double (&s)[5], const float4& a, const float4& c) {
  // f32 products of one float4 pair are exact in f64 after widening; five f64 fma chains per lane
  const double ax = a.x, ay = a.y, az = a.z, aw = a.w, cx = c.x, cy = c.y, cz = c.z, cw = c.w;
  s[0] += (ax + ay) + (az + aw);
  s[1] += (cx + cy) + (cz + cw);
  s[2] = fma(ax, ax, fma(ay, ay, fma(az, az, fma(aw, aw, s[2]))));
  s[3] = fma(cx, cx, fma(cy, cy, fma(cz, cz, fma(cw, cw, s[3]))));
  s[4] = fma(ax, cx, fma(ay, cy, fma(az, cz, fma(aw, cw, s[4]))));
}

__global__ __launch_bounds__(256) void pearson_stats_rows_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                                 int64_t B, int64_t V, int64_t T, int64_t sb, int64_t sv,
                                                                 const int64_t* __restrict__ group, int64_t n_groups,
                                                                 int64_t rows_per_wg, double* __restrict__ stats) {
  const int64_t v = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b0 = (int64_t)blockIdx.y * rows_per_wg;
  const int64_t b1 = (b0 + rows_per_wg < B) ? b0 + rows_per_wg : B;
  const int64_t T4 = T >> 2;
  double s[5] = {0, 0, 0, 0, 0};
  double cnt = 0.0;
  int64_t g_cur = -1;
  for (int64_t b = b0 + wave; b < b1; b += 4) {
    int64_t g = group ? group[b] : 0;
    if (g < 0 || g >= n_groups) continue;                       // rows of an unknown group are skipped, as before
    if (g != g_cur) {
      if (g_cur >= 0) pearson_flush(s, cnt, stats + (g_cur * V + v) * 6);
      g_cur = g;
      cnt = 0.0;
    }
    const float4* x = (const float4*)(pred + b * sb + v * sv);
    const float4* y = (const float4*)(truth + b * sb + v * sv);
    int64_t i = lane;
    for (; i + 192 < T4; i += 256) {                            // four float4 pairs requested before the first is used
      float4 a[4], c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = load_nt_f4(x + i + 64 * u); c[u] = load_nt_f4(y + i + 64 * u); }
#pragma unroll
      for (int u = 0; u < 4; ++u) pearson_acc4(s, a[u], c[u]);
    }
    for (; i < T4; i += 64) {
      const float4 a = load_nt_f4(x + i), c = load_nt_f4(y + i);
      pearson_acc4(s, a, c);
    }
    cnt += (double)T;
  }
  if (g_cur >= 0) pearson_flush(s, cnt, stats + (g_cur * V + v) * 6);
}

static void launch_pearson_stats(const float* pred, const float* truth, int64_t B, int64_t V, int64_t T, int64_t sb, int64_t sv, int64_t st,
                                 const int64_t* group, int64_t n_groups, double* stats, hipStream_t s) {
  const bool rows = st == 1 && T % 4 == 0 && sb % 4 == 0 && sv % 4 == 0 && ((uintptr_t)pred % 16) == 0 && ((uintptr_t)truth % 16) == 0 &&
                    V <= 0x7fffffff;
  if (!rows) {
    hipLaunchKernelGGL(pearson_stats_strided_kernel, dim3((unsigned)V), dim3(256), 0, s, pred, truth, B, V, T, sb, sv, st, group, n_groups, stats);
    return;
  }
  // >= ~4096 workgroups when the batch allows it (256 CUs x 8 resident), at least 4 rows (one per wave) per workgroup
  int64_t chunks = (4096 + V - 1) / V;
  const int64_t max_chunks = (B + 3) / 4;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  if (chunks > 65535) chunks = 65535;
  const int64_t rows_per_wg = (B + chunks - 1) / chunks;
  chunks = (B + rows_per_wg - 1) / rows_per_wg;
  hipLaunchKernelGGL(pearson_stats_rows_kernel, dim3((unsigned)V, (unsigned)chunks), dim3(256), 0, s, pred, truth, B, V, T, sb, sv, group,
                     n_groups, rows_per_wg, stats);
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
  return 2048 * sizeof(double);
}

extern "C" int tribe_mse_fwd(const float* pred, const float* truth, int64_t n, float* out, void* workspace,
                             size_t workspace_bytes, void* stream) {
  TRIBE_REQUIRE(pred && truth && out && workspace, "tribe_mse_fwd: null pointer");
  TRIBE_REQUIRE(n > 0, "tribe_mse_fwd: empty input");
  TRIBE_REQUIRE(workspace_bytes >= tribe_mse_workspace_bytes(n), "tribe_mse_fwd: workspace too small");
  TRIBE_REQUIRE(((uintptr_t)pred % 16) == 0 && ((uintptr_t)truth % 16) == 0, "tribe_mse_fwd: inputs must be 16-byte aligned");
  int64_t nb = (n / 4 + 1023) / 1024;    // four float4 per lane and trip
  if (nb > 2048) nb = 2048;              // tribe_mse_workspace_bytes: 2048 partial sums
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
  launch_pearson_stats(pred, truth, B, V, T, sb, sv, st, group, n_groups, stats, (hipStream_t)stream);
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
  launch_pearson_stats(pred, truth, B, V, T, sb, sv, st, (const int64_t*)nullptr, (int64_t)1, (double*)workspace, s);
  hipLaunchKernelGGL(pearson_loss_final_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, V, (int)reduction_sum, out);
  TRIBE_LAUNCH_CHECK();
  return 0;
}
