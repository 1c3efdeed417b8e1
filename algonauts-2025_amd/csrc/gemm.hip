// bf16 MFMA GEMM for gfx950:  C = epi(alpha * A . B^T),  A [M,K], B [N,K], K contiguous.
//
// Tile 128x128x64 per 256-thread workgroup (4 waves as 2x2, 64x64 per wave, 4x4
// v_mfma_f32_16x16x32_bf16 accumulators).  Operands are staged HBM -> LDS with
// global_load_lds_dwordx4 (16 B per lane, no VGPR round trip) into two LDS
// buffers; the LDS image is lane-linear (a glds requirement), so the bank-conflict
// swizzle chunk ^= (row & 7) is applied to the per-lane SOURCE address and undone
// on the ds_read_b128 fragment read (cdna_hip_programming.md rule 21 / T2).
// Roofline: MFMA (dense bf16) -- see DESIGN.md "Kernels".
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;       // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;     // A + B
constexpr int SMEM_BYTES = 2 * BUF_BYTES;     // double buffered: 64 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective "each XCD gets a contiguous chunk of tiles" remap (T1); speed only
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// ROLE only gives each call site of the path its own kernel symbol, so that rocprofv3 --stats and the
// in-library HIP-event profile (tribe_prof_*) report per-operator durations; the code is identical.
template <int OUT_BF16, int ROLE>
__global__ __launch_bounds__(256, 2) void gemm_nt_128x128x64(const tribe_gemm_desc g, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int64_t z = blockIdx.y;
  const int64_t b1 = z / g.batch0, b0 = z - b1 * g.batch0;
  const int64_t b1g = g.gather1 ? g.gather1[b1] : b1;
  const unsigned short* A = (const unsigned short*)g.A + (g.gather_a ? b1g : b1) * g.sA1 + b0 * g.sA0;
  const unsigned short* B = (const unsigned short*)g.B + b1 * g.sB1 + b0 * g.sB0;

  // ---- staging addresses: pass p covers tile rows [32p, 32p+32), wave w rows 8w.. of those ----
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;  // swizzle on the source chunk
  const unsigned short* a_src[4];
  const unsigned short* b_src[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = p * 32 + wave * 8 + srow;
    int64_t gr = m0 + r; gr = gr < g.M ? gr : g.M - 1;   // clamp: edge rows re-read a valid row, stores are masked
    int64_t gc = n0 + r; gc = gc < g.N ? gc : g.N - 1;
    a_src[p] = A + gr * g.lda + schunk * 8;
    b_src[p] = B + gc * g.ldb + schunk * 8;
  }

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * BUF_BYTES + wave * 1024;
    const int koff = kt * BK;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[p] + koff), (lptr_t)(base + p * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[p] + koff), (lptr_t)(base + TILE_BYTES + p * 4096), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int a_base = (wr * 64 + frow) * 128;
  const int b_base = TILE_BYTES + (wc * 64 + frow) * 128;

  auto compute = [&](int buf) {
    const char* base = smem + buf * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = (((ks * 4 + fq) ^ (frow & 7)) << 4);
      bf16x8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8_t*)(base + a_base + i * 2048 + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8_t*)(base + b_base + j * 2048 + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  const int nk = (int)(g.K / BK);
  stage(0, 0);
  __syncthreads();  // hipcc emits vmcnt(0) before the barrier while a glds is in flight
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    compute(cur);
    __syncthreads();
  }

  // ---- epilogue: D[row = 4*fq + reg][col = frow] per 16x16 tile ----
  const int64_t bb = g.gather_bias ? b1g : b1;
  const float* bias = g.bias ? g.bias + bb * g.sBias1 : nullptr;
  const float* res = g.res ? g.res + b1 * g.sRes1 + b0 * g.sRes0 : nullptr;
  char* Cb = (char*)g.C;
  const int64_t c_off = b1 * g.sC1 + b0 * g.sC0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t m = m0 + wr * 64 + i * 16 + fq * 4 + reg;
      if (m >= g.M) continue;
      const float* rowadd = g.rowadd ? g.rowadd + (m % g.rowadd_period) * g.ld_rowadd : nullptr;
      const float* gadd = g.gadd ? g.gadd + g.gadd_index[m / g.gadd_div] * g.ld_gadd : nullptr;
      const float brow = (g.bias_mode == TRIBE_BIAS_ROW) ? bias[m] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wc * 64 + j * 16 + frow;
        if (n >= g.N) continue;
        float v = acc[i][j][reg] * g.alpha;
        if (g.bias_mode == TRIBE_BIAS_COL) v += bias[n];
        else if (g.bias_mode == TRIBE_BIAS_ROW) v += brow;
        if (g.act == TRIBE_ACT_GELU) v = gelu_erf(v);
        if (res) {
          const float r = res[m * g.ldres + n];
          v += g.res_scale ? r * g.res_scale[n] : r;
        }
        if (rowadd) v += rowadd[n];
        if (gadd) v += gadd[n];
        const int64_t idx = c_off + m * g.ldc + n;
        if (OUT_BF16) ((unsigned short*)Cb)[idx] = f32_to_bf16(v);
        else ((float*)Cb)[idx] = v;
      }
    }
  }
}

}  // namespace


// ---------------------------------------------------------------------------------------------
// Optional in-library profile: HIP events recorded on the launch stream around every GEMM launch
// while enabled (bench.py brackets its timed region with tribe_prof_begin / tribe_prof_end).
// ---------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
namespace {
struct ProfRec { hipEvent_t start, stop; int role; double flops; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
size_t g_prof_used = 0;
bool g_prof_on = false;

int prof_before(int role, double flops, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (!g_prof_on || g_prof_used >= g_prof.size()) return -1;
  const int slot = (int)g_prof_used++;
  g_prof[slot].role = role;
  g_prof[slot].flops = flops;
  (void)hipEventRecord(g_prof[slot].start, s);
  return slot;
}
void prof_after(int slot, hipStream_t s) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  (void)hipEventRecord(g_prof[slot].stop, s);
}
}  // namespace

extern "C" int tribe_prof_begin(int32_t max_records) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  TRIBE_REQUIRE(max_records > 0, "tribe_prof_begin: max_records must be positive");
  while ((int)g_prof.size() < max_records) {
    ProfRec r{};
    hipError_t e = hipEventCreate(&r.start);
    if (e == hipSuccess) e = hipEventCreate(&r.stop);
    if (e != hipSuccess) { tribe_set_error("tribe_prof_begin: hipEventCreate failed: %s", hipGetErrorString(e)); return (int)e; }
    g_prof.push_back(r);
  }
  g_prof_used = 0;
  g_prof_on = true;
  return 0;
}

extern "C" int tribe_prof_end(int32_t n_roles, double* total_ms_host, int64_t* count_host, double* flops_host) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  TRIBE_REQUIRE(total_ms_host && count_host && flops_host && n_roles >= TRIBE_ROLE_COUNT, "tribe_prof_end: need %d role slots",
                TRIBE_ROLE_COUNT);
  g_prof_on = false;
  for (int i = 0; i < n_roles; ++i) { total_ms_host[i] = 0.0; count_host[i] = 0; flops_host[i] = 0.0; }
  for (size_t i = 0; i < g_prof_used; ++i) {
    hipError_t e = hipEventSynchronize(g_prof[i].stop);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof[i].start, g_prof[i].stop);
    if (e != hipSuccess) { tribe_set_error("tribe_prof_end: event query failed: %s", hipGetErrorString(e)); return (int)e; }
    total_ms_host[g_prof[i].role] += (double)ms;
    count_host[g_prof[i].role] += 1;
    flops_host[g_prof[i].role] += g_prof[i].flops;
  }
  g_prof_used = 0;
  return 0;
}

extern "C" int tribe_gemm_bf16(const tribe_gemm_desc* d, void* stream) {
  TRIBE_REQUIRE(d != nullptr, "tribe_gemm_bf16: null descriptor");
  TRIBE_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "tribe_gemm_bf16: M, N, K must be positive (got %lld %lld %lld)",
                (long long)d->M, (long long)d->N, (long long)d->K);
  TRIBE_REQUIRE(d->K % BK == 0, "tribe_gemm_bf16: K=%lld must be a multiple of %d (zero-pad K)", (long long)d->K, BK);
  TRIBE_REQUIRE(d->batch1 > 0 && d->batch0 > 0, "tribe_gemm_bf16: batch counts must be positive");
  TRIBE_REQUIRE(d->A && d->B && d->C, "tribe_gemm_bf16: null operand");
  TRIBE_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->sA1 % 8 == 0 && d->sA0 % 8 == 0 && d->sB1 % 8 == 0 &&
                    d->sB0 % 8 == 0,
                "tribe_gemm_bf16: lda/ldb/batch strides must be multiples of 8 elements (16-byte rows)");
  TRIBE_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0, "tribe_gemm_bf16: A/B must be 16-byte aligned");
  TRIBE_REQUIRE(d->lda >= d->K && d->ldb >= d->K && d->ldc >= d->N, "tribe_gemm_bf16: leading dimension too small");
  TRIBE_REQUIRE(d->c_dtype == TRIBE_F32 || d->c_dtype == TRIBE_BF16, "tribe_gemm_bf16: c_dtype must be f32 or bf16");
  TRIBE_REQUIRE(d->bias_mode == TRIBE_BIAS_NONE || d->bias != nullptr, "tribe_gemm_bf16: bias_mode set without bias");
  TRIBE_REQUIRE(!d->rowadd || d->rowadd_period > 0, "tribe_gemm_bf16: rowadd needs a positive period");
  TRIBE_REQUIRE(!d->gadd || (d->gadd_index && d->gadd_div > 0), "tribe_gemm_bf16: gadd needs index and divisor");
  TRIBE_REQUIRE(!(d->gather_a || d->gather_bias) || d->gather1, "tribe_gemm_bf16: gather flags set without gather1");
  const int64_t tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN;
  const int64_t nz = d->batch1 * d->batch0;
  TRIBE_REQUIRE(tiles_m * tiles_n < (1ll << 31) && nz < 65536, "tribe_gemm_bf16: grid too large");

  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)nz, 1), block(256, 1, 1);
  hipStream_t s = (hipStream_t)stream;
  const int role = (d->role >= 0 && d->role < TRIBE_ROLE_COUNT) ? d->role : TRIBE_ROLE_GENERIC;
  const double flops = 2.0 * (double)d->M * (double)d->N * (double)d->K * (double)nz;
  const int slot = prof_before(role, flops, s);
#define TRIBE_GEMM_LAUNCH(BF, ROLE)                                                                                  \
  do {                                                                                                               \
    static bool attr_done = false;                                                                                   \
    if (!attr_done) {                                                                                                \
      (void)hipFuncSetAttribute((const void*)gemm_nt_128x128x64<BF, ROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                SMEM_BYTES);                                                                         \
      attr_done = true;                                                                                              \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_nt_128x128x64<BF, ROLE>), grid, block, SMEM_BYTES, s, *d, (int)tiles_m, (int)tiles_n);  \
  } while (0)
  const bool bf = d->c_dtype == TRIBE_BF16;
  switch (role) {
    case TRIBE_ROLE_PROJECTOR: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_GENERIC); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_PROJECTOR); break;
    case TRIBE_ROLE_QKV: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_QKV); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_GENERIC); break;
    case TRIBE_ROLE_ATTN_SCORES: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_GENERIC); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_ATTN_SCORES); break;
    case TRIBE_ROLE_ATTN_PV: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_ATTN_PV); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_GENERIC); break;
    case TRIBE_ROLE_OUT_PROJ: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_GENERIC); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_OUT_PROJ); break;
    case TRIBE_ROLE_FF1: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_FF1); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_GENERIC); break;
    case TRIBE_ROLE_FF2: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_GENERIC); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_FF2); break;
    case TRIBE_ROLE_VOXEL_HEAD: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_GENERIC); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_VOXEL_HEAD); break;
    default: if (bf) TRIBE_GEMM_LAUNCH(1, TRIBE_ROLE_GENERIC); else TRIBE_GEMM_LAUNCH(0, TRIBE_ROLE_GENERIC); break;
  }
#undef TRIBE_GEMM_LAUNCH
  prof_after(slot, s);
  TRIBE_LAUNCH_CHECK();
  return 0;
}
