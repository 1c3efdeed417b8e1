// fp8 (OCP e4m3) MFMA GEMM for gfx950:  C = epi(alpha * A . B^T),  A [M,K], B [N,K] e4m3 bytes, K contiguous; alpha carries
// the product of the two per-tensor quantisation scales (BASELINE config 5: "fp8 MFMA extractor GEMMs").
//
// Same 256 x 256 tile, 8-wave 2x4 layout, 2-buffer LDS-DMA pipeline with counted vmcnt and staggered wave groups as the bf16
// kernel in gemm.hip -- a K-tile of 128 e4m3 values is the SAME 128-byte-per-row LDS image as 64 bf16 values, so staging,
// swizzle and barriers are byte-identical.  What changes is the matrix instruction: one
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit E8M0 scales, formats 0 = e4m3) per 16 x 16 x 128 step instead of two
// v_mfma_f32_16x16x32_bf16 per 16 x 16 x 64: 4x the K per instruction at 2x the cycles = 2x the bf16 rate
// (MI355X_MICROARCH.md, matrix cores).  Each lane feeds 32 consecutive k of its row (8 VGPRs): the two adjacent
// 16-byte LDS chunks 2*(lane>>4) and 2*(lane>>4)+1 (any k-permutation shared by A and B is valid for the instruction;
// measured with ab_tmp probe in round 1).  Roofline: MFMA (dense fp8, 5 PFLOP/s).
#include "gemm_common.h"

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

namespace {

constexpr int UNIT_SCALE = 0x7F7F7F7F;   // E8M0 exponent 127 = 2^0 in every byte
namespace big {
constexpr int BM = 256, BN = 256;
constexpr int A_BYTES = BM * 128;             // 32 KiB
constexpr int BUF_BYTES = 2 * A_BYTES;        // A + B of one K-tile: 64 KiB
constexpr int SMEM_BYTES = 2 * BUF_BYTES;     // 128 KiB
}  // namespace big

#define TRIBE_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define TRIBE_STAMP(var) do { } while (0)
#define TRIBE_STAMP_ACC(slot, t_from, t_to) do { } while (0)

template <int OUT_BF16, int EXT>
__global__ __launch_bounds__(512, 2) void gemm_fp8_nt_256x256x128(const tribe_gemm_desc g, int tiles_m, int tiles_n) {
  using namespace big;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int wr = wave >> 2, wc = wave & 3;

  int tm, tn;
  tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int64_t z = blockIdx.y;
  const int64_t b1 = z / g.batch0, b0 = z - b1 * g.batch0;
  const int64_t b1g = g.gather1 ? g.gather1[b1] : b1;
  const unsigned char* A = (const unsigned char*)g.A + (g.gather_a ? b1g : b1) * g.sA1 + b0 * g.sA0;   // e4m3 bytes, K contiguous
  const unsigned char* B = (const unsigned char*)g.B + (g.gather_b ? b1g : b1) * g.sB1 + b0 * g.sB0;

  // ---- staging: half-tile = the 128 tile rows read in one phase; each wave moves 2 slabs of 8 rows ----
  //   A half h: rows {wr'*128 + h*64 + 0..63, wr' = 0,1};   slab j of this wave: row0 = j*128 + h*64 + wave*8
  //   B half h: rows {wc'*64 + h*32 + 0..31, wc' = 0..3};   slab j of this wave: row0 = (2j + (wave>>2))*64 + h*32 + (wave&3)*8
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;  // swizzle on the SOURCE chunk (row & 7 == srow for every slab)
  const unsigned char* a_src[2][2];
  const unsigned char* b_src[2][2];
  int a_lds[2][2], b_lds[2][2];  // wave-uniform LDS byte offsets of the slabs inside a K-tile buffer
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ra = j * 128 + h * 64 + wave * 8;
      const int rb = (2 * j + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8;
      int64_t gr = m0 + ra + srow; gr = gr < g.M ? gr : g.M - 1;  // clamp: edge rows re-read a valid row, stores are masked
      int64_t gc = n0 + rb + srow; gc = gc < g.N ? gc : g.N - 1;
      a_src[h][j] = A + gr * g.lda + schunk * 16;
      b_src[h][j] = B + gc * g.ldb + schunk * 16;
      a_lds[h][j] = ra * 128;
      b_lds[h][j] = A_BYTES + rb * 128;
    }

  // which: 0 = A half 0, 1 = B half 0, 2 = B half 1, 3 = A half 1
  auto stage = [&](int which, int buf, int kt) {
    char* base = smem + buf * BUF_BYTES;
    const int koff = kt * 128;   // one K-tile = 128 e4m3 = 128 bytes per row: the same LDS image as 64 bf16
    if (which == 0 || which == 3) {
      const int h = which == 3;
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[h][0] + koff), (lptr_t)(base + a_lds[h][0]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[h][1] + koff), (lptr_t)(base + a_lds[h][1]), 16, 0, 0);
    } else {
      const int h = which == 2;
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[h][0] + koff), (lptr_t)(base + b_lds[h][0]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[h][1] + koff), (lptr_t)(base + b_lds[h][1]), 16, 0, 0);
    }
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int coff0 = (((2 * fq) ^ (frow & 7)) << 4), coff1 = (((2 * fq + 1) ^ (frow & 7)) << 4);   // k = 32 fq .. 32 fq + 31
  const int a_rd = (wr * 128 + frow) * 128;            // + mh*8192 + i*2048 + coff
  const int b_rd = A_BYTES + (wc * 64 + frow) * 128;   // + nh*4096 + j*2048 + coff

  i32x8_t fa[4], fb0[2], fb1[2];

#define TRIBE_LDS_A(base, MH)                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
    fa[i].lo = *(const i32x4_t*)((base) + a_rd + (MH) * 8192 + i * 2048 + coff0);              \
    fa[i].hi = *(const i32x4_t*)((base) + a_rd + (MH) * 8192 + i * 2048 + coff1);              \
  }
#define TRIBE_LDS_B(base, NH, FB)                                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
    FB[j].lo = *(const i32x4_t*)((base) + b_rd + (NH) * 4096 + j * 2048 + coff0);              \
    FB[j].hi = *(const i32x4_t*)((base) + b_rd + (NH) * 4096 + j * 2048 + coff1);              \
  }
#define TRIBE_MMA(MH, NH, FB)                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
    acc[(MH) * 4 + i][(NH) * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(        \
        fa[i], FB[j], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0, UNIT_SCALE, 0, UNIT_SCALE);    \
  }
// The fragment reads are retired BEFORE the barrier: the two wave groups (wr = 0 / 1 = the two waves of every
// SIMD) run one barrier apart, so while one group sits in this wait the other group's MFMA cluster owns the
// matrix pipe, and at every barrier all LDS reads issued so far are complete (restaging is then hazard-free).
#define TRIBE_PHASE_SYNC_MMA(MH, NH, FB)        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);            \
  __builtin_amdgcn_s_barrier();                 \
  __builtin_amdgcn_s_setprio(1);                \
  TRIBE_MMA(MH, NH, FB)                         \
  __builtin_amdgcn_s_setprio(0);                \
  __builtin_amdgcn_s_barrier();

#define TRIBE_PHASE_SYNC_MMA2(MH0, NH0, FB0, MH1, NH1, FB1) \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);                       \
  __builtin_amdgcn_s_barrier();                            \
  __builtin_amdgcn_s_setprio(1);                           \
  TRIBE_MMA(MH0, NH0, FB0)                                 \
  TRIBE_MMA(MH1, NH1, FB1)                                 \
  __builtin_amdgcn_s_setprio(0);                           \
  __builtin_amdgcn_s_barrier();                            \

  const int nk = (int)(g.K / 128);

  // ---- prologue: K-tile 0 completely, K-tile 1 minus its last half-tile ----
  stage(0, 0, 0); stage(1, 0, 0); stage(2, 0, 0); stage(3, 0, 0);
  if (nk > 1) {
    stage(0, 1, 1); stage(1, 1, 1); stage(2, 1, 1);
    TRIBE_WAIT_VMCNT(6);
  } else {
    TRIBE_WAIT_VMCNT(0);
  }
  __builtin_amdgcn_s_barrier();
  // stagger: group wr = 1 runs one barrier behind group wr = 0 for the whole K loop (LDS-read slots of one
  // group overlap MFMA slots of the other); group 0 pays the matching barrier after the loop.
  if (wr == 1) __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const char* base = smem + cur * BUF_BYTES;
    // ---- phase A: quadrants (0,0) and (0,1): 16 fragment reads, 32 MFMAs.  Restage: the last half-tile (A half 1)
    // of K-tile t+1 into the other buffer (its previous content was last read in phase B of K-tile t-1).
        TRIBE_LDS_B(base, 0, fb0)
    TRIBE_LDS_B(base, 1, fb1)
    TRIBE_LDS_A(base, 0)
    // retire A half 1 of THIS K-tile (read in phase B): behind it in the queue are the three half-tiles of
    // K-tile t+1 issued in the previous phase B and the one issued just now
    if (t + 1 < nk) { stage(3, cur ^ 1, t + 1); TRIBE_WAIT_VMCNT(8); } else { TRIBE_WAIT_VMCNT(0); }
    TRIBE_PHASE_SYNC_MMA2(0, 0, fb0, 0, 1, fb1)
    // ---- phase B: quadrants (1,1) and (1,0): 8 fragment reads, 32 MFMAs.  A half 0 and both B halves of THIS
    // buffer were last read in phase A -> restage them for K-tile t+2.
        TRIBE_LDS_A(base, 1)
    if (t + 2 < nk) {
      stage(0, cur, t + 2); stage(1, cur, t + 2); stage(2, cur, t + 2);
      TRIBE_WAIT_VMCNT(8);  // retire A0/B0/B1 of K-tile t+1; behind them: A1(t+1) and the three just issued
    } else if (t + 1 < nk) {
      TRIBE_WAIT_VMCNT(2);  // behind them: only A1(t+1)
    }
    TRIBE_PHASE_SYNC_MMA2(1, 1, fb1, 1, 0, fb0)
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef TRIBE_LDS_A
#undef TRIBE_LDS_B
#undef TRIBE_MMA
#undef TRIBE_PHASE_SYNC_MMA
#undef TRIBE_PHASE_SYNC_MMA2

  // ---- epilogue straight from registers (quad transpose -> one 16-/8-byte store per lane).  Staging the sub-tile
  // through LDS to get whole-row 256-byte stores was measured 2x SLOWER (K = 64 probe: 158 vs 75 us f32, 138 vs 43 us
  // bf16 per 16384 x 3072 output): the extra LDS round trip costs more than the wider store segments save.
  const EpiCtx ctx = make_epi_ctx(g, b1, b0, b1g);
  if (epilogue_fast_ok<EXT>(g, ctx) && n0 + BN <= g.N) {
    epilogue_fast<OUT_BF16, 8, 4, EXT>(g, ctx, acc, m0 + wr * 128, n0 + wc * 64, lane, smem + wave * 16384);
    return;
  }
  static_for<32>([&](auto t) {
    constexpr int i = decltype(t)::value / 4, j = decltype(t)::value % 4;
    epilogue_tile16<OUT_BF16, EXT>(g, ctx, acc[i][j], m0 + wr * 128 + i * 16, n0 + wc * 64 + j * 16, lane);
  });
}


// ---- quantisation: out[m, k] = e4m3(clamp(x[m, k] * inv_scale, +-448)), columns K..K_pad zero ------------------------
// v_cvt_pk_fp8_f32 rounds to nearest even into OCP e4m3fn on gfx950; the clamp makes it saturating (448 = max finite).
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ x, int64_t M, int64_t K, int64_t ld, float inv_scale,
                                                           unsigned char* __restrict__ out, int64_t K_pad) {
  const int64_t per_row = K_pad / 4;
  const int64_t total = M * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / per_row, k = (idx - m * per_row) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f = 0.f;
      if (k + j < K) {
        if (sizeof(T) == 2) f = bf16_to_f32(((const unsigned short*)x)[m * ld + k + j]);
        else f = ((const float*)x)[m * ld + k + j];
      }
      f *= inv_scale;
      v[j] = fminf(fmaxf(f, -448.f), 448.f);
    }
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
    *(int*)(out + m * K_pad + k) = pk;
  }
}

// ---- amax over a tensor (calibration of per-tensor scales): atomic max on the bit pattern of |x| ---------------------
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ x, int64_t M, int64_t K, int64_t ld, unsigned int* __restrict__ out) {
  float best = 0.f;
  const int64_t total = M * K;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / K, k = idx - m * K;
    const float f = sizeof(T) == 2 ? bf16_to_f32(((const unsigned short*)x)[m * ld + k]) : ((const float*)x)[m * ld + k];
    best = fmaxf(best, fabsf(f));
  }
  best = wave_max(best);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(best));   // non-negative floats order like their bit patterns
}

}  // namespace

extern "C" int tribe_gemm_fp8(const tribe_gemm_desc* d, void* stream) {
  TRIBE_REQUIRE(d != nullptr, "tribe_gemm_fp8: null descriptor");
  TRIBE_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "tribe_gemm_fp8: M, N, K must be positive (got %lld %lld %lld)", (long long)d->M,
                (long long)d->N, (long long)d->K);
  TRIBE_REQUIRE(d->K % 128 == 0, "tribe_gemm_fp8: K=%lld must be a multiple of 128 (zero-pad K)", (long long)d->K);
  TRIBE_REQUIRE(d->batch1 > 0 && d->batch0 > 0, "tribe_gemm_fp8: batch counts must be positive");
  TRIBE_REQUIRE(d->A && d->B && d->C, "tribe_gemm_fp8: null operand");
  TRIBE_REQUIRE(d->lda % 16 == 0 && d->ldb % 16 == 0 && d->sA1 % 16 == 0 && d->sA0 % 16 == 0 && d->sB1 % 16 == 0 && d->sB0 % 16 == 0,
                "tribe_gemm_fp8: lda/ldb/batch strides must be multiples of 16 elements (16-byte rows)");
  TRIBE_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0, "tribe_gemm_fp8: A/B must be 16-byte aligned");
  TRIBE_REQUIRE(d->lda >= d->K && d->ldb >= d->K && d->ldc >= ((d->act == TRIBE_ACT_SWIGLU || d->act == TRIBE_ACT_GLU) ? d->N / 2 : d->N),
                "tribe_gemm_fp8: leading dimension too small");
  TRIBE_REQUIRE((d->act != TRIBE_ACT_SWIGLU && d->act != TRIBE_ACT_GLU) || (d->N % 2 == 0 && !d->res && !d->rowadd && !d->gadd && d->bias_mode != TRIBE_BIAS_ROW),
                "tribe_gemm_fp8: SWIGLU / GLU need an even N and no residual / row adds");
  TRIBE_REQUIRE(d->c_dtype == TRIBE_F32 || d->c_dtype == TRIBE_BF16, "tribe_gemm_fp8: c_dtype must be f32 or bf16");
  TRIBE_REQUIRE(d->bias_mode == TRIBE_BIAS_NONE || d->bias != nullptr, "tribe_gemm_fp8: bias_mode set without bias");
  TRIBE_REQUIRE(!d->rowadd || d->rowadd_period > 0, "tribe_gemm_fp8: rowadd needs a positive period");
  TRIBE_REQUIRE(!d->gadd || (d->gadd_index && d->gadd_div > 0), "tribe_gemm_fp8: gadd needs index and divisor");
  TRIBE_REQUIRE(!(d->gather_a || d->gather_bias || d->gather_b) || d->gather1, "tribe_gemm_fp8: gather flags set without gather1");
  TRIBE_REQUIRE(d->act != TRIBE_ACT_GELU_BWD && !d->aux, "tribe_gemm_fp8: the training-only epilogues (aux, GELU_BWD) are bf16-only");
  const int64_t nz = d->batch1 * d->batch0;
  const int64_t tiles_m = (d->M + big::BM - 1) / big::BM, tiles_n = (d->N + big::BN - 1) / big::BN;
  TRIBE_REQUIRE(tiles_m * tiles_n < (1ll << 31) && nz < 65536, "tribe_gemm_fp8: grid too large");
  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)nz, 1);
  hipStream_t s = (hipStream_t)stream;
#define TRIBE_FP8_LAUNCH(BF, EXT)                                                                                            \
  do {                                                                                                                       \
    static bool attr_done = false;                                                                                           \
    if (!attr_done) {                                                                                                        \
      (void)hipFuncSetAttribute((const void*)gemm_fp8_nt_256x256x128<BF, EXT>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                big::SMEM_BYTES);                                                                            \
      attr_done = true;                                                                                                      \
    }                                                                                                                        \
    hipLaunchKernelGGL((gemm_fp8_nt_256x256x128<BF, EXT>), grid, dim3(512, 1, 1), big::SMEM_BYTES, s, *d, (int)tiles_m, (int)tiles_n); \
  } while (0)
  const bool bf = d->c_dtype == TRIBE_BF16;
  const bool ext = d->act == TRIBE_ACT_SWIGLU || d->act == TRIBE_ACT_GLU || d->act == TRIBE_ACT_SILU;
  if (ext) { if (bf) TRIBE_FP8_LAUNCH(1, 1); else TRIBE_FP8_LAUNCH(0, 1); }
  else { if (bf) TRIBE_FP8_LAUNCH(1, 0); else TRIBE_FP8_LAUNCH(0, 0); }
#undef TRIBE_FP8_LAUNCH
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_quantize_fp8_fwd(const void* x, int32_t x_dtype, int64_t M, int64_t K, int64_t ld, float inv_scale, uint8_t* out,
                                      int64_t K_pad, void* stream) {
  TRIBE_REQUIRE(x && out, "tribe_quantize_fp8_fwd: null pointer");
  TRIBE_REQUIRE(M > 0 && K > 0 && ld >= K && K_pad >= K && K_pad % 16 == 0, "tribe_quantize_fp8_fwd: bad shape M=%lld K=%lld ld=%lld K_pad=%lld",
                (long long)M, (long long)K, (long long)ld, (long long)K_pad);
  TRIBE_REQUIRE(inv_scale > 0.f && ((uintptr_t)out % 4) == 0, "tribe_quantize_fp8_fwd: inv_scale must be positive, out 4-byte aligned");
  int64_t blocks = (M * (K_pad / 4) + 255) / 256;
  if (blocks > 65536 * 8) blocks = 65536 * 8;
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == TRIBE_F32)
    hipLaunchKernelGGL(quantize_fp8_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, M, K, ld, inv_scale, out, K_pad);
  else if (x_dtype == TRIBE_BF16)
    hipLaunchKernelGGL(quantize_fp8_kernel<unsigned short>, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)x, M, K, ld,
                       inv_scale, out, K_pad);
  else
    TRIBE_REQUIRE(false, "tribe_quantize_fp8_fwd: x_dtype must be f32 or bf16");
  TRIBE_LAUNCH_CHECK();
  return 0;
}

extern "C" int tribe_absmax_fwd(const void* x, int32_t x_dtype, int64_t M, int64_t K, int64_t ld, float* out, int32_t accumulate, void* stream) {
  TRIBE_REQUIRE(x && out, "tribe_absmax_fwd: null pointer");
  TRIBE_REQUIRE(M > 0 && K > 0 && ld >= K, "tribe_absmax_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), s);
    if (e != hipSuccess) { tribe_set_error("tribe_absmax_fwd: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  }
  int64_t blocks = (M * K + 256 * 16 - 1) / (256 * 16);
  if (blocks > 4096) blocks = 4096;
  if (x_dtype == TRIBE_F32)
    hipLaunchKernelGGL(absmax_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, M, K, ld, (unsigned int*)out);
  else if (x_dtype == TRIBE_BF16)
    hipLaunchKernelGGL(absmax_kernel<unsigned short>, dim3((unsigned)blocks), dim3(256), 0, s, (const unsigned short*)x, M, K, ld, (unsigned int*)out);
  else
    TRIBE_REQUIRE(false, "tribe_absmax_fwd: x_dtype must be f32 or bf16");
  TRIBE_LAUNCH_CHECK();
  return 0;
}
