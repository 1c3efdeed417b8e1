// bf16 MFMA GEMM for gfx950:  C = epi(alpha * A . B^T),  A [M,K], B [N,K], K contiguous.
//
// Two tile configurations share one operator epilogue (gemm_common.h):
//  * 256x256x64, 512 threads (8 waves as 2x4, 128x64 per wave, 8x4 v_mfma_f32_16x16x32_bf16
//    accumulators), 128 KiB LDS = 2 K-tile buffers.  Operands go HBM/L2 -> LDS with
//    global_load_lds_dwordx4 (no VGPR round trip) in "half-tiles" of 128 rows.  A half-tile is the set
//    of tile rows whose fragments are read in ONE phase, so it can be restaged right after that phase:
//    per K-tile there are two phases (two C quadrants = 32 MFMAs per wave each); phase B restages three
//    half-tiles of K-tile t+2, phase A the fourth, and every phase ends its read slot with a COUNTED
//    s_waitcnt vmcnt(8) -- four half-tiles (80 KiB per CU) stay in flight across the raw s_barriers,
//    each with two full phases of slack (the counted-vmcnt schedule of cdna_hip_programming.md
//    section 5, re-derived for 2 LDS buffers).  The two wave groups wr = 0 / 1 (= the two waves of each
//    SIMD) run one barrier apart, so one group's LDS-read slot overlaps the other group's MFMA slot.
//    In-kernel stamps (scripts/gemm_stamps.py): MFMA cluster 42 %, LDS reads 18 %, stage + vmcnt 21 %,
//    barriers 18 % of a wave's K-loop time; matrix pipe ~78 % busy inside the loop.
//  * 128x128x64, 256 threads (4 waves, 64x64 per wave), double buffered, one vmcnt(0)+barrier per
//    K-tile: used when M or N is too small to fill 256-wide tiles.
// The LDS image is lane-linear (a glds requirement), so the bank-conflict swizzle
// chunk ^= (row & 7) is applied to the per-lane SOURCE address and undone on the ds_read_b128
// fragment read (guide rule 21 / T2).  Roofline: MFMA (dense bf16) -- see DESIGN.md "Kernels".
#include "gemm_common.h"

namespace {

constexpr int BK = 64;
typedef __attribute__((ext_vector_type(4))) short s16x4_tn;   // operand of ds_read_b64_tr_b16
constexpr int TRIBE_ROLE_EXT = 100;  // kernel instantiation carrying the extended epilogue (see gemm_common.h)

// =============================================================================================
// 256 x 256 x 64, counted-vmcnt pipeline
// =============================================================================================
namespace big {
constexpr int BM = 256, BN = 256;
constexpr int A_BYTES = BM * BK * 2;          // 32 KiB
constexpr int BUF_BYTES = 2 * A_BYTES;        // A + B of one K-tile: 64 KiB
constexpr int SMEM_BYTES = 2 * BUF_BYTES;     // 128 KiB
}  // namespace big

#define TRIBE_WAIT_VMCNT_(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define TRIBE_WAIT_VMCNT(n) TRIBE_WAIT_VMCNT_(n)
// Loads a wave leaves in flight at the two steady-state waits.  Experiment (profiles/r02_h_gemm_inflight_experiment.txt): 4 / 2
// instead of 8 cost FF1 +13 % / +28 % and FF2 +30 % / +34 % -- the loop needs its full K-tile of lead; a THIRD A slot (160 KiB
// LDS, A half-tiles issued two K-tiles ahead) bought nothing (3.88 vs 3.89 ms), so the two-buffer ring stays.
#ifndef TRIBE_GEMM_VM_STEADY
#define TRIBE_GEMM_VM_STEADY 8
#endif

// Diagnostic build only (-DTRIBE_GEMM_STAMPS): s_memtime stamps around the slots of the K loop, summed per wave and
// written to a side buffer that no other code reads (its pointer rides in desc.gadd_index while desc.gadd == NULL).
// Never quote the run time of this build; read the SHARES.
#ifdef TRIBE_GEMM_STAMPS
#define TRIBE_STAMP(var)                                                                  \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#define TRIBE_STAMP_ACC(slot, t_from, t_to) stamp_acc[slot] += (t_to) - (t_from)
#else
#define TRIBE_STAMP(var) do { } while (0)
#define TRIBE_STAMP_ACC(slot, t_from, t_to) do { } while (0)
#endif

// TN = 1 (desc.trans_ab): the operands arrive TRANSPOSED -- At [K, M] and Bt [K, N] row-major, the layout of the two factors of a
// weight gradient dW = dY^T X as the forward left them -- and C[m][n] = sum_k At[k][m] Bt[k][n].  Same schedule, same LDS bytes; what
// changes is the LDS image (per half-tile [64 k][128 out] rows of 256 bytes, 16-byte chunks XOR-swizzled by (k & 7) << 1 on the source
// side) and the fragment reads (two ds_read_b64_tr_b16 per fragment instead of one ds_read_b128: the hardware transposes; A and B use
// the same k order inside an MFMA, so the sum is unchanged).  Replaces the explicit bf16 transposes of the wgrad operands.
//
// NB = tile columns / 64 (round 3): the tile is 256 x (64 NB), a wave owns 128 x (16 NB), NB accumulator columns of 16.  NB = 4 is
// the 256 x 256 kernel; NB = 3 (256 x 192) exists for grids whose 256-wide tiles quantise badly on 256 CUs -- the BASELINE config at
// B = 4 has M = 4096: QKV 576 tiles (2.25 rounds), out-proj / FF2 192 tiles (0.75 of the chip) become 768 / 256 / 256.  The B tile is
// staged as NB LDS-DMA loads per wave (rows in tile order), all read in phase A and restaged in phase B, so the counted waits leave
// 4 + NB loads in flight.
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Stream-K schedule of the transposed-operand (weight-gradient) form, desc.stream_k: the tiles of the last, PARTIAL round -- rem = tiles % W
// on W CUs -- are cut into W equal runs of q = ceil(rem nk / W) K-steps, one per CU; a run that crosses a tile boundary is two workgroups (its
// part of the first tile, its part of the next), so no workgroup loops.  blockIdx.x: [0, tiles_dp) whole tiles as ever; then W "first parts"
// (worker w = index); then the "second parts" in DESCENDING length (second_worker[]), so that the dispatcher hands the CU whose first part
// was shortest the longest second part -- every CU ends up with ~q K-steps.  A workgroup that holds a whole tile stores it; a partial one
// leaves its accumulators in the workspace (slot 2 w + part, 256 KiB, in register order: 1 KiB per store instruction) and
// streamk_reduce_kernel, launched behind it, sums the parts of every split tile IN ORDER (deterministic) and writes alpha * sum to C.
// (First version: f32 atomics into a zeroed C -- the chip sustains only ~0.65 TB/s of them, ~100 us per 256-KiB part when every CU adds at
// once; profiles/r03_z11_gemm_streamk.txt.)  dW of out-proj at B = 16: 144 tiles = 0.56 rounds -> 256 CUs busy.
struct SkSched {
  int tiles_dp;   // tiles (linear index < tiles_dp) computed whole; >= tiles_m * tiles_n: no stream-K
  int rem_units;  // K-steps of the stream-K region: (tiles - tiles_dp) * nk
  int q;          // K-steps per worker
  int workers;
  float* ws;      // 2 * workers slots of 256 x 256 f32
  unsigned short second_worker[256];
};
constexpr int SK_SLOT_FLOATS = 256 * 256;

template <int OUT_BF16, int ROLE, int TN = 0, int NB = 4>
__global__ __launch_bounds__(512, 2) void gemm_nt_256x256x64(const tribe_gemm_desc g, int tiles_m, int tiles_n, const SkSched sk) {
  static_assert(NB >= 2 && NB <= 4 && (!TN || NB == 4), "tile columns: 128, 192 or 256; transposed operands: 256 only");
  constexpr int BM = big::BM, BN = 64 * NB, A_BYTES = big::A_BYTES, BUF_BYTES = A_BYTES + BN * BK * 2;
  constexpr int WN = 16 * NB;                      // columns per wave
  constexpr int VM_STEADY = NB == 4 ? TRIBE_GEMM_VM_STEADY : 4 + NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int wr = wave >> 2, wc = wave & 3;

  // which tile, which K-steps of it [k_first, k_first + nk), and whether that is the whole tile
  int tile_lin = (int)blockIdx.x, k_first = 0, nk = (int)(g.K / BK);
  bool partial = false;
  int sk_slot = 0;
  if constexpr (TN) {
    if ((int)blockIdx.x >= sk.tiles_dp) {
      const int sidx = (int)blockIdx.x - sk.tiles_dp;
      const bool second = sidx >= sk.workers;
      const int w = second ? (int)sk.second_worker[sidx - sk.workers] : sidx;
      const int u0 = w * sk.q;
      if (u0 >= sk.rem_units) return;                                    // (ceil: the last workers may have nothing)
      const int run = min(sk.q, sk.rem_units - u0);
      const int t_loc = u0 / nk, x = u0 - t_loc * nk, first_len = min(run, nk - x);
      if (second && first_len >= run) return;                            // this worker's run stays inside one tile
      tile_lin = sk.tiles_dp + t_loc + (second ? 1 : 0);
      k_first = second ? 0 : x;
      const int len = second ? run - first_len : first_len;
      partial = len != nk;
      nk = len;
      sk_slot = 2 * w + (second ? 1 : 0);
    }
  }
  int tm, tn;
  tile_coords(tile_lin, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int64_t z = blockIdx.y;
  const int64_t b1 = z / g.batch0, b0 = z - b1 * g.batch0;
  const int64_t b1g = g.gather1 ? g.gather1[b1] : b1;
  const unsigned short* A = (const unsigned short*)g.A + (g.gather_a ? b1g : b1) * g.sA1 + b0 * g.sA0;
  const unsigned short* B = (const unsigned short*)g.B + (g.gather_b ? b1g : b1) * g.sB1 + b0 * g.sB0;

  // ---- staging: each LDS-DMA instruction of a wave moves a slab of 8 tile rows x 128 bytes (row r of an operand tile at r * 128) ----
  //   A half h = the 128 tile rows phase h reads: {wr' * 128 + h * 64 + 0..63, wr' = 0, 1}; slab j of this wave: row0 = j * 128 + h * 64 + wave * 8
  //   B: slab q (0 .. NB-1) of this wave: row0 = q * 64 + wave * 8 (TN: the two half images of the round-2 layout, see below)
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;  // swizzle on the SOURCE chunk (row & 7 == srow for every slab)
#ifdef TRIBE_ABL_SAME_TILE   // ablation build: every workgroup streams the operand panels of tile (0, 0) -> all L2 hits (results are wrong)
#define m0 ((int64_t)0)
#define n0 ((int64_t)0)
#endif
  const unsigned short* a_src[2][2];
  const unsigned short* b_src[NB];
  int a_lds[2][2], b_lds[NB];  // wave-uniform LDS byte offsets of the slabs inside a K-tile buffer
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (TN) {
        // half-image h of an operand: [64 k][128 out] = 16 pieces of 4 k-rows; piece p = 8 j + wave; lane l fills physical chunk l & 15
        // of k-row 4 p + (l >> 4) from the logical chunk the swizzle maps there.  Logical chunk c of A's half h holds tile rows
        // (c >> 3) * 128 + h * 64 + (c & 7) * 8 .. + 7 (the two wave groups wr), of B's half h tile columns (c >> 2) * 64 + h * 32 +
        // (c & 3) * 8 .. + 7 (the four wc): exactly the rows / columns phase h reads.
        const int piece = 8 * j + wave, krow = 4 * piece + (lane >> 4);
        const int lc = (lane & 15) ^ ((krow & 7) << 1);
        int64_t gm = m0 + (lc >> 3) * 128 + h * 64 + (lc & 7) * 8; gm = gm + 8 <= g.M ? gm : g.M - 8;   // clamp to the last whole chunk
        int64_t gn = n0 + (lc >> 2) * 64 + h * 32 + (lc & 3) * 8; gn = gn + 8 <= g.N ? gn : g.N - 8;
        a_src[h][j] = A + (int64_t)krow * g.lda + gm;
        b_src[(2 * h + j) % NB] = B + (int64_t)krow * g.ldb + gn;
        a_lds[h][j] = h * 16384 + piece * 1024;
        b_lds[(2 * h + j) % NB] = A_BYTES + h * 16384 + piece * 1024;
      } else {
        const int ra = j * 128 + h * 64 + wave * 8;
        int64_t gr = m0 + ra + srow; gr = gr < g.M ? gr : g.M - 1;  // clamp: edge rows re-read a valid row, stores are masked
        a_src[h][j] = A + gr * g.lda + schunk * 8;
        a_lds[h][j] = ra * 128;
      }
    }
  if (!TN) {
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int rb = q * 64 + wave * 8;
      int64_t gc = n0 + rb + srow; gc = gc < g.N ? gc : g.N - 1;
      b_src[q] = B + gc * g.ldb + schunk * 8;
      b_lds[q] = A_BYTES + rb * 128;
    }
  }

#ifdef TRIBE_ABL_SAME_TILE
#undef m0
#undef n0
#endif
  // which: 0 = A half 0, 1 = the B tile (NB loads), 3 = A half 1
  auto stage = [&](int which, int buf, int kt) {
#ifdef TRIBE_ABL_NO_STAGE   // ablation build (scripts/gemm_ablation.py): no LDS-DMA, the K loop computes on whatever LDS holds
    return;
#endif
    char* base = smem + buf * BUF_BYTES;
    const int64_t koff_a = TN ? (int64_t)kt * BK * g.lda : (int64_t)kt * BK;   // K runs along the rows of a transposed operand
    const int64_t koff_b = TN ? (int64_t)kt * BK * g.ldb : (int64_t)kt * BK;
    if (which == 0 || which == 3) {
      const int h = which == 3;
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[h][0] + koff_a), (lptr_t)(base + a_lds[h][0]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[h][1] + koff_a), (lptr_t)(base + a_lds[h][1]), 16, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < NB; ++q)
        __builtin_amdgcn_global_load_lds((gptr_t)(b_src[q] + koff_b), (lptr_t)(base + b_lds[q]), 16, 0, 0);
    }
  };

  f32x4_t acc[8][NB];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int coff0 = ((fq ^ (frow & 7)) << 4), coff1 = (((4 + fq) ^ (frow & 7)) << 4);
  const int a_rd = (wr * 128 + frow) * 128;            // + mh*8192 + i*2048 + coff
  // NT form (TACC): the MFMAs run with their operands SWAPPED and the B fragments are read with their row quads permuted (0, 2, 1, 3), so that
  // every 16 x 16 accumulator holds its sub-tile transposed with lanes l / l + 32 on adjacent column quads -- what epilogue_fast<.., TACC = 1>
  // stores without a quad transpose and, for bf16 outputs, 16 bytes per lane.  The transposed-operand (TN) form keeps the
  // round-2 orientation.
#ifdef TRIBE_GEMM_NO_TACC
  constexpr int TACC = 0;
#else
  constexpr int TACC = TN ? 0 : 1;
#endif
  const int prow = TACC ? ((frow & 3) | ((frow & 4) << 1) | ((frow & 8) >> 1)) : frow;
  const int bcoff0 = ((fq ^ (prow & 7)) << 4), bcoff1 = (((4 + fq) ^ (prow & 7)) << 4);
  const int b_rd = A_BYTES + (wc * WN + prow) * 128;   // + j*2048 + bcoff

  bf16x8_t fa[4][2], fb[NB][2];
#ifdef TRIBE_ABL_NO_LDSREAD
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      fa[i][k] = bf16x8_t{(short)(0x3c00 + lane), (short)0x3f80, (short)(0xbf00 + i), (short)0x3e00, (short)0xbe80, (short)0x3f00, (short)(0x3d00 + k), (short)0xbd00};
      if (i < NB) fb[i][k] = fa[i][k];
    }
#endif

  // TN fragment reads: lane i of a 16-lane group supplies k-row 4 fq + (i >> 2) (and + 16) of the k-step and 4 of the fragment's 16
  // tile rows; after the hardware transpose lane (frow, fq) holds k = {4 fq .. + 3, 16 + 4 fq .. + 3} of tile row frow, for A and B alike.
  // Fragment i of A's half image sits at logical chunks wr * 8 + 2 i (+ 1), fragment j of B's half image nh at wc * 4 + 2 j (+ 1).
  const int tq = frow >> 2, tp = frow & 3;
  const int t_row = 4 * fq + tq, t_sw = (t_row & 7) << 1;
  int ta_off[4], tb_off[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) ta_off[i] = t_row * 256 + (((wr * 8 + 2 * i + (tp >> 1)) ^ t_sw) << 4) + (tp & 1) * 8;
#pragma unroll
  for (int j = 0; j < 2; ++j) tb_off[j] = A_BYTES + t_row * 256 + (((wc * 4 + 2 * j + (tp >> 1)) ^ t_sw) << 4) + (tp & 1) * 8;
  auto tr_frag = [&](const char* p) -> bf16x8_t {   // k-rows r and r + 16 of one k-step
    const s16x4_tn lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_tn*)(p));
    const s16x4_tn hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_tn*)(p + 16 * 256));
    bf16x8_t f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    return f;
  };

#ifdef TRIBE_ABL_NO_LDSREAD   // ablation build: fragments stay what the prologue put in the registers (opaque to the optimiser)
#define TRIBE_LDS_A(base, MH) _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(fa[i][0]), "+v"(fa[i][1])); }
#define TRIBE_LDS_B(base) _Pragma("unroll") for (int j = 0; j < NB; ++j) { asm volatile("" : "+v"(fb[j][0]), "+v"(fb[j][1])); }
#else
#define TRIBE_LDS_A(base, MH)                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                              \
    if (TN) {                                                                                  \
      fa[i][0] = tr_frag((base) + (MH) * 16384 + ta_off[i]);                                   \
      fa[i][1] = tr_frag((base) + (MH) * 16384 + ta_off[i] + 32 * 256);                        \
    } else {                                                                                   \
      fa[i][0] = *(const bf16x8_t*)((base) + a_rd + (MH) * 8192 + i * 2048 + coff0);           \
      fa[i][1] = *(const bf16x8_t*)((base) + a_rd + (MH) * 8192 + i * 2048 + coff1);           \
    }                                                                                          \
  }
#define TRIBE_LDS_B(base)                                                                      \
  _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                             \
    if (TN) {                                                                                  \
      fb[j][0] = tr_frag((base) + (j >> 1) * 16384 + tb_off[j & 1]);                           \
      fb[j][1] = tr_frag((base) + (j >> 1) * 16384 + tb_off[j & 1] + 32 * 256);                \
    } else {                                                                                   \
      fb[j][0] = *(const bf16x8_t*)((base) + b_rd + j * 2048 + bcoff0);                        \
      fb[j][1] = *(const bf16x8_t*)((base) + b_rd + j * 2048 + bcoff1);                        \
    }                                                                                          \
  }
#endif
#ifdef TRIBE_ABL_NO_MFMA   // ablation build: the fragments are consumed by an empty asm instead of the matrix pipe
#define TRIBE_MMA(MH)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(fa[i][0]), "v"(fa[i][1])); } \
  _Pragma("unroll") for (int j = 0; j < NB; ++j) { asm volatile("" :: "v"(fb[j][0]), "v"(fb[j][1])); }
#else
#define TRIBE_MMA(MH)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
  _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                             \
    acc[(MH) * 4 + i][j] = TACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][0], fa[i][0], acc[(MH) * 4 + i][j], 0, 0, 0)   \
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[j][0], acc[(MH) * 4 + i][j], 0, 0, 0);  \
    acc[(MH) * 4 + i][j] = TACC ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][1], fa[i][1], acc[(MH) * 4 + i][j], 0, 0, 0)   \
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[j][1], acc[(MH) * 4 + i][j], 0, 0, 0);  \
  }
#endif
// Priority: ONE s_setprio 1 for the younger wave group (wr = 1, the arbitration loser on every slot) before the K loop instead of
// flips around every MFMA cluster (guide T5, static form); the clusters are pinned between their barriers by sched_barrier.  Same-box
// A/B (profiles/r03_b / r03_c_gemm_ablation.txt): FF1 +0.7 ... +5 %, FF2 +0.9 ... +3.6 % across two boxes, 8192^3 +-0.3 %.
#ifdef TRIBE_GEMM_PRIO_FLIPS
#define TRIBE_PRIO_UP() __builtin_amdgcn_s_setprio(1)
#define TRIBE_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#else
#define TRIBE_PRIO_UP() __builtin_amdgcn_sched_barrier(0)
#define TRIBE_PRIO_DOWN() __builtin_amdgcn_sched_barrier(0)
#endif
// The fragment reads are retired BEFORE the barrier: the two wave groups (wr = 0 / 1 = the two waves of every
// SIMD) run one barrier apart, so while one group sits in this wait the other group's MFMA cluster owns the
// matrix pipe, and at every barrier all LDS reads issued so far are complete (restaging is then hazard-free).
#define TRIBE_PHASE_SYNC_MMA(MH)                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);                       \
  TRIBE_STAMP(ts2);                                        \
  __builtin_amdgcn_s_barrier();                            \
  TRIBE_STAMP(ts3);                                        \
  TRIBE_PRIO_UP();                                         \
  TRIBE_MMA(MH)                                            \
  TRIBE_PRIO_DOWN();                                       \
  TRIBE_STAMP(ts4);                                        \
  __builtin_amdgcn_s_barrier();                            \
  TRIBE_STAMP(ts5);                                        \
  TRIBE_STAMP_ACC(1, ts1, ts2); TRIBE_STAMP_ACC(2, ts2, ts3); TRIBE_STAMP_ACC(3, ts3, ts4); TRIBE_STAMP_ACC(4, ts4, ts5);

  // K rotation: the workgroups that share an operand panel in an XCD's L2 (the 4 x 8 patch of tile_coords) walk K from different
  // starting K-tiles, one apart, and wrap around.  Walking in lockstep, all sharers of a line wait on the SAME fill from beyond L2;
  // one K-tile apart, the first brings the line in and the others hit (profiles/r03_b_gemm_ablation.txt: with every load an L2 hit
  // FF1 / FF2 run 5 - 12 / 9 - 17 % faster).  Changes only the order of the f32 accumulation.
  // Measured (profiles/r03_c_gemm_ablation.txt, same box, interleaved): FF1 +2.0 %, FF2 +1.3 %, 8192^3 +0.1 %.
#ifdef TRIBE_GEMM_NO_KROT
  const int krot = 0;
#else
  const int krot = ((tn & 7) + (tm & 3)) % nk;
#endif
  // (Also tried, profiles/r03_h_gemm_kperm.txt: a cyclic shift inside every window of 8 K-tiles, which spreads the first touches evenly
  // over the sharers instead of leaving them to the one that walks ahead -- FF1 / FF2 / QKV within +-0.7 % of the plain rotation.)
  auto ktile = [&](int t) { const int k = t + krot; return k_first + (k >= nk ? k - nk : k); };
#ifdef TRIBE_GEMM_STAMPS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;
  unsigned long long stamp_acc[5] = {0, 0, 0, 0, 0};  // 0 LDS reads, 1 stage + vmcnt wait, 2 barrier 1, 3 MFMA cluster, 4 barrier 2
#endif

  // ---- prologue: K-tile 0 completely, K-tile 1 minus its last half-tile ----
  stage(0, 0, ktile(0)); stage(1, 0, ktile(0)); stage(3, 0, ktile(0));
  if (nk > 1) {
    stage(0, 1, ktile(1)); stage(1, 1, ktile(1));
    wait_vmcnt<2 + NB>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  // stagger: group wr = 1 runs one barrier behind group wr = 0 for the whole K loop (LDS-read slots of one
  // group overlap MFMA slots of the other); group 0 pays the matching barrier after the loop.
  if (wr == 1) __builtin_amdgcn_s_barrier();
#if !defined(TRIBE_GEMM_PRIO_FLIPS) && !defined(TRIBE_ABL_NO_PRIO)
  if (wr == 1) __builtin_amdgcn_s_setprio(1);
#endif

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const char* base = smem + cur * BUF_BYTES;
    // ---- phase A: accumulator rows 0..3 x all NB columns: 2 NB + 8 fragment reads, 8 NB MFMAs.  Restage: the last half-tile
    // (A half 1) of K-tile t+1 into the other buffer (its previous content was last read in phase B of K-tile t-1).
    TRIBE_STAMP(ts0);
    TRIBE_LDS_B(base)
    TRIBE_LDS_A(base, 0)
    TRIBE_STAMP(ts1);
    TRIBE_STAMP_ACC(0, ts0, ts1);
    // retire A half 1 of THIS K-tile (read in phase B): behind it in the queue are A half 0 + B of K-tile t+1 issued in the
    // previous phase B (2 + NB loads) and the half-tile issued just now (2)
    if (t + 1 < nk) { stage(3, cur ^ 1, ktile(t + 1)); wait_vmcnt<VM_STEADY>(); } else { wait_vmcnt<0>(); }
    TRIBE_PHASE_SYNC_MMA(0)
    // ---- phase B: accumulator rows 4..7: 8 fragment reads (the B fragments stay in registers), 8 NB MFMAs.  A half 0 and the
    // B tile of THIS buffer were last read in phase A -> restage them for K-tile t+2.
    TRIBE_STAMP(ts0);
    TRIBE_LDS_A(base, 1)
    TRIBE_STAMP(ts1);
    TRIBE_STAMP_ACC(0, ts0, ts1);
    if (t + 2 < nk) {
      const int k2 = ktile(t + 2);
      stage(0, cur, k2); stage(1, cur, k2);
      wait_vmcnt<VM_STEADY>();  // retire A0 / B of K-tile t+1; behind them: A1(t+1) (2) and the 2 + NB just issued
    } else if (t + 1 < nk) {
      wait_vmcnt<2>();  // behind them: only A1(t+1)
    }
    TRIBE_PHASE_SYNC_MMA(1)
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_setprio(0);
#ifdef TRIBE_GEMM_STAMPS
  if (g.gadd == nullptr && g.gadd_index != nullptr && lane == 0 && blockIdx.y == 0) {
    unsigned long long* dbg = (unsigned long long*)g.gadd_index + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int i = 0; i < 5; ++i) dbg[i] = stamp_acc[i];
  }
#endif
#undef TRIBE_LDS_A
#undef TRIBE_LDS_B
#undef TRIBE_MMA
#undef TRIBE_PHASE_SYNC_MMA

  // ---- epilogue straight from registers (quad transpose -> one 16-/8-byte store per lane).  Staging the sub-tile
  // through LDS to get whole-row 256-byte stores was measured 2x SLOWER (K = 64 probe: 158 vs 75 us f32, 138 vs 43 us
  // bf16 per 16384 x 3072 output): the extra LDS round trip costs more than the wider store segments save.
  const EpiCtx ctx = make_epi_ctx(g, b1, b0, b1g);
  if constexpr (TN && !OUT_BF16) {
    if (partial) {
      // part of a tile's reduction (stream-K): the raw accumulators go to this part's workspace slot in register order
      float* slot = sk.ws + (int64_t)sk_slot * SK_SLOT_FLOATS + tid * 4;
      static_for<8 * NB>([&](auto t) {
        constexpr int i = decltype(t)::value / NB, j = decltype(t)::value % NB;
        *(f32x4_t*)(slot + (i * NB + j) * 2048) = acc[i][j];
      });
      return;
    }
  }
  if (epilogue_fast_ok<(ROLE == TRIBE_ROLE_EXT)>(g, ctx) && n0 + BN <= g.N) {
    // nothing inside the sub-tile loop waits on memory (gemm_common.h); the staging buffers are idle by now
    epilogue_fast<OUT_BF16, 8, NB, (ROLE == TRIBE_ROLE_EXT), TACC>(g, ctx, acc, m0 + wr * 128, n0 + wc * WN, lane, smem + wave * (4 * NB * 1024));
    return;
  }
  static_for<8 * NB>([&](auto t) {
    constexpr int i = decltype(t)::value / NB, j = decltype(t)::value % NB;
    epilogue_tile16<OUT_BF16, (ROLE == TRIBE_ROLE_EXT), TACC>(g, ctx, acc[i][j], m0 + wr * 128 + i * 16, n0 + wc * WN + j * 16, lane);
  });
}

// Second launch of a stream-K GEMM: workgroup t sums the parts of tile tiles_dp + t in worker order (first parts and second parts of the runs
// that touch it) and writes alpha * sum.  Thread tid holds the same 128 values the GEMM kernel's thread tid held: accumulator register r of
// sub-tile (i, j) of lane l is D[4 (l >> 4) + r][l & 15] of the 16 x 16 sub-tile at rows wr 128 + 16 i, columns wc 64 + 16 j.
__global__ __launch_bounds__(512) void streamk_reduce_kernel(const tribe_gemm_desc g, int tiles_m, int tiles_n, const SkSched sk) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
  const int nk = (int)(g.K / BK);
  const int t_loc = (int)blockIdx.x;
  const int u_lo = t_loc * nk, u_hi = u_lo + nk - 1;
  const int w_lo = u_lo / sk.q, w_hi = u_hi / sk.q;
  if (w_lo == w_hi) return;   // one run holds the whole tile: stored by the GEMM kernel itself
  int tm, tn;
  tile_coords(sk.tiles_dp + t_loc, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
  float* C = (float*)g.C;
  for (int sub = 0; sub < 32; ++sub) {
    f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
    for (int w = w_lo; w <= w_hi; ++w) {
      const int part = w * sk.q < u_lo ? 1 : 0;   // the run began in the previous tile: this tile holds its second part
      sum += *(const f32x4_t*)(sk.ws + (int64_t)(2 * w + part) * SK_SLOT_FLOATS + sub * 2048 + tid * 4);
    }
    const int i = sub >> 2, j = sub & 3;
    const int64_t col = n0 + wc * 64 + j * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = m0 + wr * 128 + i * 16 + 4 * (lane >> 4) + r;
      if (row < g.M && col < g.N) C[row * g.ldc + col] = sum[r] * g.alpha;
    }
  }
}

// =============================================================================================
// 128 x 128 x 64, double buffered
// =============================================================================================
namespace small {
constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = BM * BK * 2;       // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;     // A + B
constexpr int SMEM_BYTES = 2 * BUF_BYTES;     // double buffered: 64 KiB
}  // namespace small

// ROLE only gives each call site of the path its own kernel symbol, so that rocprofv3 --stats and the
// in-library HIP-event profile (tribe_prof_*) report per-operator durations; the code is identical.
template <int OUT_BF16, int ROLE>
__global__ __launch_bounds__(256, 2) void gemm_nt_128x128x64(const tribe_gemm_desc g, int tiles_m, int tiles_n) {
  using namespace small;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  int tm, tn;
  tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int64_t z = blockIdx.y;
  const int64_t b1 = z / g.batch0, b0 = z - b1 * g.batch0;
  const int64_t b1g = g.gather1 ? g.gather1[b1] : b1;
  const unsigned short* A = (const unsigned short*)g.A + (g.gather_a ? b1g : b1) * g.sA1 + b0 * g.sA0;
  const unsigned short* B = (const unsigned short*)g.B + (g.gather_b ? b1g : b1) * g.sB1 + b0 * g.sB0;

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

  const EpiCtx ctx = make_epi_ctx(g, b1, b0, b1g);
  if (epilogue_fast_ok<(ROLE == TRIBE_ROLE_EXT)>(g, ctx) && n0 + BN <= g.N) {
    // (the K loop ends with a barrier: every wave's fragment reads are done, the buffers can stage the residual)
    epilogue_fast<OUT_BF16, 4, 4, (ROLE == TRIBE_ROLE_EXT)>(g, ctx, acc, m0 + wr * 64, n0 + wc * 64, lane, smem + wave * 16384);
    return;
  }
  static_for<16>([&](auto t) {
    constexpr int i = decltype(t)::value / 4, j = decltype(t)::value % 4;
    epilogue_tile16<OUT_BF16, (ROLE == TRIBE_ROLE_EXT)>(g, ctx, acc[i][j], m0 + wr * 64 + i * 16, n0 + wc * 64 + j * 16, lane);
  });
}


// =============================================================================================
// 128 x 128 x 64 ring (round 3): small grids -- at most ~one workgroup per CU
// =============================================================================================
// The double-buffered 128^2 kernel above waits vmcnt(0) + barrier once per K-tile with one K-tile in flight.  That is fine with two
// or three workgroups per CU covering for each other, and slow when the grid gives a CU ONE workgroup (BASELINE config at B = 4:
// projector and voxel head are 256 tiles of 128^2 -- 0.18 / 0.16 of the MFMA peak in BENCH_r02): every K-step then pays a whole
// L2 -> LDS round trip.  Here: 8 waves (two per SIMD; a wave owns 32 x 64), four 32-KiB stages = 128 KiB of LDS, THREE K-tiles in
// flight by LDS-DMA across ONE raw barrier per K-tile with counted waits:
//   iteration t:  s_waitcnt vmcnt(8)  (own loads of K-tile t landed; t+1, t+2 stay in flight)
//                 s_barrier           (everybody's have; and everybody finished reading K-tile t-1: its reads were retired before
//                                      the MFMAs of iteration t-1)
//                 issue K-tile t+3 into the stage K-tile t-1 occupied
//                 12 ds_read_b128, 16 MFMAs
// Per K-step a CU moves 32 KiB from L2 for 512 MFMA cycles, so the stream (~0.5-0.6 us per 32 KiB per CU) bounds it, not the
// matrix pipe; 96 KiB in flight keep that stream busy.
namespace ring {
constexpr int BM = 128, BN = 128, STAGES = 4;
constexpr int TILE_BYTES = BM * BK * 2;           // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // A + B
constexpr int SMEM_BYTES = STAGES * STAGE_BYTES;  // 128 KiB
}  // namespace ring

// Split-K (round 3, desc.stream_k on an NT launch): grids far below one round of tiles -- M = 128 rows, BASELINE config 1: FF2 is 24 tiles of
// 192 K-steps -- run `splits` workgroups per tile, each over a contiguous share of the K-steps; they store their raw accumulators to
// ws[split][M][N] (f32) and splitk_epilogue_kernel, launched behind, sums the shares in order and applies the launch's epilogue operators.
template <int OUT_BF16, int ROLE>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring128(const tribe_gemm_desc g, int tiles_m, int tiles_n, int splits, float* ws) {
  using namespace ring;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int wr = wave >> 1, wc = wave & 1;                    // 4 x 2 waves of 32 x 64

  const int ntile = tiles_m * tiles_n;
  const int split = splits > 1 ? (int)blockIdx.x / ntile : 0;
  int tm, tn;
  tile_coords((int)blockIdx.x - split * ntile, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int64_t z = blockIdx.y;
  const int64_t b1 = z / g.batch0, b0 = z - b1 * g.batch0;
  const int64_t b1g = g.gather1 ? g.gather1[b1] : b1;
  const unsigned short* A = (const unsigned short*)g.A + (g.gather_a ? b1g : b1) * g.sA1 + b0 * g.sA0;
  const unsigned short* B = (const unsigned short*)g.B + (g.gather_b ? b1g : b1) * g.sB1 + b0 * g.sB0;

  // staging: a tile is 16 slabs of 8 rows x 128 bytes (row r at r * 128, 16-byte chunks XOR-swizzled by r & 7 on the SOURCE side);
  // this wave moves slabs `wave` and `wave + 8` of both operands: 4 LDS-DMA loads per K-tile
  const int srow = lane >> 3;
  const int schunk = (lane & 7) ^ srow;
  const unsigned short* a_src[2];
  const unsigned short* b_src[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = (wave + 8 * p) * 8 + srow;
    int64_t gr = m0 + r; gr = gr < g.M ? gr : g.M - 1;   // clamp: edge rows re-read a valid row, stores are masked
    int64_t gc = n0 + r; gc = gc < g.N ? gc : g.N - 1;
    a_src[p] = A + gr * g.lda + schunk * 8;
    b_src[p] = B + gc * g.ldb + schunk * 8;
  }
  const int nk_all = (int)(g.K / BK);
  const int k_first = splits > 1 ? (int)((int64_t)split * nk_all / splits) : 0;
  const int nk = splits > 1 ? (int)((int64_t)(split + 1) * nk_all / splits) - k_first : nk_all;
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES + wave * 1024;
    const int koff = (k_first + kt) * BK;
    __builtin_amdgcn_global_load_lds((gptr_t)(a_src[0] + koff), (lptr_t)(base), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(b_src[0] + koff), (lptr_t)(base + TILE_BYTES), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(a_src[1] + koff), (lptr_t)(base + 8192), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(b_src[1] + koff), (lptr_t)(base + TILE_BYTES + 8192), 16, 0, 0);
  };

  f32x4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int coff0 = ((fq ^ (frow & 7)) << 4), coff1 = (((4 + fq) ^ (frow & 7)) << 4);
  const int a_rd = (wr * 32 + frow) * 128;                 // + i * 2048 + coff
  const int b_rd = TILE_BYTES + (wc * 64 + frow) * 128;    // + j * 2048 + coff

  if (0 < nk) stage(0, 0);
  if (1 < nk) stage(1, 1);
  if (2 < nk) stage(2, 2);
  bf16x8_t fa[2][2], fb[4][2];
  auto sync_and_restage = [&](int t) {
    const int ahead = nk - 1 - t;   // K-tiles issued behind K-tile t (capped at 2 by the ring)
    if (ahead >= 2) wait_vmcnt<8>(); else if (ahead == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (t + 3 < nk) stage((t + 3) & 3, t + 3);
  };
  auto read_frags = [&](int t) {
    const char* base = smem + (t & 3) * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fb[j][0] = *(const bf16x8_t*)(base + b_rd + j * 2048 + coff0);
      fb[j][1] = *(const bf16x8_t*)(base + b_rd + j * 2048 + coff1);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[i][0] = *(const bf16x8_t*)(base + a_rd + i * 2048 + coff0);
      fa[i][1] = *(const bf16x8_t*)(base + a_rd + i * 2048 + coff1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read retired before the next barrier can be reached (restaging is then safe)
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  // Tried and left off (-DTRIBE_RING_STAGGER): the two waves of a SIMD (w and w + 4) run the same program, so waves 4..7 were made to
  // run their MFMAs half an iteration late (multiply K-tile t-1 while waves 0..3 read K-tile t; guide: "two waves per SIMD that run the
  // same program with one barrier per block: try a stagger").  Same-box A/B (profiles/r03_f_gemm_tiles*.txt): projector 46.2 vs 44.1 us
  // without, voxel head 36.7 vs 36.2 -- this loop waits on the L2 -> LDS stream, not on the matrix pipe, so there is nothing to overlap.
#ifdef TRIBE_RING_STAGGER
  if (wave >= 4) {
    for (int t = 0; t < nk; ++t) {
      sync_and_restage(t);
      if (t > 0) mma();
      read_frags(t);
    }
    mma();
  } else
#endif
  {
    for (int t = 0; t < nk; ++t) {
      sync_and_restage(t);
      read_frags(t);
      mma();
    }
  }
  __builtin_amdgcn_s_barrier();   // the staging buffers become the epilogue's scratch: every wave is past its last fragment read

  if (splits > 1) {   // this workgroup's share of the reduction, raw: [split][M][N] f32, four consecutive columns of a row per lane
    float* wsp = ws + (int64_t)split * g.M * g.N;
    static_for<8>([&](auto t) {
      constexpr int i = decltype(t)::value / 4, j = decltype(t)::value % 4;
      float v[4];
      quad_transpose(acc[i][j], 1.0f, lane, v);
      const int64_t m = m0 + wr * 32 + i * 16 + ((lane >> 4) << 2) + (lane & 3), n = n0 + wc * 64 + j * 16 + (((lane & 15) >> 2) << 2);
      if (m < g.M && n < g.N) *(float4*)(wsp + m * g.N + n) = make_float4(v[0], v[1], v[2], v[3]);
    });
    return;
  }
  const EpiCtx ctx = make_epi_ctx(g, b1, b0, b1g);
  if (epilogue_fast_ok<(ROLE == TRIBE_ROLE_EXT)>(g, ctx) && n0 + BN <= g.N) {
    epilogue_fast<OUT_BF16, 2, 4, (ROLE == TRIBE_ROLE_EXT)>(g, ctx, acc, m0 + wr * 32, n0 + wc * 64, lane, smem + wave * 8192);
    return;
  }
  static_for<8>([&](auto t) {
    constexpr int i = decltype(t)::value / 4, j = decltype(t)::value % 4;
    epilogue_tile16<OUT_BF16, (ROLE == TRIBE_ROLE_EXT)>(g, ctx, acc[i][j], m0 + wr * 32 + i * 16, n0 + wc * 64 + j * 16, lane);
  });
}


// Second launch of a split-K GEMM: thread = four consecutive columns of one row; sums the shares in split order (deterministic) and applies
// the operators of the wait-free epilogue in its order: alpha, row_scale, row / column bias, GELU (the bf16 / f32 forms of the GEMM epilogues),
// scaled residual or periodic row add, store, bf16 copy, row sums of squares (slot n / 64: 16 consecutive lanes hold 64 columns of a row).
template <int OUT_BF16>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const tribe_gemm_desc g, int splits, const float* __restrict__ ws) {
  const int64_t n4 = g.N >> 2, total = g.M * n4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < total;
  const int64_t id = live ? idx : total - 1;
  const int64_t m = id / n4, n = (id - m * n4) << 2;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sp = 0; sp < splits; ++sp) {
    const float4 p = *(const float4*)(ws + ((int64_t)sp * g.M + m) * g.N + n);
    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
  }
  float v[4] = {acc.x * g.alpha, acc.y * g.alpha, acc.z * g.alpha, acc.w * g.alpha};
  if (g.row_scale) {
    const float sc = g.row_scale[m];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] *= sc;
  }
  if (g.bias_mode == TRIBE_BIAS_ROW) {
    const float b = g.bias[m];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += b;
  } else if (g.bias_mode == TRIBE_BIAS_COL) {
    const float4 b = *(const float4*)(g.bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (g.act == TRIBE_ACT_GELU) {
    if (OUT_BF16) {
      const f32x2_t lo = gelu_poly2(f32x2_t{v[0], v[1]}), hi = gelu_poly2(f32x2_t{v[2], v[3]});
      v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = gelu_erf(v[k]);
    }
  }
  if (g.res) {
    const float4 r = *(const float4*)(g.res + m * g.ldres + n);
    if (g.res_scale) {
      const float4 sc = *(const float4*)(g.res_scale + n);
      v[0] += r.x * sc.x; v[1] += r.y * sc.y; v[2] += r.z * sc.z; v[3] += r.w * sc.w;
    } else {
      v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
  } else if (g.rowadd) {
    const float4 r = *(const float4*)(g.rowadd + (m % g.rowadd_period) * g.ld_rowadd + n);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  u16x4_t o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = f32_to_bf16(v[k]);
  if (OUT_BF16) {
    if (live) *(u16x4_t*)((unsigned short*)g.C + m * g.ldc + n) = o;
  } else {
    if (live) *(float4*)((float*)g.C + m * g.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    if (g.c_bf16 && live) *(u16x4_t*)(g.c_bf16 + m * g.ld_c_bf16 + n) = o;
    if (g.row_sumsq) {   // (the planner guarantees N % 64 == 0: a 16-lane group never straddles two rows)
      float t = live ? v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3] : 0.f;
      t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64);
      if (live && (threadIdx.x & 15) == 0) g.row_sumsq[m * g.ld_row_sumsq + (n >> 6)] = t;
    }
  }
}

// =============================================================================================
// 256 x 256 x 64, ONE wave per SIMD (round 3): 4 waves, 128 x 128 per wave, accumulators in 256 AGPRs
// =============================================================================================
// VERDICT r2 item 1.  The 8-wave kernel above spends 58 % of a wave's loop outside its MFMA cluster and relies on the SIMD partner to
// cover it; here a wave owns a whole SIMD and interleaves everything itself:
//   * 64 accumulators of 16 x 16 (v_mfma_f32_16x16x32_bf16; the 32 x 32 x 16 shape holds a lower clock on this part, guide DVFS item 7)
//     in compiler-allocated AGPRs ("+a" operands of inline-asm MFMAs: nothing in the loop does arithmetic on them, so hipcc never moves
//     them), two fragment sets of 64 VGPRs (k-step 0 / 1 of a K-tile);
//   * the K loop is a sequence of volatile asm statements in SOURCE ORDER -- hipcc's own schedule of the same loop ran 1.78 us per K-step
//     against 1.36 for this one (scripts/probes/gemm_1wave_probe.hip, profiles/r03_*_gemm_1wave_probe.txt):
//       phase B(t):  MFMAs (t, k-step 0)  ||  ds_reads (t, k-step 1), one per 2 MFMAs  ||  from MFMA 40 on, behind "lgkmcnt(0) + barrier"
//                    (every wave has finished reading K-tile t), LDS-DMA pieces of K-tile t+2 into the buffer K-tile t occupied
//       phase A(t+1): MFMAs (t, k-step 1) ||  more pieces  ||  at MFMA 16 "vmcnt(10) + barrier" (K-tile t+1 landed; the 10 youngest loads
//                    are K-tile t+2's) and then the ds_reads (t+1, k-step 0)
//     i.e. one LDS-DMA piece per 4 MFMAs, spread over 64 MFMAs; its last piece has 104 MFMAs of flight before its wait.  LDS reads per CU
//     and K-tile fall from 192 KiB to 128 KiB; each wave reads 32 fragments for 128 MFMAs.
//   * an LDS-DMA piece costs a lone wave ~35 cycles of MFMA issue (no partner wave to hide it), so the pieces are made as cheap as the ISA
//     allows: global_load_lds with a SCALAR base + fixed 32-bit per-lane offsets (no vector arithmetic per piece), four pieces per M0
//     write (the immediate offset steps the LDS and the memory address alike; the per-lane offsets compensate), no M0 save / restore.
// Probe at 8192^3, same box, naive epilogue: 1574 vs 1505 TFLOP/s for the 8-wave kernel (+4.6 %); 2363 cycles per K-tile against 2048
// of pure MFMA issue (the 8-wave kernel: ~2800), at a clock the chip lowers from 1.95 to 1.87 GHz as the stream gets denser.
// Tried and dropped (profiles/r03_w_4w_lab_persistent.txt): a PERSISTENT form (one workgroup per CU walking the tiles, the next tile's first two
// K-tiles requested before the current tile's epilogue stores): QKV +2.5 % instead of +4.1 % over the 8-wave kernel, FF1 +1.0 % instead of
// +3.2 % -- the dispatcher already starts the next workgroup while the previous one's stores drain, and a static tile walk gives up its
// load balancing over 36 - 48 tiles per CU.
// NT form, K % 64 == 0, N % 256 == 0, the operator sets of the four encoder GEMMs (epilogue_w4); everything else keeps the 8-wave kernel.
#ifdef TRIBE_GEMM_STAMPS4W
__device__ unsigned long long g_dbg_4w[8];
#endif
namespace w4 {
constexpr int SMEM_BYTES = 2 * 65536;   // two K-tile buffers: A 32 KiB + B 32 KiB each
constexpr int WB = 40, RB = 16, DPM = 4;                      // buffer-free barrier at MFMA WB of phase B, landed barrier at MFMA RB of phase A
constexpr int NB_PIECES = (64 - WB) / DPM, NA1 = RB / DPM;    // pieces of K-tile t+2 issued before the landed barrier: 6 + 4
static_assert((64 - WB) % DPM == 0 && RB % DPM == 0 && NB_PIECES + NA1 <= 16 && (16 - NB_PIECES) * DPM <= 64, "piece schedule");

__device__ __forceinline__ void mfma(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void lds_rd(bf16x8_t& f, unsigned addr) {   // destination valid after the caller's own lgkmcnt wait (guide 5.7 item 1, form iii)
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
template <int IMM>
__device__ __forceinline__ void glds(unsigned voff, const void* sbase) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
__device__ __forceinline__ void set_m0(unsigned v) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(v) : "memory"); }
}  // namespace w4

// The epilogue of the one-wave-per-SIMD kernel, specialised at COMPILE time by the operator set of the four encoder GEMMs it serves
// (QKV: [row scale] -> bf16; FF1: [row scale] + column bias + GELU -> bf16; out-proj / FF2: [column bias] + scaled f32 residual in place
// [+ bf16 copy + row sums of squares]).  epilogue_fast decides every operator at run time inside every sub-tile; with 64 sub-tiles per
// wave that is ~70 000 instructions of mostly skipped code per kernel -- the instruction cache thrashed and a tile's epilogue took
// 37 000 (bf16) to 72 000 (f32) cycles next to a 120 000-cycle K loop (stamps, profiles/r03_q_4w_lab.txt).  Here a sub-tile is ~60
// instructions.  The launcher (w4_role_ok) only sends descriptors whose operators match the role; everything else keeps the 8-wave kernel.
template <int OUT_BF16, int ROLE>
__device__ __forceinline__ void epilogue_w4(const tribe_gemm_desc& g, f32x4_t (&acc)[8][8], int64_t mw, int64_t nw, int lane, char* lds_wave) {
  constexpr bool BIAS = ROLE == TRIBE_ROLE_FF1 || ROLE == TRIBE_ROLE_FF2;
  constexpr bool GELU = ROLE == TRIBE_ROLE_FF1;
  constexpr bool RES = ROLE == TRIBE_ROLE_OUT_PROJ || ROLE == TRIBE_ROLE_FF2;
  static_assert(OUT_BF16 == (RES ? 0 : 1), "QKV / FF1 write bf16, out-proj / FF2 the f32 residual stream");
  // The K loop multiplies with the operands SWAPPED (B fragment as the MFMA's A operand), so a 16 x 16 accumulator holds the sub-tile
  // transposed: register r of lane l is C[row = l & 15][column = 4 (l >> 4) + r] -- four CONSECUTIVE columns of one row per lane, the
  // shape of a 16-byte / 8-byte store, with no quad transpose (16 vector instructions per sub-tile in the 8-wave kernel's epilogue; a
  // lone wave per SIMD issues one vector instruction per ~4.5 cycles, so with 64 sub-tiles they would be 5000 cycles per tile).
  // The B fragments are read with their row quads permuted (sigma = 0, 2, 1, 3: see the kernel), so lane group q = l >> 4 holds columns
  // 4 sigma(q) .. + 3: lanes l and l + 32 own adjacent quads of a row, and for a PAIR of sub-tiles (j, j + 1) two v_permlane32_swap per
  // pair hand the lower half-wave 8 consecutive bf16 columns of sub-tile j and the upper half those of j + 1: ONE 16-byte store per lane
  // and pair instead of two 8-byte ones (guide T21: an epilogue's store tail is bound by store INSTRUCTIONS, not bytes).
  const int q = lane >> 4;
  const int64_t row0 = mw + (lane & 15);                                   // + 16 i
  const int64_t col0 = nw + 4 * ((q & 1) * 2 + (q >> 1));                  // + 16 j   (4 sigma(q))
  const int64_t colp = nw + 16 * (lane >> 5) + 8 * (q & 1);                // + 16 j (j even): first of the 8 columns the paired store writes
  const bool has_rs = !RES && g.row_scale != nullptr, has_rsc = RES && g.res_scale != nullptr;
  const bool has_cb = RES && g.c_bf16 != nullptr, has_ssq = RES && g.row_sumsq != nullptr;
  const bool interior = mw + 128 <= g.M;   // wave-uniform
  float4 bcol[8], rsc[8];
  static_for<8>([&](auto jt) {
    constexpr int j = decltype(jt)::value;
    if constexpr (BIAS) bcol[j] = *(const float4*)(g.bias + col0 + j * 16);
    if constexpr (RES) rsc[j] = has_rsc ? *(const float4*)(g.res_scale + col0 + j * 16) : make_float4(1.f, 1.f, 1.f, 1.f);
  });
  // The residual tile comes in by LDS-DMA, 16 sub-tiles (16 KiB per wave) per round, DOUBLE-BUFFERED: round r+1 is requested before round r
  // is processed, and the wait for a round is COUNTED -- vmcnt counts loads and stores in issue order, so a full drain before every round
  // would also wait for the previous round's 32 stores (that serialisation was 70 000 cycles per tile, stamps of profiles/r03_s_4w_lab.txt).
  auto res_dma = [&](auto rc) {
    constexpr int r = decltype(rc)::value;
    static_for<2>([&](auto it) {
      constexpr int i2 = decltype(it)::value;
      int64_t row = row0 + (r * 2 + i2) * 16;
      row = row < g.M ? row : g.M - 1;   // rows past M re-read a valid row (never stored): ALL 16 requests of a round are always issued,
      const float* src = g.res + row * g.ldres + col0;   // the counted waits below depend on it
      static_for<8>([&](auto jt) {
        constexpr int j = decltype(jt)::value;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 16), (lptr_t)(lds_wave + (r & 1) * 16384 + (i2 * 8 + j) * 1024), 16, 0, 0);
      });
    });
  };
  if constexpr (RES) {
    res_dma(std::integral_constant<int, 0>{});
    res_dma(std::integral_constant<int, 1>{});
  }
  static_for<4>([&](auto rt) {   // rounds of 2 sub-tile rows x 8 columns = 16 sub-tiles
    constexpr int round = decltype(rt)::value;
    float srow[2], ssq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    static_for<2>([&](auto it) {
      const int64_t r = row0 + (round * 2 + decltype(it)::value) * 16;
      srow[decltype(it)::value] = (has_rs && r < g.M) ? g.row_scale[r] : 1.0f;
    });
    if constexpr (RES) {
      // A wait "vmcnt(N)" retires this round's requests iff at least N operations were issued after them.  Interior tile (every row of
      // the wave < M: every sub-tile stores): younger = [round >= 1: the >= 16 stores of round - 1] + [round <= 2: the 16 requests of
      // round + 1].  A tile that crosses M may skip whole stores, so it drains.
      if (interior) {
        if constexpr (round == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr (round < 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    uint2 pend = make_uint2(0u, 0u);
    static_for<16>([&](auto st) {
      constexpr int sidx = decltype(st)::value, i2 = sidx / 8, i = round * 2 + i2, j = sidx % 8;
      const int64_t row = row0 + i * 16;
      if (OUT_BF16 == 0 && row >= g.M) return;   // (bf16 roles: every lane takes part in the half-wave exchange; the store is masked below)
      float v[4] = {acc[i][j][0] * g.alpha, acc[i][j][1] * g.alpha, acc[i][j][2] * g.alpha, acc[i][j][3] * g.alpha};
      if constexpr (!RES) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= srow[i2];
      }
      if constexpr (BIAS) { v[0] += bcol[j].x; v[1] += bcol[j].y; v[2] += bcol[j].z; v[3] += bcol[j].w; }
      if constexpr (GELU) {
        const f32x2_t lo = gelu_poly2(f32x2_t{v[0], v[1]}), hi = gelu_poly2(f32x2_t{v[2], v[3]});
        v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
      }
      if constexpr (RES) {
        const float4 r = *(const float4*)(lds_wave + (round & 1) * 16384 + sidx * 1024 + lane * 16);
        v[0] += r.x * rsc[j].x; v[1] += r.y * rsc[j].y; v[2] += r.z * rsc[j].z; v[3] += r.w * rsc[j].w;
      }
      const int64_t idx = row * g.ldc + col0 + j * 16;
      if constexpr (OUT_BF16) {
        u16x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = f32_to_bf16(v[k]);
        const uint2 pk = __builtin_bit_cast(uint2, o);
        if constexpr ((j & 1) == 0) {
          pend = pk;                       // first sub-tile of the pair: keep
        } else {
          // vdst = the pair's first sub-tile, src = its second: the upper half of `pend` and the lower half of `pk` change places
          const auto sx = __builtin_amdgcn_permlane32_swap(pend.x, pk.x, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(pend.y, pk.y, false, false);
          // lower half: own quad of sub-tile j-1 | the partner's (the next quad of that sub-tile); upper half: the partner's quad of sub-tile j | own
          if (row < g.M) *(uint4*)((unsigned short*)g.C + row * g.ldc + colp + (j - 1) * 16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      } else {
        *(float4*)((float*)g.C + idx) = make_float4(v[0], v[1], v[2], v[3]);
        if (has_cb) {
          u16x4_t o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = f32_to_bf16(v[k]);
          *(u16x4_t*)(g.c_bf16 + row * g.ld_c_bf16 + col0 + j * 16) = o;
        }
        ssq[i2][j / 4] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      }
    });
    if constexpr (RES && round + 2 < 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this round's LDS reads are retired: its half of the buffer takes round + 2
      res_dma(std::integral_constant<int, round + 2>{});
    }
    if constexpr (RES) {
      if (has_ssq) {   // one slot per 64 columns (the layout every other kernel writes): this wave fills two per row
        static_for<4>([&](auto qt) {
          constexpr int i2 = decltype(qt)::value / 2, hh = decltype(qt)::value % 2;
          float t = ssq[i2][hh];
          t += __shfl_xor(t, 16, 64);   // the four lanes l, l + 16, l + 32, l + 48 share a row
          t += __shfl_xor(t, 32, 64);
          const int64_t r = row0 + (round * 2 + i2) * 16;
          if (lane < 16 && r < g.M) g.row_sumsq[r * g.ld_row_sumsq + (int64_t)((unsigned)nw >> 6) + hh] = t;
        });
      }
    }
  });
}

template <int OUT_BF16, int ROLE>
__global__ __launch_bounds__(256, 1) void gemm_nt_4w256(const tribe_gemm_desc g, int tiles_m, int tiles_n) {
  using namespace w4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef TRIBE_GEMM_STAMPS4W
  const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3
  const int wr = wave >> 1, wc = wave & 1;                    // 2 x 2 waves of 128 x 128

  int tm, tn;
  tile_coords(blockIdx.x, tiles_m, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;

  const int64_t z = blockIdx.y;
  const int64_t b1 = z / g.batch0, b0 = z - b1 * g.batch0;
  const int64_t b1g = g.gather1 ? g.gather1[b1] : b1;
  const unsigned short* A = (const unsigned short*)g.A + (g.gather_a ? b1g : b1) * g.sA1 + b0 * g.sA0;
  const unsigned short* B = (const unsigned short*)g.B + (g.gather_b ? b1g : b1) * g.sB1 + b0 * g.sB0;

  // ---- LDS-DMA pieces: piece p (0..7 per operand) of this wave = 8 tile rows x 128 bytes, rows (wave * 8 + p) * 8 ..; row r of an operand
  // tile at r * 128 with the 16-byte chunks XOR-swizzled by r & 7 on the SOURCE side.  Per-lane byte offsets from the tile's first row,
  // shifted so that the immediate offset 1024 * (p & 3) of the instruction lands on the right memory row (see the header).
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  unsigned voff_a[8], voff_b[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = (wave * 8 + p) * 8 + srow;
    const int64_t ra = (m0 + r < g.M ? m0 + r : g.M - 1) - m0, rb = (n0 + r < g.N ? n0 + r : g.N - 1) - n0;   // clamp: edge rows re-read a valid row
    voff_a[p] = (unsigned)(ra * g.lda * 2 + schunk * 16 + 3072 - 1024 * (p & 3));
    voff_b[p] = (unsigned)(rb * g.ldb * 2 + schunk * 16 + 3072 - 1024 * (p & 3));
  }
  // (the operand bases are wave-uniform by construction; readfirstlane makes that provable, so that they can be "s" operands -- guide T20)
  auto uniform = [](const char* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
  };
  const char* abase = uniform((const char*)(A + m0 * g.lda) - 3072);
  const char* bbase = uniform((const char*)(B + n0 * g.ldb) - 3072);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)smem);
  unsigned dst_cur = lds0 + wave * 8192, dst_nxt = dst_cur + 65536;   // this wave's piece area in the buffer of K-tile t / t+1

  f32x4_t acc[8][8];
  const int frow = lane & 15, fq = lane >> 4;
  const unsigned coff0 = ((fq ^ (frow & 7)) << 4);
  // fragment read addresses of the current / next buffer, k-step 0 / 1 (fragment i at + i * 2048, an immediate)
  unsigned ca0 = lds0 + (wr * 128 + frow) * 128 + coff0, ca1 = lds0 + (wr * 128 + frow) * 128 + (coff0 ^ 64);
  // B fragment rows are read PERMUTED: MFMA row n' = 4 q + r (q = 0..3) holds tile row 4 sigma(q) + r, sigma = (0, 2, 1, 3), so that after the
  // swapped-operand MFMA lanes l and l + 32 hold ADJACENT column quads of one row -- what the 16-byte bf16 store of epilogue_w4 needs
  const int prow = (frow & 3) | ((frow & 4) << 1) | ((frow & 8) >> 1);
  const unsigned bsw = (unsigned)(((fq ^ (prow & 7)) << 4));   // the source-side swizzle follows the row actually read
  unsigned cb0 = lds0 + 32768 + (wc * 128 + prow) * 128 + bsw, cb1 = lds0 + 32768 + (wc * 128 + prow) * 128 + (bsw ^ 64);
  unsigned na0 = ca0 + 65536, na1 = ca1 + 65536, nb0 = cb0 + 65536, nb1 = cb1 + 65536;

  const int nk = (int)(g.K / BK);
#ifdef TRIBE_GEMM_NO_KROT
  const int krot = 0;
#else
  const int krot = ((tn & 7) + (tm & 3)) % nk;   // K rotation (see the 8-wave kernel)
#endif
  auto kbytes = [&](int t) { int k = t + krot; k = k >= nk ? k - nk : k; return (int64_t)k * (BK * 2); };

  bf16x8_t fa[2][8], fb[2][8];   // [fragment set][fragment]
  auto dma = [&](auto pc, unsigned dst, const char* ab, const char* bb) {
    constexpr int p = decltype(pc)::value;   // 0..7 A pieces, 8..15 B pieces
    if constexpr ((p & 3) == 0) set_m0(dst + (p & 4) * 1024 + (p >= 8 ? 32768 : 0));
    glds<1024 * (p & 3)>(p >= 8 ? voff_b[p & 7] : voff_a[p & 7], p >= 8 ? bb : ab);
  };
  auto rd = [&](auto idx, int set, unsigned aaddr, unsigned baddr) {   // B fragments first: the next phase's first MFMAs need all eight
    constexpr int r = decltype(idx)::value;
    if constexpr (r < 8) lds_rd<r * 2048>(fb[set][r], baddr); else lds_rd<(r - 8) * 2048>(fa[set][r - 8], aaddr);
  };

  // ---- prologue: K-tile 0 landed, its k-step 0 in fragment set 0; K-tile 1 in flight ----
  // Both K-tiles are requested back to back (their cold-miss latencies overlap; requested one after the other's landing, K-tile 1 stalled
  // the first landed barrier: ~11 000 cycles of prologue per tile in the stamps of profiles/r03_u_4w_lab.txt); the accumulators are
  // zeroed while the loads fly.
  {
    const char* a0p = abase + kbytes(0);
    const char* b0p = bbase + kbytes(0);
    static_for<16>([&](auto pc) { dma(pc, dst_cur, a0p, b0p); });
  }
  if (nk > 1) {
    const char* a1p = abase + kbytes(1);
    const char* b1p = bbase + kbytes(1);
    static_for<16>([&](auto pc) { dma(pc, dst_nxt, a1p, b1p); });
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+a"(acc[i][j]));   // (pins the zeroing HERE, under the loads' flight, instead of after the wait)
    }
  if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // K-tile 0 landed; the 16 youngest requests are K-tile 1's
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<16>([&](auto r) { rd(r, 0, ca0, cb0); });
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" ::: "memory");   // (the nop: compiler-written accumulator zeros -> first asm MFMA)
  __builtin_amdgcn_sched_barrier(0);

  // One K-tile: phase B then phase A.  DMA = 1: K-tile t+2 exists and is fetched (operand bases an / bn) into the buffer K-tile t leaves;
  // DMA = 0 (the last two K-tiles): nothing to fetch, the landed wait is a full drain.  NEXT = 0 on the very last K-tile: no reads ahead.
  auto ktile_body = [&](auto dma_c, auto next_c, const char* an, const char* bn) {
    constexpr int DMA = decltype(dma_c)::value, NEXT = decltype(next_c)::value;
    static_for<64>([&](auto mc) {   // ---- phase B: MFMAs (t, 0) on set 0
      constexpr int mm = decltype(mc)::value, i = mm / 8, j = mm % 8;
      if constexpr (mm == WB) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave's reads of K-tile t are retired: its buffer may be overwritten
      }
      mfma(acc[i][j], fb[0][j], fa[0][i]);   // operands swapped: the sub-tile accumulates TRANSPOSED (see epilogue_w4)
      if constexpr (mm % 2 == 1 && mm / 2 < 16) rd(std::integral_constant<int, mm / 2>{}, 1, ca1, cb1);
      if constexpr (DMA && mm >= WB && (mm - WB) % DPM == DPM - 1) dma(std::integral_constant<int, (mm - WB) / DPM>{}, dst_cur, an, bn);
    });
    static_for<64>([&](auto mc) {   // ---- phase A: MFMAs (t, 1) on set 1
      constexpr int mm = decltype(mc)::value, i = mm / 8, j = mm % 8;
      if constexpr (mm == RB && NEXT) {
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_PIECES + NA1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // K-tile t+1 has landed for every wave
      }
      mfma(acc[i][j], fb[1][j], fa[1][i]);
      if constexpr (NEXT && mm >= RB && (mm - RB) % 2 == 1 && (mm - RB) / 2 < 16) rd(std::integral_constant<int, (mm - RB) / 2>{}, 0, na0, nb0);
      if constexpr (DMA && mm % DPM == DPM - 1 && NB_PIECES + mm / DPM < 16) dma(std::integral_constant<int, NB_PIECES + mm / DPM>{}, dst_cur, an, bn);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the next K-tile's buffer becomes the current one
    unsigned t0;
    t0 = ca0; ca0 = na0; na0 = t0;  t0 = ca1; ca1 = na1; na1 = t0;
    t0 = cb0; cb0 = nb0; nb0 = t0;  t0 = cb1; cb1 = nb1; nb1 = t0;
    t0 = dst_cur; dst_cur = dst_nxt; dst_nxt = t0;
  };
#ifdef TRIBE_GEMM_STAMPS4W   // diagnostic build only: cycles / 100-MHz ticks of workgroup 0's K loop -> tribe_debug_4w (never quote its run time)
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  int t = 0;
  for (; t + 2 < nk; ++t) ktile_body(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, abase + kbytes(t + 2), bbase + kbytes(t + 2));
  if (t + 1 < nk) { ktile_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, abase, bbase); ++t; }
  ktile_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, abase, bbase);
#ifdef TRIBE_GEMM_STAMPS4W
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    g_dbg_4w[0] = __builtin_amdgcn_s_memtime() - st_c0;
    g_dbg_4w[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
    g_dbg_4w[2] = (unsigned long long)nk;
    g_dbg_4w[3] = st_c0 - st_entry;   // prologue: entry -> K loop
  }
  const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
#endif
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // last asm MFMAs -> compiler reads of the accumulators
  __builtin_amdgcn_s_barrier();                       // every wave is past its last fragment read: the buffers become the epilogue's scratch

  // (the launcher guarantees: un-batched, N a multiple of 256, 16-byte aligned operands, the operator set of ROLE -- w4_role_ok)
  epilogue_w4<OUT_BF16, ROLE>(g, acc, m0 + wr * 128, n0 + wc * 128, lane, smem + wave * 32768);
#ifdef TRIBE_GEMM_STAMPS4W
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) g_dbg_4w[4] = __builtin_amdgcn_s_memtime() - st_e0;   // epilogue incl. store drain
#endif
}

}  // namespace




// ---------------------------------------------------------------------------------------------
// Launch tables.  gemm.hip is compiled in four parts (Makefile: -DTRIBE_GEMM_PART=0 .. 3 -> gemm.o, gemm_p1.o, gemm_p2.o, gemm_p3.o) so that
// the instantiations build side by side: part 0 = the C entry points + the 8-wave 256 x 256 kernel (and its transposed-operand form),
// part 1 = the 256 x 192 kernel, part 2 = the two 128 x 128 kernels, part 3 = the 4-wave 256 x 256 kernel.  A build without the macro compiles everything in one unit.
// ROLE only gives each call site of the path its own kernel symbol, so that rocprofv3 --stats reports per-operator rows.
// ---------------------------------------------------------------------------------------------
#ifndef TRIBE_GEMM_PART
#define TRIBE_GEMM_PART -1
#endif
#define TRIBE_GEMM_HAS_PART(p) (TRIBE_GEMM_PART < 0 || TRIBE_GEMM_PART == (p))

namespace tribe_gemm_detail {
enum { KIND_SMALL = 0, KIND_BIG = 1, KIND_RING = 2, KIND_BIG4W = 3 };
// (output dtype, role symbol) pairs that exist as kernels; every other combination runs under the GENERIC symbol
#define TRIBE_GEMM_PAIRS(X)                                                                                            \
  X(0, 1, TRIBE_ROLE_GENERIC) X(1, 0, TRIBE_ROLE_GENERIC) X(2, 1, TRIBE_ROLE_EXT) X(3, 0, TRIBE_ROLE_EXT)              \
  X(4, 0, TRIBE_ROLE_PROJECTOR) X(5, 1, TRIBE_ROLE_QKV) X(6, 0, TRIBE_ROLE_ATTN_SCORES) X(7, 1, TRIBE_ROLE_ATTN_PV)    \
  X(8, 0, TRIBE_ROLE_OUT_PROJ) X(9, 1, TRIBE_ROLE_FF1) X(10, 0, TRIBE_ROLE_FF2) X(11, 0, TRIBE_ROLE_VOXEL_HEAD)
void launch_big4(int pair, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n);
void launch_big4_tn(int bf, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n, const SkSched& sk);
void launch_big3(int pair, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n);
void launch_ring(int pair, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n, int splits, float* ws);
void launch_splitk_epilogue(int bf, hipStream_t s, const tribe_gemm_desc* d, int splits, const float* ws);
void launch_4w(int pair, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n);
void launch_small(int pair, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n);

// the dynamic-LDS attribute is set once per kernel AND device (one process may drive several GPUs)
template <auto KERNEL, int THREADS, int SMEM, class... Extra>
static void launch_k(dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n, const Extra&... extra) {
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS, 1, 1), SMEM, s, *d, tiles_m, tiles_n, extra...);
}
inline const SkSched& no_stream_k() {
  static const SkSched none = [] { SkSched k{}; k.tiles_dp = 0x7fffffff; return k; }();
  return none;
}
#define TRIBE_GEMM_TABLE(NAME)                                                                                         \
  void NAME(int pair, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n) {                  \
    switch (pair) {                                                                                                    \
      TRIBE_GEMM_PAIRS(TRIBE_GEMM_CASE_##NAME)                                                                         \
      default: break;                                                                                                  \
    }                                                                                                                  \
  }
#if TRIBE_GEMM_HAS_PART(0)
#define TRIBE_GEMM_CASE_launch_big4(idx, bf, role) \
  case idx: launch_k<gemm_nt_256x256x64<bf, role, 0, 4>, 512, big::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n, no_stream_k()); break;
TRIBE_GEMM_TABLE(launch_big4)
void launch_big4_tn(int bf, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n, const SkSched& sk) {
  if (bf) launch_k<gemm_nt_256x256x64<1, TRIBE_ROLE_GENERIC, 1, 4>, 512, big::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n, sk);
  else launch_k<gemm_nt_256x256x64<0, TRIBE_ROLE_GENERIC, 1, 4>, 512, big::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n, sk);
}
#endif
#if TRIBE_GEMM_HAS_PART(1)
#define TRIBE_GEMM_CASE_launch_big3(idx, bf, role) \
  case idx: launch_k<gemm_nt_256x256x64<bf, role, 0, 3>, 512, 2 * (big::A_BYTES + 192 * BK * 2)>(grid, s, d, tiles_m, tiles_n, no_stream_k()); break;
TRIBE_GEMM_TABLE(launch_big3)
#endif
#if TRIBE_GEMM_HAS_PART(2)
#define TRIBE_GEMM_CASE_launch_ring(idx, bf, role) \
  case idx: launch_k<gemm_nt_ring128<bf, role>, 512, ring::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n, splits, ws); break;
void launch_ring(int pair, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n, int splits, float* ws) {
  switch (pair) {
    TRIBE_GEMM_PAIRS(TRIBE_GEMM_CASE_launch_ring)
    default: break;
  }
}
void launch_splitk_epilogue(int bf, hipStream_t s, const tribe_gemm_desc* d, int splits, const float* ws) {
  const int64_t threads = d->M * (d->N / 4);
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (bf) hipLaunchKernelGGL(splitk_epilogue_kernel<1>, grid, dim3(256), 0, s, *d, splits, ws);
  else hipLaunchKernelGGL(splitk_epilogue_kernel<0>, grid, dim3(256), 0, s, *d, splits, ws);
}
#define TRIBE_GEMM_CASE_launch_small(idx, bf, role) \
  case idx: launch_k<gemm_nt_128x128x64<bf, role>, 256, small::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n); break;
TRIBE_GEMM_TABLE(launch_small)
#endif
#if TRIBE_GEMM_HAS_PART(3)
#ifdef TRIBE_GEMM_STAMPS4W
extern "C" int tribe_debug_4w(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_4w), 64); }
#endif
void launch_4w(int pair, dim3 grid, hipStream_t s, const tribe_gemm_desc* d, int tiles_m, int tiles_n) {
  switch (pair) {   // the four encoder GEMMs with compile-time operator sets (epilogue_w4)
    case 5: launch_k<gemm_nt_4w256<1, TRIBE_ROLE_QKV>, 256, w4::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n); break;
    case 8: launch_k<gemm_nt_4w256<0, TRIBE_ROLE_OUT_PROJ>, 256, w4::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n); break;
    case 9: launch_k<gemm_nt_4w256<1, TRIBE_ROLE_FF1>, 256, w4::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n); break;
    case 10: launch_k<gemm_nt_4w256<0, TRIBE_ROLE_FF2>, 256, w4::SMEM_BYTES>(grid, s, d, tiles_m, tiles_n); break;
    default: break;
  }
}
#endif
}  // namespace tribe_gemm_detail

#if TRIBE_GEMM_HAS_PART(0)
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

int tribe_internal_prof_before(int role, double flops, hipStream_t s) { return prof_before(role, flops, s); }
void tribe_internal_prof_after(int slot, hipStream_t s) { prof_after(slot, s); }

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

namespace {
// 16 lanes per row (each 4 partials per step, shuffle reduction): at M = 4096 one thread per row was 16 workgroups walking 48-64 partials
// serially -- 9.6 us per launch, 15 launches per forward = 1.9 % of the B = 4 step (profiles/r03_e_r1_kernel_stats.txt).
__global__ __launch_bounds__(256) void rownorm_scale_kernel(const float* __restrict__ partial, int64_t rows, int64_t n_partial,
                                                            const float* __restrict__ g, float gain_scale, float eps, float* __restrict__ scale) {
  const int sub = threadIdx.x & 15;
  const int64_t m = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  float ss = 0.f;
  if (m < rows) {
    const float* p = partial + m * n_partial;
    if ((n_partial & 3) == 0 && (((uintptr_t)partial) & 15) == 0) {
      for (int64_t i = sub * 4; i < n_partial; i += 64) {
        const float4 v = *(const float4*)(p + i);
        ss += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int64_t i = sub; i < n_partial; i += 16) ss += p[i];
    }
  }
  ss += __shfl_xor(ss, 1, 64);
  ss += __shfl_xor(ss, 2, 64);
  ss += __shfl_xor(ss, 4, 64);
  ss += __shfl_xor(ss, 8, 64);
  if (m < rows && sub == 0) scale[m] = g[0] * gain_scale / fmaxf(sqrtf(ss), eps);
}
}  // namespace

extern "C" int tribe_rownorm_scale_fwd(const float* partial, int64_t rows, int64_t n_partial, const float* g, float gain_scale, float eps,
                                       float* scale, void* stream) {
  TRIBE_REQUIRE(partial && g && scale && rows > 0 && n_partial > 0, "tribe_rownorm_scale_fwd: bad argument");
  hipLaunchKernelGGL(rownorm_scale_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, (hipStream_t)stream, partial, rows, n_partial, g,
                     gain_scale, eps, scale);
  TRIBE_LAUNCH_CHECK();
  return 0;
}


#ifndef TRIBE_GEMM_4W_DEFAULT
#define TRIBE_GEMM_4W_DEFAULT 0   // see gemm_plan: after the 8-wave kernel took over its epilogue techniques the two tie; tile_hint 5 selects it
#endif
namespace tribe_gemm_detail {
// Which kernel and tile a launch gets.  sumsq_cols = columns per row_sumsq slot (one slot per wave column group).
struct GemmPlan { int kind, bm, bn, sumsq_cols; int splits = 1; };
// true when the descriptor carries exactly the operator set epilogue_w4 compiles for its role (and the alignments its vector accesses need)
static bool w4_role_ok(const tribe_gemm_desc* d) {
  const bool res_role = d->role == TRIBE_ROLE_OUT_PROJ || d->role == TRIBE_ROLE_FF2;
  const bool bf_role = d->role == TRIBE_ROLE_QKV || d->role == TRIBE_ROLE_FF1;
  if (!res_role && !bf_role) return false;
  if (d->batch1 * d->batch0 != 1 || d->trans_ab || d->gather1 || d->N % 256 != 0 || d->rowadd || d->gadd || d->aux) return false;
  if (d->c_dtype != (bf_role ? TRIBE_BF16 : TRIBE_F32) || d->ldc % 4 != 0 || ((uintptr_t)d->C % 16) != 0) return false;
  const bool wants_bias = d->role == TRIBE_ROLE_FF1 || d->role == TRIBE_ROLE_FF2;
  if (wants_bias ? (d->bias_mode != TRIBE_BIAS_COL || !d->bias || ((uintptr_t)d->bias % 16) != 0) : d->bias_mode != TRIBE_BIAS_NONE) return false;
  if (d->act != (d->role == TRIBE_ROLE_FF1 ? TRIBE_ACT_GELU : TRIBE_ACT_NONE)) return false;
  if (res_role) {
    if (!d->res || d->ldres % 4 != 0 || ((uintptr_t)d->res % 16) != 0 || d->row_scale) return false;
    if (d->res_scale && ((uintptr_t)d->res_scale % 16) != 0) return false;
    if (d->c_bf16 && (d->ld_c_bf16 % 4 != 0 || ((uintptr_t)d->c_bf16 % 8) != 0)) return false;
  } else {
    if (d->res || d->res_scale || d->c_bf16 || d->row_sumsq) return false;
  }
  return (int64_t)255 * (d->lda > d->ldb ? d->lda : d->ldb) * 2 + 8192 < (1ll << 32);   // 32-bit per-lane offsets of the LDS-DMA pieces
}
static GemmPlan gemm_plan(const tribe_gemm_desc* d) {
  const int64_t nz = d->batch1 * d->batch0;
  auto tiles = [&](int64_t bm, int64_t bn) { return ((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn) * nz; };
  const int64_t t256 = tiles(256, 256), t128 = tiles(128, 128);
  const bool fused_norm = d->c_bf16 || d->row_sumsq || d->row_scale;
  // 256^2 tiles when both extents fill them and the grid still covers the chip, else 128^2
  int use_big = (d->M >= 256 && d->N >= 256 && t256 >= 96);
  // Narrow, short-K products whose 256^2 grid is under two rounds of the 256 CUs (ViT-g attention projection: 8192 x 1408 x 1408 =
  // 192 tiles) run faster on 128^2 tiles (60 -> 47 us; profiles/r02_o_gemm_shapes.txt); with K > 2048 or N > 2048 the 256^2 tiles'
  // higher operand reuse wins back more than the idle CUs cost.
  if (use_big && t256 < 512 && t128 >= 512 && d->K <= 2048 && d->N <= 2048) use_big = 0;
  // 256-wide tiles that hang a quarter or more over the edge of a narrow C (dQ = dS K per head: N = 384 fills 1.5 of them) lose more MFMA
  // work than 128^2 tiles cost: batched 1024 x 384 x 1024, 128 batches: 246 -> 204 us
  if (use_big && t128 >= 512 && (double)t128 * 128 * 128 * 1.25 <= (double)t256 * 256 * 256) use_big = 0;
  if (fused_norm && d->N % 128 != 0) use_big = 1;   // (the launcher then reports the N it needs)
  if (d->tile_hint == 1 || d->tile_hint == 3) use_big = 0;
  if (d->tile_hint == 2 || d->tile_hint == 4 || d->tile_hint == 5) use_big = 1;
  if (d->trans_ab) return {KIND_BIG, 256, 256, 64};
  if (!use_big) {
    // one workgroup per CU or fewer: the ring kernel (three K-tiles in flight); more: the double-buffered kernel, whose two or three
    // co-resident workgroups per CU cover for each other
    const bool ring = d->tile_hint == 3 || (d->tile_hint != 1 && t128 <= 256 && d->K >= 256);
    GemmPlan plan{ring ? KIND_RING : KIND_SMALL, 128, 128, 64};
    // split-K (desc.stream_k): a grid of at most half a round whose tiles have K-steps to share out -- as many workgroups per tile as fit one
    // round, at least 8 K-steps each, at most 8 shares; operators the second launch knows, 16-byte accessible operands
    if (ring && d->stream_k && nz == 1 && !d->gather1 && !d->gadd && !d->aux && (d->act == TRIBE_ACT_NONE || d->act == TRIBE_ACT_GELU) &&
        !(d->res && d->rowadd) && d->N % 4 == 0 && (!d->row_sumsq || d->N % 64 == 0) && d->ldc % 4 == 0 && ((uintptr_t)d->C % 16) == 0 &&
        (d->bias_mode != TRIBE_BIAS_COL || ((uintptr_t)d->bias % 16) == 0) && (!d->res || (d->ldres % 4 == 0 && ((uintptr_t)d->res % 16) == 0)) &&
        (!d->res_scale || ((uintptr_t)d->res_scale % 16) == 0) && (!d->rowadd || (d->ld_rowadd % 4 == 0 && ((uintptr_t)d->rowadd % 16) == 0)) &&
        (!d->c_bf16 || (d->ld_c_bf16 % 4 == 0 && ((uintptr_t)d->c_bf16 % 8) == 0))) {
      const int64_t nk = d->K / BK;
      int64_t sp = 256 / t128;
      if (sp > nk / 8) sp = nk / 8;
      if (sp > 8) sp = 8;
      if (sp >= 2) plan.splits = (int)sp;
    }
    return plan;
  }
  // Tile quantisation on 256 CUs: a grid of 256 x 192 tiles when that cuts the rounds' worth of work (BASELINE config at B = 4, M = 4096:
  // QKV 576 -> 768 tiles = 3 rounds of 3/4-size tiles instead of 3 of full size; out-proj / FF2 192 -> 256 tiles: the whole chip
  // instead of 3/4 of it).  A 192-wide tile costs ~0.78 of a 256-wide one; large grids keep 256^2 (higher operand reuse).
  int bn = 256;
  if (d->tile_hint == 4) bn = 192;
  else if (d->tile_hint == 0 && d->N % 192 == 0 && t256 <= 2048) {
    const int64_t t192 = tiles(256, 192);
    const double cost4 = (double)((t256 + 255) / 256), cost3 = (double)((t192 + 255) / 256) * 0.78;
    if (cost3 < 0.95 * cost4) bn = 192;
  }
  if (fused_norm && d->N % bn != 0) bn = (d->N % 256 == 0) ? 256 : 192;
  // 256 x 256 tiles of the four encoder GEMMs: the one-wave-per-SIMD kernel, whose epilogue is compiled for exactly their operator sets
  // (tile_hint 2 keeps the 8-wave form for A/B runs; everything else -- other roles, batched, transposed operands -- stays 8-wave)
  // The one-wave-per-SIMD kernel (tile_hint 5; QKV / FF1 / out-proj / FF2 with their model epilogues).  History of the default: with the round-2
  // epilogue in the 8-wave kernel it won QKV / FF1 by 4.4 / 4.5 % (profiles/r03_z1_4w_lab.txt); its epilogue techniques (transposed accumulation,
  // paired 16-byte bf16 stores) then went into the 8-wave kernel (TACC), after which the two tie in isolation (profiles/r03_z2_4w_lab.txt) and in the
  // whole step (same box, bench.py: 8-wave only 639.7 - 640.4 k TRs/s, 4-wave for QKV / FF1 637.2 - 638.2 k: the step runs at the chip's power
  // limit, a kernel that draws less lets its neighbours clock higher and vice versa).  Default: 8-wave everywhere.
  const bool w4_default = TRIBE_GEMM_4W_DEFAULT && d->tile_hint == 0 && (d->role == TRIBE_ROLE_QKV || d->role == TRIBE_ROLE_FF1);
  if (bn == 256 && (d->tile_hint == 5 || w4_default) && w4_role_ok(d)) return {KIND_BIG4W, 256, 256, 64};
  return {KIND_BIG, 256, bn, bn / 4};
}
}  // namespace tribe_gemm_detail

namespace tribe_gemm_detail {
// Stream-K plan of a launch with `tiles` output tiles of nk K-steps each (see SkSched): fills *sk and returns the grid size.  Whole rounds of
// W tiles stay whole; the remainder is cut into W runs.  Not worth it (or not possible) -> the plain grid.
unsigned plan_stream_k(int tiles, int nk, SkSched* sk) {
  static int cus[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  const int W = cus[dev] < 256 ? cus[dev] : 256;
  const int rem = tiles % W;
  *sk = no_stream_k();
  // Worth it only for remainders up to half a round.  Measured at B = 16 (profiles/r03_z12_gemm_streamk.txt, same numbers with atomics and
  // with the workspace): remainder 64 of 256 (dW of FF1 / FF2: 576 tiles; the projectors: 64 tiles) 1284 -> 1076 us / 1294 -> 1086 us /
  // 328 -> 161 us -- there a run is a clean quarter of a tile; remainder 176 (QKV) 848 -> 1007 us and 144 (out-proj) 334 -> 439 us: SLOWER --
  // every run straddles two tiles at its own K offset, so the 32 workgroups of an XCD stop sharing operand panels in its L2 (the 4 x 8 patch
  // of tile_coords) and the region turns memory-bound; and every part pays a prologue + epilogue (~45 us) for at most 1/2 tile of work.
  // Also: runs of at least 8 K-steps.
  if (rem == 0 || rem * 2 > W || (int64_t)rem * nk < (int64_t)W * 8) return (unsigned)tiles;
  sk->tiles_dp = tiles - rem;
  sk->rem_units = rem * nk;
  sk->q = (sk->rem_units + W - 1) / W;
  sk->workers = W;
  // second parts, longest first
  int len2[256], n2 = 0;
  unsigned short who[256];
  for (int w = 0; w < W; ++w) {
    const int u0 = w * sk->q;
    if (u0 >= sk->rem_units) break;
    const int run = sk->q < sk->rem_units - u0 ? sk->q : sk->rem_units - u0;
    const int x = u0 % nk, first = run < nk - x ? run : nk - x;
    if (first < run) { len2[n2] = run - first; who[n2] = (unsigned short)w; ++n2; }
  }
  for (int i = 1; i < n2; ++i) {   // insertion sort, descending, stable
    const int l = len2[i]; const unsigned short v = who[i];
    int j = i - 1;
    for (; j >= 0 && len2[j] < l; --j) { len2[j + 1] = len2[j]; who[j + 1] = who[j]; }
    len2[j + 1] = l; who[j + 1] = v;
  }
  for (int i = 0; i < n2; ++i) sk->second_worker[i] = who[i];
  return (unsigned)(sk->tiles_dp + W + n2);
}
}  // namespace tribe_gemm_detail

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
  TRIBE_REQUIRE((d->trans_ab || (d->lda >= d->K && d->ldb >= d->K)) && d->ldc >= ((d->act == TRIBE_ACT_SWIGLU || d->act == TRIBE_ACT_GLU) ? d->N / 2 : d->N),
                "tribe_gemm_bf16: leading dimension too small");
  TRIBE_REQUIRE((d->act != TRIBE_ACT_SWIGLU && d->act != TRIBE_ACT_GLU) || (d->N % 2 == 0 && !d->res && !d->rowadd && !d->gadd && d->bias_mode != TRIBE_BIAS_ROW),
                "tribe_gemm_bf16: SWIGLU needs an even N and no residual / row adds");
  TRIBE_REQUIRE(d->c_dtype == TRIBE_F32 || d->c_dtype == TRIBE_BF16, "tribe_gemm_bf16: c_dtype must be f32 or bf16");
  TRIBE_REQUIRE(d->bias_mode == TRIBE_BIAS_NONE || d->bias != nullptr, "tribe_gemm_bf16: bias_mode set without bias");
  TRIBE_REQUIRE(!d->rowadd || d->rowadd_period > 0, "tribe_gemm_bf16: rowadd needs a positive period");
  TRIBE_REQUIRE(!d->gadd || (d->gadd_index && d->gadd_div > 0), "tribe_gemm_bf16: gadd needs index and divisor");
  TRIBE_REQUIRE(!(d->gather_a || d->gather_bias || d->gather_b) || d->gather1, "tribe_gemm_bf16: gather flags set without gather1");
  TRIBE_REQUIRE(d->act != TRIBE_ACT_GELU_BWD || d->aux, "tribe_gemm_bf16: GELU_BWD needs the saved pre-activation in aux");
  TRIBE_REQUIRE(d->act != TRIBE_ACT_MUL_AUX || (d->aux && d->ld_aux == d->ldc), "tribe_gemm_bf16: MUL_AUX needs aux laid out like C (ld_aux == ldc)");
  TRIBE_REQUIRE((d->act != TRIBE_ACT_EXP2 && d->act != TRIBE_ACT_MUL_AUX) || (!d->res && !d->rowadd && !d->gadd && d->bias_mode != TRIBE_BIAS_COL),
                "tribe_gemm_bf16: EXP2 / MUL_AUX take a row bias only");
  // (aux is addressed with C's batch offsets: the GELU pre-activation is un-batched, MUL_AUX's factor shares C's layout)
  TRIBE_REQUIRE(!d->aux || (d->ld_aux >= d->N && (d->batch1 * d->batch0 == 1 || d->act == TRIBE_ACT_MUL_AUX)),
                "tribe_gemm_bf16: aux is supported for un-batched GEMMs (and for MUL_AUX)");
  TRIBE_REQUIRE(d->sBias0 == 0 || d->bias_mode == TRIBE_BIAS_ROW, "tribe_gemm_bf16: sBias0 belongs to a row bias");
  const int64_t nz = d->batch1 * d->batch0;
  if (d->c_bf16 || d->row_sumsq || d->row_scale) {
    // fused ScaleNorm operands exist only in the wait-free epilogue: insist on everything that path needs
    TRIBE_REQUIRE(nz == 1 && !d->rowadd && !d->gadd && !d->aux && (d->act == TRIBE_ACT_NONE || d->act == TRIBE_ACT_GELU),
                  "tribe_gemm_bf16: c_bf16 / row_sumsq / row_scale need an un-batched launch with plain operators");
    TRIBE_REQUIRE(d->ldc % 4 == 0 && ((uintptr_t)d->C % 16) == 0 && (!d->bias || ((uintptr_t)d->bias % 16) == 0) &&
                      (!d->res || (d->ldres % 4 == 0 && ((uintptr_t)d->res % 16) == 0)) && (!d->res_scale || ((uintptr_t)d->res_scale % 16) == 0),
                  "tribe_gemm_bf16: c_bf16 / row_sumsq / row_scale need 16-byte aligned operands");
    TRIBE_REQUIRE((!d->c_bf16 && !d->row_sumsq) || d->c_dtype == TRIBE_F32, "tribe_gemm_bf16: c_bf16 / row_sumsq accompany an f32 C");
    TRIBE_REQUIRE(!d->c_bf16 || (d->ld_c_bf16 >= d->N && d->ld_c_bf16 % 4 == 0 && ((uintptr_t)d->c_bf16 % 8) == 0), "tribe_gemm_bf16: bad c_bf16");
  }
  const tribe_gemm_detail::GemmPlan plan = tribe_gemm_detail::gemm_plan(d);
  if (d->c_bf16 || d->row_sumsq || d->row_scale) {
    TRIBE_REQUIRE(d->N % plan.bn == 0, "tribe_gemm_bf16: c_bf16 / row_sumsq / row_scale need N (%lld) to be a multiple of the tile width %d",
                  (long long)d->N, plan.bn);
    TRIBE_REQUIRE(!d->row_sumsq || d->ld_row_sumsq >= d->N / plan.sumsq_cols, "tribe_gemm_bf16: ld_row_sumsq must cover N / %d slots (tribe_gemm_sumsq_slots)",
                  plan.sumsq_cols);
  }
  if (d->trans_ab) {
    TRIBE_REQUIRE(d->M >= 8 && d->N >= 8 && d->M % 8 == 0 && d->N % 8 == 0 && d->lda >= d->M && d->ldb >= d->N,
                  "tribe_gemm_bf16: trans_ab takes At [K, M] and Bt [K, N] with M and N multiples of 8 and lda >= M, ldb >= N");
    TRIBE_REQUIRE(!d->aux && d->act != TRIBE_ACT_SWIGLU && d->act != TRIBE_ACT_GLU && d->act != TRIBE_ACT_SILU && d->act != TRIBE_ACT_GELU_BWD &&
                      d->act != TRIBE_ACT_EXP2 && d->act != TRIBE_ACT_MUL_AUX && !d->gather_a && !d->gather_b,
                  "tribe_gemm_bf16: trans_ab supports the plain epilogue operators and no operand gather");
  }
  const int64_t tiles_m = (d->M + plan.bm - 1) / plan.bm, tiles_n = (d->N + plan.bn - 1) / plan.bn;
  TRIBE_REQUIRE(tiles_m * tiles_n < (1ll << 31) && nz < 65536, "tribe_gemm_bf16: grid too large");

  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)nz, 1);
  hipStream_t s = (hipStream_t)stream;
  const int role = (d->role >= 0 && d->role < TRIBE_ROLE_COUNT) ? d->role : TRIBE_ROLE_GENERIC;
  const double flops = 2.0 * (double)d->M * (double)d->N * (double)d->K * (double)nz;
  const int slot = prof_before(role, flops, s);
  const bool bf = d->c_dtype == TRIBE_BF16;
  if (d->trans_ab) {   // transposed operands: the 256^2 kernel only, plain epilogue operators
    SkSched sk = tribe_gemm_detail::no_stream_k();
    if (d->stream_k) {
      TRIBE_REQUIRE(!bf && nz == 1 && d->bias_mode == TRIBE_BIAS_NONE && d->act == TRIBE_ACT_NONE && !d->res && !d->rowadd && !d->gadd,
                    "tribe_gemm_bf16: stream_k takes a plain un-batched f32 product");
      grid.x = tribe_gemm_detail::plan_stream_k((int)(tiles_m * tiles_n), (int)(d->K / BK), &sk);
      if (sk.tiles_dp < tiles_m * tiles_n) {
        const int64_t need = (int64_t)2 * sk.workers * SK_SLOT_FLOATS * 4;
        TRIBE_REQUIRE(d->stream_k_ws && d->stream_k_ws_bytes >= need && ((uintptr_t)d->stream_k_ws % 16) == 0,
                      "tribe_gemm_bf16: stream_k needs a 16-byte aligned workspace of %lld bytes (tribe_gemm_stream_k_workspace_bytes)", (long long)need);
        sk.ws = (float*)d->stream_k_ws;
      }
    }
    tribe_gemm_detail::launch_big4_tn(bf ? 1 : 0, grid, s, d, (int)tiles_m, (int)tiles_n, sk);
    if (sk.tiles_dp < tiles_m * tiles_n)
      hipLaunchKernelGGL(streamk_reduce_kernel, dim3((unsigned)(tiles_m * tiles_n - sk.tiles_dp)), dim3(512), 0, s, *d, (int)tiles_m, (int)tiles_n, sk);
    prof_after(slot, s);
    TRIBE_LAUNCH_CHECK();
    return 0;
  }
  const bool ext = d->aux != nullptr || d->act == TRIBE_ACT_SWIGLU || d->act == TRIBE_ACT_GLU || d->act == TRIBE_ACT_SILU ||
                   d->act == TRIBE_ACT_GELU_BWD || d->act == TRIBE_ACT_EXP2 || d->act == TRIBE_ACT_MUL_AUX;
  int pair = bf ? 0 : 1;   // GENERIC
  if (ext) {
    pair = bf ? 2 : 3;
  } else {
    switch (role) {     // the (dtype, role) combinations the encode path launches have kernels of their own
    case TRIBE_ROLE_PROJECTOR: if (!bf) pair = 4; break;
    case TRIBE_ROLE_QKV: if (bf) pair = 5; break;
    case TRIBE_ROLE_ATTN_SCORES: if (!bf) pair = 6; break;
    case TRIBE_ROLE_ATTN_PV: if (bf) pair = 7; break;
    case TRIBE_ROLE_OUT_PROJ: if (!bf) pair = 8; break;
    case TRIBE_ROLE_FF1: if (bf) pair = 9; break;
    case TRIBE_ROLE_FF2: if (!bf) pair = 10; break;
    case TRIBE_ROLE_VOXEL_HEAD: if (!bf) pair = 11; break;
    default: break;
    }
  }
  using namespace tribe_gemm_detail;
  if (plan.kind == KIND_BIG4W) launch_4w(pair, grid, s, d, (int)tiles_m, (int)tiles_n);
  else if (plan.kind == KIND_BIG && plan.bn == 192) launch_big3(pair, grid, s, d, (int)tiles_m, (int)tiles_n);
  else if (plan.kind == KIND_BIG) launch_big4(pair, grid, s, d, (int)tiles_m, (int)tiles_n);
  else if (plan.kind == KIND_RING && plan.splits > 1) {
    const int64_t need = (int64_t)plan.splits * d->M * d->N * 4;
    TRIBE_REQUIRE(d->stream_k_ws && d->stream_k_ws_bytes >= need && ((uintptr_t)d->stream_k_ws % 16) == 0,
                  "tribe_gemm_bf16: this launch is split over K: stream_k_ws must hold %lld bytes (tribe_gemm_stream_k_workspace_bytes)", (long long)need);
    TRIBE_REQUIRE(tiles_m * tiles_n * plan.splits < (1ll << 31) && d->M * (d->N / 4) < (1ll << 39), "tribe_gemm_bf16: grid too large");
    grid.x = (unsigned)(tiles_m * tiles_n * plan.splits);
    launch_ring(pair, grid, s, d, (int)tiles_m, (int)tiles_n, plan.splits, (float*)d->stream_k_ws);
    launch_splitk_epilogue(bf ? 1 : 0, s, d, plan.splits, (const float*)d->stream_k_ws);
  } else if (plan.kind == KIND_RING) launch_ring(pair, grid, s, d, (int)tiles_m, (int)tiles_n, 1, nullptr);
  else launch_small(pair, grid, s, d, (int)tiles_m, (int)tiles_n);
  prof_after(slot, s);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

// the schedule itself, for inspection (tests/test_abi_and_host.py replays the kernel's decode on the host and checks that every K-step of
// every tile is covered exactly once): out = {tiles_dp, rem_units, q, workers, second_worker[0 .. 255]}; returns the grid size
extern "C" int tribe_gemm_stream_k_plan(int32_t tiles, int32_t nk, int32_t* out) {
  TRIBE_REQUIRE(tiles > 0 && nk > 0 && out != nullptr, "tribe_gemm_stream_k_plan: bad argument");
  SkSched sk;
  const unsigned grid = tribe_gemm_detail::plan_stream_k(tiles, nk, &sk);
  out[0] = sk.tiles_dp; out[1] = sk.rem_units; out[2] = sk.q; out[3] = sk.workers;
  for (int i = 0; i < 256; ++i) out[4 + i] = sk.second_worker[i];
  return (int)grid;
}

extern "C" int64_t tribe_gemm_stream_k_workspace_bytes(const tribe_gemm_desc* d) {
  TRIBE_REQUIRE(d != nullptr && d->M > 0 && d->N > 0 && d->K > 0 && d->K % BK == 0, "tribe_gemm_stream_k_workspace_bytes: bad descriptor");
  if (!d->stream_k || d->batch1 * d->batch0 != 1) return 0;
  if (!d->trans_ab) {   // NT form: split-K of a small grid (ring kernel)
    const tribe_gemm_detail::GemmPlan plan = tribe_gemm_detail::gemm_plan(d);
    return plan.splits > 1 ? (int64_t)plan.splits * d->M * d->N * 4 : 0;
  }
  SkSched sk;
  const int tiles = (int)(((d->M + 255) / 256) * ((d->N + 255) / 256));
  if (tribe_gemm_detail::plan_stream_k(tiles, (int)(d->K / BK), &sk) == (unsigned)tiles) return 0;
  return (int64_t)2 * sk.workers * SK_SLOT_FLOATS * 4;
}

extern "C" int tribe_gemm_sumsq_slots(const tribe_gemm_desc* d) {
  TRIBE_REQUIRE(d != nullptr && d->M > 0 && d->N > 0 && d->K > 0 && d->batch1 > 0 && d->batch0 > 0, "tribe_gemm_sumsq_slots: bad descriptor");
  const tribe_gemm_detail::GemmPlan plan = tribe_gemm_detail::gemm_plan(d);
  return (int)(d->N / plan.sumsq_cols);
}
#endif  // part 0
