// Probe for VERDICT r2 item 1: would a ONE-wave-per-SIMD 256 x 256 x 64 GEMM tile (4 waves, 128 x 128 per wave, 256 accumulator registers,
// fragment reads and LDS-DMA issued by the SAME wave that issues the MFMAs) beat the shipped two-waves-per-SIMD kernel?
// The probe is that structure in its simplest honest form: two 64-KiB LDS buffers, per K-tile 16 global_load_lds_dwordx4 per wave for
// K-tile t+1 (MODE 0: in one burst before the MFMAs; MODE 1: one after every 8 MFMAs), 32 ds_read_b128 + 128 v_mfma_f32_16x16x32_bf16
// on K-tile t, one vmcnt(0) + barrier per K-tile.  MODE 2 issues no LDS-DMA at all (reads + MFMAs on stale LDS): the ceiling of the
// structure if staging were free.  It computes a real product (checked against a host reference on a few entries) so that the
// compiler cannot drop anything; plain f32 store.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/gemm_1wave_probe.hip -o ab_tmp/gemm_1wave_probe && ab_tmp/gemm_1wave_probe
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include "../../include/tribe_hip.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <type_traits>
#include <utility>
#include <vector>

__device__ unsigned long long g_clk[4];   // diagnostic: [0] shader cycles, [1] 100-MHz ticks of workgroup 0's K loop, [2] cycles in waits + barrier
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ C,
                                               int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned short* a_src[8];
  const unsigned short* b_src[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = (wave * 8 + p) * 8 + srow;
    a_src[p] = A + (m0 + r) * K + schunk * 8;
    b_src[p] = B + (n0 + r) * K + schunk * 8;
  }
  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  const int a_rd = (wr * 128 + frow) * 128, b_rd = 32768 + (wc * 128 + frow) * 128;
  const int nk = K / 64;
  auto glds = [&](int p, int buf, int kt, bool is_b) {
    char* dst = smem + buf * 65536 + (is_b ? 32768 : 0) + (wave * 8 + p) * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)((is_b ? b_src[p] : a_src[p]) + kt * 64), (lptr_t)dst, 16, 0, 0);
  };
  for (int p = 0; p < 8; ++p) { glds(p, 0, 0, false); glds(p, 0, 0, true); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const char* base = smem + cur * 65536;
    const bool more = t + 1 < nk;
    if (MODE == 0 && more) {
#pragma unroll
      for (int p = 0; p < 8; ++p) { glds(p, cur ^ 1, t + 1, false); glds(p, cur ^ 1, t + 1, true); }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = (((ks * 4 + fq) ^ (frow & 7)) << 4);
      bf16x8_t fa[8], fb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *(const bf16x8_t*)(base + a_rd + i * 2048 + coff);
#pragma unroll
      for (int j = 0; j < 8; ++j) fb[j] = *(const bf16x8_t*)(base + b_rd + j * 2048 + coff);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (MODE == 1 && more) {   // one LDS-DMA piece per 8 MFMAs: 16 pieces over the K-tile's 128 MFMAs
          const int p = ks * 8 + i;
          glds(p & 7, cur ^ 1, t + 1, p >= 8);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(m0 + wr * 128 + i * 16 + fq * 4 + r) * N + n0 + wc * 128 + j * 16 + frow] = acc[i][j][r];
}


// MODE 3: the same tile with a software pipeline across the K-tile boundary.  Two fragment sets F0 / F1 (k-step 0 / 1 of a K-tile):
//   phase A(t): MFMAs of (t-1, k-step 1) on F1  ||  ds_reads of (t, k-step 0) into F0  ||  the 16 LDS-DMA pieces of K-tile t+1
//   phase B(t): MFMAs of (t, k-step 0) on F0    ||  ds_reads of (t, k-step 1) into F1
//   lgkmcnt(0), vmcnt(0), ONE barrier per K-tile (all reads of K-tile t retired, K-tile t+1 landed), then phase A(t+1).
// One ds_read (and in phase A one LDS-DMA piece) per 4 MFMAs, pinned with sched_group_barrier.
template <int DMA_PER_GROUP>
__global__ __launch_bounds__(256, 1) void probe_pipe(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ C,
                                                    int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned short* a_src[8];
  const unsigned short* b_src[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = (wave * 8 + p) * 8 + srow;
    a_src[p] = A + (m0 + r) * K + schunk * 8;
    b_src[p] = B + (n0 + r) * K + schunk * 8;
  }
  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  const int a_rd = (wr * 128 + frow) * 128, b_rd = 32768 + (wc * 128 + frow) * 128;
  const int coff0 = ((fq ^ (frow & 7)) << 4), coff1 = (((4 + fq) ^ (frow & 7)) << 4);
  const int nk = K / 64;
  auto glds = [&](int p, int buf, int kt) {   // p = 0..15: A pieces 0..7, B pieces 8..15
    char* dst = smem + buf * 65536 + (p >= 8 ? 32768 : 0) + (wave * 8 + (p & 7)) * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)((p >= 8 ? b_src[p & 7] : a_src[p & 7]) + kt * 64), (lptr_t)dst, 16, 0, 0);
  };
  bf16x8_t f0a[8], f0b[8], f1a[8], f1b[8];
  // prologue: K-tile 0 landed; F0 <- (0, k-step 0); nothing pending in F1
  for (int p = 0; p < 16; ++p) glds(p, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#define RD(F, base, idx, coff) F[(idx) & 7] = *(const bf16x8_t*)((base) + ((idx) < 8 ? a_rd : b_rd) + ((idx) & 7) * 2048 + (coff))
  {
    const char* base = smem;
#pragma unroll
    for (int r = 0; r < 8; ++r) { f0a[r] = *(const bf16x8_t*)(base + a_rd + r * 2048 + coff0); f0b[r] = *(const bf16x8_t*)(base + b_rd + r * 2048 + coff0); }
    if (1 < nk) { for (int p = 0; p < 16; ++p) glds(p, 1, 1); }
  }
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const char* base = smem + cur * 65536;
    // ---- phase B(t): MFMAs (t, 0) on F0 || reads (t, 1) into F1
#pragma unroll
    for (int grp = 0; grp < 16; ++grp) {
      if (grp < 8) f1a[grp] = *(const bf16x8_t*)(base + a_rd + grp * 2048 + coff1);
      else f1b[grp - 8] = *(const bf16x8_t*)(base + b_rd + (grp - 8) * 2048 + coff1);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0a[i], f0b[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int grp = 0; grp < 16; ++grp) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase A(t+1): MFMAs (t, 1) on F1 || reads (t+1, 0) into F0 || LDS-DMA of K-tile t+2 into the buffer K-tile t occupied
    const bool next = t + 1 < nk, next2 = t + 2 < nk;
    const char* nbase = smem + (cur ^ 1) * 65536;
    if (next) {
#pragma unroll
      for (int grp = 0; grp < 16; ++grp) {
        if (grp < 8) f0a[grp] = *(const bf16x8_t*)(nbase + a_rd + grp * 2048 + coff0);
        else f0b[grp - 8] = *(const bf16x8_t*)(nbase + b_rd + (grp - 8) * 2048 + coff0);
      }
    }
    if (next2) {
#pragma unroll
      for (int p = 0; p < 16; ++p) glds(p, cur, t + 2);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1a[i], f1b[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int grp = 0; grp < 16; ++grp) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      if (DMA_PER_GROUP) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (LDS-DMA)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(m0 + wr * 128 + i * 16 + fq * 4 + r) * N + n0 + wc * 128 + j * 16 + frow] = acc[i][j][r];
}


// MODE 4: the pipeline of probe_pipe with the K loop written as inline-asm statements in source order (volatile asm statements keep their
// order; hipcc only adds address arithmetic): accumulators in compiler-allocated AGPRs ("+a"), fragments in VGPRs, explicit waits.
//   RPM = MFMAs per ds_read in a phase (reads of the NEXT k-step issued in the first 16 * RPM MFMAs), DPM = MFMAs per LDS-DMA piece in the
//   phase behind the barrier.
__device__ __forceinline__ void mfma_a(f32x4_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void lds_rd(bf16x8_t& f, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void glds_asm(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N, class F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (N > 0) { sfor<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

template <int RPM, int DPM>
__global__ __launch_bounds__(256, 1) void probe_asm(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  const unsigned short* src[16];   // per-lane source of piece p: A pieces 0..7, B pieces 8..15
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int r = (wave * 8 + p) * 8 + srow;
    src[p] = A + (m0 + r) * K + schunk * 8;
    src[8 + p] = B + (n0 + r) * K + schunk * 8;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned piece_dst = lds0 + wave * 8192;   // + (p & 7) * 1024 + (p >= 8 ? 32768 : 0) + buf * 65536
  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  const unsigned coff0 = ((fq ^ (frow & 7)) << 4);
  // fragment read addresses: [buffer][k-step]; fragment i at + i * 2048 (immediate)
  unsigned a_ad[2][2], b_ad[2][2];
#pragma unroll
  for (int bu = 0; bu < 2; ++bu)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_ad[bu][ks] = lds0 + bu * 65536 + (wr * 128 + frow) * 128 + (coff0 ^ (ks * 64));
      b_ad[bu][ks] = lds0 + bu * 65536 + 32768 + (wc * 128 + frow) * 128 + (coff0 ^ (ks * 64));
    }
  const int nk = K / 64;   // even
  bf16x8_t fa[2][8], fb[2][8];   // [fragment set][fragment]
  auto dma = [&](int p, int buf, int kt) { glds_asm(src[p] + (int64_t)kt * 64, piece_dst + (p & 7) * 1024 + (p >= 8 ? 32768 : 0) + buf * 65536); };
  auto rd = [&](auto idx, int set, unsigned aaddr, unsigned baddr) {   // read #idx of a k-step: 0..7 A fragments, 8..15 B fragments
    constexpr int r = decltype(idx)::value;
    if constexpr (r < 8) lds_rd<r * 2048>(fa[set][r], aaddr); else lds_rd<(r - 8) * 2048>(fb[set][r - 8], baddr);
  };
  // prologue
  for (int p = 0; p < 16; ++p) dma(p, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  sfor<16>([&](auto r) { rd(r, 0, a_ad[0][0], b_ad[0][0]); });
  if (nk > 1) for (int p = 0; p < 16; ++p) dma(p, 1, 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // one phase: 64 MFMAs on fragment set CS, meanwhile the 16 reads of the next k-step into set CS ^ 1 and (DMA_T >= 0) the 16 LDS-DMA pieces
#define PHASE(CS, RA, RB, DO_DMA, DBUF, DKT)                                                                  \
  sfor<64>([&](auto mc) {                                                                                     \
    constexpr int mm = decltype(mc)::value, i = mm / 8, j = mm % 8;                                           \
    mfma_a(acc[i][j], fa[CS][i], fb[CS][j]);                                                                  \
    if constexpr (mm % RPM == RPM - 1 && mm / RPM < 16) rd(std::integral_constant<int, mm / RPM>{}, (CS) ^ 1, RA, RB); \
    if constexpr (mm % DPM == DPM - 1 && mm / DPM < 16) { if (DO_DMA) dma(mm / DPM, DBUF, DKT); }             \
  });
  for (int t = 0; t < nk; t += 2) {
    const bool d2 = t + 2 < nk, d3 = t + 3 < nk;
    // ---- K-tile t (buffer 0): phase B = MFMAs (t, 0) on set 0 || reads (t, 1) into set 1
    PHASE(0, a_ad[0][1], b_ad[0][1], false, 0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // phase A = MFMAs (t, 1) on set 1 || reads (t+1, 0) from buffer 1 into set 0 || LDS-DMA of K-tile t+2 into buffer 0
    PHASE(1, a_ad[1][0], b_ad[1][0], d2, 0, t + 2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- K-tile t+1 (buffer 1)
    PHASE(0, a_ad[1][1], b_ad[1][1], false, 0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PHASE(1, a_ad[0][0], b_ad[0][0], d3, 1, t + 3)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#undef PHASE
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(m0 + wr * 128 + i * 16 + fq * 4 + r) * N + n0 + wc * 128 + j * 16 + frow] = acc[i][j][r];
}


// MODE 5: as probe_asm, with the LDS-DMA issue made as cheap as the ISA allows:
//   * global_load_lds with a SCALAR base + 32-bit per-lane offset: the per-lane offsets of the 16 pieces never change, the two operand
//     bases advance by 128 bytes per K-tile with scalar adds -- no vector instruction per piece;
//   * four pieces share one M0 write: the instruction's immediate offset (added to the memory AND the LDS address) steps 1024 bytes
//     per piece, the per-lane offsets compensate on the memory side (voff = row offset + 3072 - imm, base - 3072);
//   * M0 is written, not saved and restored (nothing else in the loop uses it);
//   * no branch per piece: past the last K-tile the loads re-fetch the last tile into the dead buffer.
template <int IMM>
__device__ __forceinline__ void glds_s(unsigned voff, const void* sbase) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
__device__ __forceinline__ void set_m0(unsigned v) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(v) : "memory"); }

// the shipped kernel's tile walk (gemm_common.h): every XCD a contiguous run of tile ids, bands of 4 tile rows walked column-major
__device__ __forceinline__ void walk(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int nwg = tiles_m * tiles_n, q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int tile = ((xcd < r8) ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  const int band = tile / (4 * tiles_n), first_m = band * 4;
  const int gm = (tiles_m - first_m < 4) ? tiles_m - first_m : 4;
  const int r = tile - band * 4 * tiles_n;
  tn = r / gm;
  tm = first_m + (r - tn * gm);
}
template <int RPM, int DPM, int DFIRST, int NODMA = 0, int WALK = 0>
__global__ __launch_bounds__(256, 1) void probe_asm2(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ C,
                                                    int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  if (WALK) walk(blockIdx.x, M / 256, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  unsigned voff[8];   // per-lane byte offset of piece p (same for A and B: both are [rows][K] with the same K)
#pragma unroll
  for (int p = 0; p < 8; ++p) voff[p] = (unsigned)(((wave * 8 + p) * 8 + srow) * K * 2 + schunk * 16 + 3072 - 1024 * (p & 3));
  const char* abase = (const char*)(A + m0 * K) - 3072;
  const char* bbase = (const char*)(B + n0 * K) - 3072;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned piece_dst = lds0 + wave * 8192;
  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  const unsigned coff0 = ((fq ^ (frow & 7)) << 4);
  unsigned a_ad[2][2], b_ad[2][2];
#pragma unroll
  for (int bu = 0; bu < 2; ++bu)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_ad[bu][ks] = lds0 + bu * 65536 + (wr * 128 + frow) * 128 + (coff0 ^ (ks * 64));
      b_ad[bu][ks] = lds0 + bu * 65536 + 32768 + (wc * 128 + frow) * 128 + (coff0 ^ (ks * 64));
    }
  const int nk = K / 64;   // even
  bf16x8_t fa[2][8], fb[2][8];
  // piece p (0..15: A 0..7, B 8..15) of the K-tile whose operand bases are (ab, bb) into buffer buf
  auto dma = [&](auto pc, int buf, const char* ab, const char* bb) {
    constexpr int p = decltype(pc)::value;
    if constexpr ((p & 3) == 0) set_m0(piece_dst + (p & 4) * 1024 + (p >= 8 ? 32768 : 0) + buf * 65536);
    glds_s<1024 * (p & 3)>(voff[p & 7], p >= 8 ? bb : ab);
  };
  auto rd = [&](auto idx, int set, unsigned aaddr, unsigned baddr) {   // B fragments first: the next phase's first MFMAs need all of them
    constexpr int r = decltype(idx)::value;
    if constexpr (r < 8) lds_rd<r * 2048>(fb[set][r], baddr); else lds_rd<(r - 8) * 2048>(fa[set][r - 8], aaddr);
  };
  sfor<16>([&](auto pc) { dma(pc, 0, abase, bbase); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  sfor<16>([&](auto r) { rd(r, 0, a_ad[0][0], b_ad[0][0]); });
  sfor<16>([&](auto pc) { dma(pc, 1, abase + 128, bbase + 128); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  const char* an = abase + 256;   // operand bases of the K-tile the NEXT LDS-DMA burst fetches (t + 2), clamped to the last tile
  const char* bn = bbase + 256;
  const char* alast = abase + (int64_t)(nk - 1) * 128;
  const char* blast = bbase + (int64_t)(nk - 1) * 128;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(), wsum = 0;
#define WAITS_BARRIER()                                                                  \
  {                                                                                      \
    const unsigned long long w0 = __builtin_amdgcn_s_memtime();                          \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");             \
    __builtin_amdgcn_s_barrier();                                                        \
    wsum += __builtin_amdgcn_s_memtime() - w0;                                           \
  }

#define PHASE2(CS, RA, RB, DO_DMA, DBUF)                                                                      \
  sfor<64>([&](auto mc) {                                                                                     \
    constexpr int mm = decltype(mc)::value, i = mm / 8, j = mm % 8;                                           \
    mfma_a(acc[i][j], fa[CS][i], fb[CS][j]);                                                                  \
    if constexpr (mm % RPM == RPM - 1 && mm / RPM < 16) rd(std::integral_constant<int, mm / RPM>{}, (CS) ^ 1, RA, RB); \
    if constexpr (DO_DMA && !NODMA && mm >= DFIRST && (mm - DFIRST) % DPM == DPM - 1 && (mm - DFIRST) / DPM < 16) \
      dma(std::integral_constant<int, (mm - DFIRST) / DPM>{}, DBUF, an, bn);                                  \
  });
  for (int t = 0; t < nk; t += 2) {
    PHASE2(0, a_ad[0][1], b_ad[0][1], false, 0)
    WAITS_BARRIER()
    PHASE2(1, a_ad[1][0], b_ad[1][0], true, 0)
    an = an + 128 < alast ? an + 128 : alast;
    bn = bn + 128 < blast ? bn + 128 : blast;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PHASE2(0, a_ad[1][1], b_ad[1][1], false, 0)
    WAITS_BARRIER()
    PHASE2(1, a_ad[0][0], b_ad[0][0], true, 1)
    an = an + 128 < alast ? an + 128 : alast;
    bn = bn + 128 < blast ? bn + 128 : blast;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk[0] = __builtin_amdgcn_s_memtime() - c0;
    g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    g_clk[2] = wsum;
  }
#undef PHASE2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(m0 + wr * 128 + i * 16 + fq * 4 + r) * N + n0 + wc * 128 + j * 16 + frow] = acc[i][j][r];
}


// MODE 6: as probe_asm2 + XCD-aware walk, with the LDS-DMA of K-tile t+2 given a longer flight.  Two barriers per K-tile:
//   phase B(t)  = MFMAs (t, 0) || reads (t, 1) in the first 32 MFMAs; at MFMA WB: lgkmcnt(0) + barrier (every wave has finished reading
//                 K-tile t: its buffer is free) ; from there on LDS-DMA pieces of K-tile t+2, one per DPM MFMAs
//   phase A(t+1) = MFMAs (t, 1); pieces continue; at MFMA RB: vmcnt(pieces of t+2 issued so far) + barrier (K-tile t+1 landed); then the
//                 reads (t+1, 0), one per 2 MFMAs, and the remaining pieces.
template <int WB, int RB, int DPM>
__global__ __launch_bounds__(256, 1) void probe_asm3(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ C,
                                                    int M, int N, int K, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  walk(blockIdx.x, M / 256, tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256, n0 = (int64_t)tn * 256;
  const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
  unsigned voff[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) voff[p] = (unsigned)(((wave * 8 + p) * 8 + srow) * K * 2 + schunk * 16 + 3072 - 1024 * (p & 3));
  const char* abase = (const char*)(A + m0 * K) - 3072;
  const char* bbase = (const char*)(B + n0 * K) - 3072;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned piece_dst = lds0 + wave * 8192;
  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  const unsigned coff0 = ((fq ^ (frow & 7)) << 4);
  unsigned a_ad[2][2], b_ad[2][2];
#pragma unroll
  for (int bu = 0; bu < 2; ++bu)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_ad[bu][ks] = lds0 + bu * 65536 + (wr * 128 + frow) * 128 + (coff0 ^ (ks * 64));
      b_ad[bu][ks] = lds0 + bu * 65536 + 32768 + (wc * 128 + frow) * 128 + (coff0 ^ (ks * 64));
    }
  const int nk = K / 64;   // even
  bf16x8_t fa[2][8], fb[2][8];
  auto dma = [&](auto pc, int buf, const char* ab, const char* bb) {
    constexpr int p = decltype(pc)::value;
    if constexpr ((p & 3) == 0) set_m0(piece_dst + (p & 4) * 1024 + (p >= 8 ? 32768 : 0) + buf * 65536);
    glds_s<1024 * (p & 3)>(voff[p & 7], p >= 8 ? bb : ab);
  };
  auto rd = [&](auto idx, int set, unsigned aaddr, unsigned baddr) {
    constexpr int r = decltype(idx)::value;
    if constexpr (r < 8) lds_rd<r * 2048>(fb[set][r], baddr); else lds_rd<(r - 8) * 2048>(fa[set][r - 8], aaddr);
  };
  sfor<16>([&](auto pc) { dma(pc, 0, abase, bbase); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  sfor<16>([&](auto r) { rd(r, 0, a_ad[0][0], b_ad[0][0]); });
  sfor<16>([&](auto pc) { dma(pc, 1, abase + 128, bbase + 128); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  const char* an = abase + 256;
  const char* bn = bbase + 256;
  const char* alast = abase + (int64_t)(nk - 1) * 128;
  const char* blast = bbase + (int64_t)(nk - 1) * 128;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(), wsum = 0;
  constexpr int NB_PIECES = (64 - WB + DPM - 1) / DPM;        // pieces issued in phase B (from MFMA WB on)
  constexpr int NA1 = (RB + DPM - 1) / DPM;                   // pieces issued in phase A before the landing barrier
  static_assert(NB_PIECES + NA1 <= 16, "too many pieces before the landing barrier");
#define PHASE_B(RA, RB_, DBUF)                                                                                \
  sfor<64>([&](auto mc) {                                                                                     \
    constexpr int mm = decltype(mc)::value, i = mm / 8, j = mm % 8;                                           \
    if constexpr (mm == WB) {                                                                                 \
      const unsigned long long w0 = __builtin_amdgcn_s_memtime();                                             \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
      __builtin_amdgcn_s_barrier();                                                                           \
      wsum += __builtin_amdgcn_s_memtime() - w0;                                                              \
    }                                                                                                         \
    mfma_a(acc[i][j], fa[0][i], fb[0][j]);                                                                    \
    if constexpr (mm % 2 == 1 && mm / 2 < 16) rd(std::integral_constant<int, mm / 2>{}, 1, RA, RB_);          \
    if constexpr (mm >= WB && (mm - WB) % DPM == DPM - 1) dma(std::integral_constant<int, (mm - WB) / DPM>{}, DBUF, an, bn); \
  });
#define PHASE_A(RA, RB_, DBUF)                                                                                \
  sfor<64>([&](auto mc) {                                                                                     \
    constexpr int mm = decltype(mc)::value, i = mm / 8, j = mm % 8;                                           \
    if constexpr (mm == RB) {                                                                                 \
      const unsigned long long w0 = __builtin_amdgcn_s_memtime();                                             \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_PIECES + NA1) : "memory");                                  \
      __builtin_amdgcn_s_barrier();                                                                           \
      wsum += __builtin_amdgcn_s_memtime() - w0;                                                              \
    }                                                                                                         \
    mfma_a(acc[i][j], fa[1][i], fb[1][j]);                                                                    \
    if constexpr (mm >= RB && (mm - RB) % 2 == 1 && (mm - RB) / 2 < 16) rd(std::integral_constant<int, (mm - RB) / 2>{}, 0, RA, RB_); \
    if constexpr (mm % DPM == DPM - 1 && NB_PIECES + mm / DPM < 16) dma(std::integral_constant<int, NB_PIECES + mm / DPM>{}, DBUF, an, bn); \
  });
  for (int t = 0; t < nk; t += 2) {
    PHASE_B(a_ad[0][1], b_ad[0][1], 0)
    PHASE_A(a_ad[1][0], b_ad[1][0], 0)
    an = an + 128 < alast ? an + 128 : alast;
    bn = bn + 128 < blast ? bn + 128 : blast;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PHASE_B(a_ad[1][1], b_ad[1][1], 1)
    PHASE_A(a_ad[0][0], b_ad[0][0], 1)
    an = an + 128 < alast ? an + 128 : alast;
    bn = bn + 128 < blast ? bn + 128 : blast;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#undef PHASE_A
#undef PHASE_B
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_clk[0] = __builtin_amdgcn_s_memtime() - c0;
    g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    g_clk[2] = wsum;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(m0 + wr * 128 + i * 16 + fq * 4 + r) * N + n0 + wc * 128 + j * 16 + frow] = acc[i][j][r];
}

static unsigned short f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

typedef void (*kern_t)(const unsigned short*, const unsigned short*, float*, int, int, int, int);
static void run(kern_t kern, const char* name, const unsigned short* dA, const unsigned short* dB, float* dC, int M, int N, int K, const std::vector<unsigned short>& hA,
                const std::vector<unsigned short>& hB, bool check) {
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  dim3 grid((M / 256) * (N / 256));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 131072, 0, dA, dB, dC, M, N, K, N / 256);
  float best = 1e30f;
  for (int rnd = 0; rnd < 5; ++rnd) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, grid, dim3(256), 131072, 0, dA, dB, dC, M, N, K, N / 256);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms / 10 < best) best = ms / 10;
  }
  double maxerr = 0;
  if (check) {
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    for (int s = 0; s < 64; ++s) {
      const int m = (s * 7919) % M, n = (s * 104729) % N;
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
      const double e = fabs(ref - hC[(size_t)m * N + n]);
      if (e > maxerr) maxerr = e;
    }
  }
  unsigned long long clk[4] = {0, 0, 0, 0};
  (void)hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
  const unsigned long long zero[4] = {0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_clk), zero, sizeof(zero));
  const double tiles_per_cu = (double)(M / 256) * (N / 256) / 256.0;
  printf("%-46s %8.4f ms  %7.1f TFLOP/s  %6.3f us per K-step%s\n", name, best, 2.0 * M * N * K / best / 1e9, best * 1e3 / (tiles_per_cu * (K / 64)),
         check ? (maxerr < 0.05 ? "  (ok)" : "  *** MISMATCH vs host reference ***") : "");
  if (clk[1]) printf("      workgroup 0 K loop: %llu cycles = %.0f per K-tile (2048 = MFMA-bound), %.0f of them in waits + barrier; clock %.3f GHz\n", clk[0],
                     (double)clk[0] / (K / 64), (double)clk[2] / (K / 64), (double)clk[0] / ((double)clk[1] * 10.0) );
}

int main(int argc, char** argv) {
  const int M = 8192, N = 8192, K = 8192;
  std::vector<unsigned short> hA((size_t)M * K), hB((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = f2bf((rand() / (float)RAND_MAX - 0.5f));
  for (auto& v : hB) v = f2bf((rand() / (float)RAND_MAX - 0.5f));
  unsigned short *dA, *dB; float* dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  {   // the shipped kernel on the same operands, same process (plain f32 store)
    void* h = dlopen(argc > 1 ? argv[1] : "algonauts-2025_amd/tribe_hip/libtribe_hip.so", RTLD_NOW);
    if (h) {
      typedef int (*gemm_fn)(const tribe_gemm_desc*, void*);
      gemm_fn fn = (gemm_fn)dlsym(h, "tribe_gemm_bf16");
      tribe_gemm_desc d;
      memset(&d, 0, sizeof(d));
      d.M = M; d.N = N; d.K = K; d.batch1 = 1; d.batch0 = 1; d.A = dA; d.lda = K; d.B = dB; d.ldb = K; d.C = dC; d.ldc = N; d.c_dtype = TRIBE_F32; d.alpha = 1.0f;
      d.tile_hint = 2;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) fn(&d, 0);
      float best = 1e30f;
      for (int rnd = 0; rnd < 5; ++rnd) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) fn(&d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
      }
      printf("%-46s %8.4f ms  %7.1f TFLOP/s  %6.3f us per K-step\n", "SHIPPED kernel (8 waves, 2 per SIMD), same box", best, 2.0 * M * N * K / best / 1e9,
             best * 1e3 / (4.0 * (K / 64)));
    } else printf("(shipped library not found: %s)\n", dlerror());
  }
  printf("one-wave-per-SIMD 256x256x64 GEMM tile probe, 8192^3 bf16, random operands (shipped two-waves-per-SIMD kernel: ~1.42-1.49 us per K-step)\n");
  run(probe<0>, "4 waves, LDS-DMA in a burst before the MFMAs", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe<1>, "4 waves, one LDS-DMA piece per 8 MFMAs", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe<2>, "4 waves, NO LDS-DMA (reads + MFMAs only)", dA, dB, dC, M, N, K, hA, hB, false);
  run(probe_pipe<1>, "4 waves, pipelined across K-tiles, DMA 1 per 4 MFMAs", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_pipe<0>, "4 waves, pipelined, DMA placement left to hipcc", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm<3, 2>, "4 waves, asm-ordered: read / 3 MFMAs, DMA / 2 MFMAs", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm<3, 4>, "4 waves, asm-ordered: read / 3 MFMAs, DMA / 4 MFMAs", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm<4, 4>, "4 waves, asm-ordered: read / 4 MFMAs, DMA / 4 MFMAs", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm<2, 4>, "4 waves, asm-ordered: read / 2 MFMAs, DMA / 4 MFMAs", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<2, 4, 0>, "asm2 (scalar-base DMA, shared M0): rd/2, DMA/4", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<2, 2, 0>, "asm2: rd/2, DMA/2", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<1, 3, 16>, "asm2: rd/1, DMA/3 from MFMA 16", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<2, 1, 32>, "asm2: rd/2, DMA/1 from MFMA 32", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<3, 4, 0>, "asm2: rd/3, DMA/4", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<3, 4, 0, 1>, "asm2: rd/3, NO DMA (reads + MFMAs + barrier)", dA, dB, dC, M, N, K, hA, hB, false);
  run(probe_asm2<3, 4, 0, 0, 1>, "asm2: rd/3, DMA/4 + XCD-aware tile walk", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<4, 4, 0, 0, 1>, "asm2: rd/4, DMA/4 + XCD-aware tile walk", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<3, 4, 0, 1, 1>, "asm2: rd/3, NO DMA + XCD-aware tile walk", dA, dB, dC, M, N, K, hA, hB, false);
  run(probe_asm3<40, 16, 4>, "asm3: 2 barriers (free @40, landed @16), DMA/4", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm3<40, 24, 4>, "asm3: free @40, landed @24, DMA/4", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm3<36, 16, 6>, "asm3: free @36, landed @16, DMA/6", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm3<48, 16, 4>, "asm3: free @48, landed @16, DMA/4", dA, dB, dC, M, N, K, hA, hB, true);
  run(probe_asm2<4, 4, 0>, "asm2: rd/4, DMA/4", dA, dB, dC, M, N, K, hA, hB, true);
  return 0;
}
