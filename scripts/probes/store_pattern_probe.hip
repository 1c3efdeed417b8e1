// Probe for the GEMM epilogues ("a store costs a wave ~400 cycles whatever its width"): what does a wave pay per
// global_store_dwordx4 as a function of the ADDRESS PATTERN of its 64 lanes?  Every pattern writes the same 1 KiB per instruction into an
// f32 [65536, 3072] matrix, tile by tile as the 256 x 256 kernel's epilogue does (a workgroup per tile, a wave per 128 x 64 (8 waves) or
// 128 x 128 (4 waves) block), with nothing else in the kernel:
//   P16x64   16 rows x 64 B per instruction  -- what the epilogue issues today (one 16 x 16 f32 sub-tile; TACC layout)
//   P8x128    8 rows x 128 B                 -- whole 128-byte lines (two sub-tiles side by side, rows split 0..7 / 8..15)
//   P4x256    4 rows x 256 B
//   P2x512    2 rows x 512 B (4 waves only: the wave block is 128 columns wide)
// and each with WAVES = 8 (two per SIMD) and 4 (one per SIMD).  Reported: GB/s over the launch and shader cycles per store instruction as
// seen by wave 0 of workgroup 0 (s_memtime around its stores, s_waitcnt vmcnt(0) included).
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/store_pattern_probe.hip -o ab_tmp/store_pattern_probe && ab_tmp/store_pattern_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ unsigned long long g_clk[2];
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ROWS x (1024 / ROWS) bytes per instruction; a wave block is BR rows x BC columns (f32)
template <int ROWS, int WAVES, int NT>
__global__ __launch_bounds__(WAVES * 64) void probe(float* __restrict__ C, int64_t ldc, int tiles_n) {
  constexpr int LPR = 64 / ROWS;                       // lanes per row
  constexpr int BR = 128, BC = WAVES == 8 ? 64 : 128;  // wave block
  constexpr int CPI = LPR * 4;                         // columns per instruction
  static_assert(CPI <= BC, "an instruction must fit the wave block");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = WAVES == 8 ? wave >> 2 : wave >> 1, wc = WAVES == 8 ? wave & 3 : wave & 1;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  float* base = C + ((int64_t)tm * 256 + wr * BR + lane / LPR) * ldc + (int64_t)tn * 256 + wc * BC + (lane % LPR) * 4;
  const f32x4_t v = {(float)lane, (float)wave, (float)blockIdx.x, 1.0f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int n = 0;
#pragma unroll 1
  for (int r = 0; r < BR; r += ROWS)
#pragma unroll
    for (int c = 0; c < BC; c += CPI) {
      if (NT) __builtin_nontemporal_store(v, (f32x4_t*)(base + (int64_t)r * ldc + c));
      else *(f32x4_t*)(base + (int64_t)r * ldc + c) = v;
      ++n;
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_amdgcn_s_memtime() - t0; g_clk[1] = (unsigned long long)n; }
}

template <int ROWS, int WAVES, int NT>
static void run(const char* name, float* C, int64_t M, int64_t N) {
  const int tiles_m = (int)(M / 256), tiles_n = (int)(N / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<ROWS, WAVES, NT>), dim3(tiles_m * tiles_n), dim3(WAVES * 64), 0, 0, C, N, tiles_n);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<ROWS, WAVES, NT>), dim3(tiles_m * tiles_n), dim3(WAVES * 64), 0, 0, C, N, tiles_n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  unsigned long long clk[2];
  hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
  printf("%-10s %d waves%s  %8.1f us  %7.0f GB/s   wave 0 of workgroup 0: %6.0f cycles per store (%llu stores)\n", name, WAVES, NT ? " nt" : "   ",
         ms * 1e3, (double)M * N * 4 / (ms * 1e-3) / 1e9, (double)clk[0] / (double)clk[1], clk[1]);
  fflush(stdout);
}

int main() {
  const int64_t M = 65536, N = 3072;
  float* C = nullptr;
  if (hipMalloc(&C, M * N * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  printf("f32 [%lld, %lld] written tile by tile (256 x 256 per workgroup), 1 KiB per store instruction, nothing else in the kernel\n", (long long)M, (long long)N);
  run<16, 8, 0>("P16x64", C, M, N);
  run<8, 8, 0>("P8x128", C, M, N);
  run<4, 8, 0>("P4x256", C, M, N);
  run<16, 4, 0>("P16x64", C, M, N);
  run<8, 4, 0>("P8x128", C, M, N);
  run<4, 4, 0>("P4x256", C, M, N);
  run<2, 4, 0>("P2x512", C, M, N);
  run<16, 8, 1>("P16x64", C, M, N);
  run<8, 8, 1>("P8x128", C, M, N);
  // fewer workgroups than CUs: is the per-store cost a per-CU limit or the chip's HBM write rate shared by 256 CUs?
  printf("-- partial grids (12 tiles per row of tiles): 1 / 4 / 16 / 64 rows of tiles = 12 / 48 / 192 / 768 workgroups\n");
  for (int64_t rows : {256, 1024, 4096, 16384}) {
    run<16, 8, 0>("P16x64", C, rows, N);
    run<16, 4, 0>("P16x64", C, rows, N);
  }
  hipFree(C);
  return 0;
}
