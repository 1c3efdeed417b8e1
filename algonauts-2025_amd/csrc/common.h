// Shared helpers for the TRIBE gfx950 kernels (device + host side of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tribe_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator fragment
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;

#define TRIBE_WAVE 64

// ---- bf16 <-> f32 ------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// ---- wave / block reductions ---------------------------------------------------
// streaming (read-once) 16-byte load: keeps a one-pass reduction from evicting what L2 / MALL hold for the GEMMs
typedef float tribe_f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_nt_f4(const void* p) {
  const tribe_f32x4_nt v = __builtin_nontemporal_load((const tribe_f32x4_nt*)p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- host-side error plumbing ----------------------------------------------------
void tribe_set_error(const char* fmt, ...);

#define TRIBE_REQUIRE(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      tribe_set_error(__VA_ARGS__);   \
      return -1;                      \
    }                                 \
  } while (0)

#define TRIBE_LAUNCH_CHECK()                                           \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      tribe_set_error("HIP launch failed: %s", hipGetErrorString(e__)); \
      return (int)e__;                                                 \
    }                                                                  \
  } while (0)

// in-library HIP-event profile (gemm.hip; bench.py brackets its timed region with tribe_prof_begin / tribe_prof_end): other launchers
// of the path (fused attention) take a slot for their role the same way the GEMM launcher does
int tribe_internal_prof_before(int role, double flops, hipStream_t s);
void tribe_internal_prof_after(int slot, hipStream_t s);

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
