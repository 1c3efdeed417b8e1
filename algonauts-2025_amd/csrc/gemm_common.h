// Pieces shared by the two MFMA GEMM tile configurations (gemm.hip): the operator epilogue and the
// XCD-aware tile remap.
#pragma once
#include "common.h"
#include <utility>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// bijective "each XCD gets a contiguous chunk of tiles" remap (guide T1); changes speed only
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// tile id -> (tm, tn): the XCD remap first gives every XCD one contiguous run of ids; inside a run ids walk
// bands of GM = 4 tile rows column by column, so the ~32 workgroups an XCD runs at once form a 4 x 8 patch
// that shares 4 A panels and 8 B panels in that XCD's L2 (instead of 1 + 32 with a plain row-major walk).
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
#ifndef TRIBE_GEMM_BAND_ROWS
#define TRIBE_GEMM_BAND_ROWS 4   // scripts/gemm_band_experiment.py builds 2 / 8 / 16 for the L2-reuse experiment of DESIGN 4.1
#endif
  constexpr int GM = TRIBE_GEMM_BAND_ROWS;
  const int tile = xcd_remap(bid, tiles_m * tiles_n);
  const int band = tile / (GM * tiles_n);
  const int first_m = band * GM;
  const int gm = (tiles_m - first_m < GM) ? tiles_m - first_m : GM;
  const int r = tile - band * GM * tiles_n;
  tn = r / gm;
  tm = first_m + (r - tn * gm);
}

// GELU(v) = v * Phi(v) for TWO values at once, used where the result is rounded to bf16 anyway.  No transcendental:
// Phi(v) - 0.5 = v * Q(v^2) on |v| <= 4.4 (degree-8 minimax-style fit, |Phi error| <= 1.9e-5), argument clamped beyond
// (Phi -> 0 / 1 within 1e-5).  |GELU error| <= 8.3e-5 absolute over all v (checked in f32 on a 2M-point grid) -- two
// orders below the bf16 rounding of the values it feeds.  Every step is a packed-f32 operation (v_pk_mul_f32 /
// v_pk_fma_f32: two values per issue): 12 issues per pair against ~44 for two erfc-polynomial evaluations with
// v_exp_f32 + v_rcp_f32 at quarter rate (the round-1 epilogue): FF1 3.93 -> 3.80 ms once the epilogue stopped waiting
// on memory.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ f32x2_t gelu_poly2(f32x2_t v) {
  f32x2_t c;
  c.x = __builtin_amdgcn_fmed3f(v.x, -4.4f, 4.4f);
  c.y = __builtin_amdgcn_fmed3f(v.y, -4.4f, 4.4f);
  const f32x2_t t = c * c;
  f32x2_t q = 4.4347532590638394e-11f;
  q = q * t + -4.493755145773548e-09f;
  q = q * t + 2.0033526482166053e-07f;
  q = q * t + -5.224721007834887e-06f;
  q = q * t + 8.98004655027762e-05f;
  q = q * t + -0.0010902436915785074f;
  q = q * t + 0.009767354466021061f;
  q = q * t + -0.06628739088773727f;
  q = q * t + 0.3988831341266632f;
  const f32x2_t phi = c * q + 0.5f;
  return v * phi;
}

// d/dv [v * Phi(v)] = Phi(v) + v * phi(v)
__device__ __forceinline__ float gelu_grad(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
  return cdf + v * pdf;
}
// The same derivative for TWO values, for factors that multiply a gradient about to be rounded to bf16: Phi from the polynomial of
// gelu_poly2 (|error| <= 1.9e-5), phi(v) = exp2(-0.5 log2(e) v^2) / sqrt(2 pi) with one v_exp_f32 -- against erff + expf per value
// (|difference to gelu_grad| of the order of the polynomial's 1.9e-5; tests/test_gpu_training.py::test_gelu_backward_epilogue bounds it at 1e-4
// against torch's exact derivative through the epilogue).
__device__ __forceinline__ f32x2_t gelu_grad2(f32x2_t v) {
  f32x2_t c;
  c.x = __builtin_amdgcn_fmed3f(v.x, -4.4f, 4.4f);
  c.y = __builtin_amdgcn_fmed3f(v.y, -4.4f, 4.4f);
  const f32x2_t t = c * c;
  f32x2_t q = 4.4347532590638394e-11f;
  q = q * t + -4.493755145773548e-09f;
  q = q * t + 2.0033526482166053e-07f;
  q = q * t + -5.224721007834887e-06f;
  q = q * t + 8.98004655027762e-05f;
  q = q * t + -0.0010902436915785074f;
  q = q * t + 0.009767354466021061f;
  q = q * t + -0.06628739088773727f;
  q = q * t + 0.3988831341266632f;
  const f32x2_t cdf = c * q + 0.5f;
  const f32x2_t e = v * v * -0.72134752044448170368f;   // -0.5 * log2(e) * v^2
  f32x2_t pdf;
  pdf.x = 0.3989422804014327f * __builtin_amdgcn_exp2f(e.x);
  pdf.y = 0.3989422804014327f * __builtin_amdgcn_exp2f(e.y);
  return cdf + v * pdf;
}
__device__ __forceinline__ void gelu_grad_mul4(float (&v)[4], const u16x4_t pre) {
  const f32x2_t lo = gelu_grad2(f32x2_t{bf16_to_f32(pre[0]), bf16_to_f32(pre[1])}), hi = gelu_grad2(f32x2_t{bf16_to_f32(pre[2]), bf16_to_f32(pre[3])});
  v[0] *= lo.x; v[1] *= lo.y; v[2] *= hi.x; v[3] *= hi.y;
}

__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.4426950408889634f));
}

__device__ __forceinline__ float sigmoid_f(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.4426950408889634f));
}

// quad-lane exchanges as DPP moves (no LDS traffic): lane i <- lane i^1 / i^2 within each group of 4
__device__ __forceinline__ float quad_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

struct EpiCtx {
  const float* bias;   // already offset by the (gathered) batch
  const float* res;    // already offset by the batch
  char* C;             // base of C (bytes), NOT offset
  int64_t c_off;       // element offset of this batch
  bool vec;            // every operand the epilogue touches allows 4-wide vector access
};

__device__ __forceinline__ EpiCtx make_epi_ctx(const tribe_gemm_desc& g, int64_t b1, int64_t b0, int64_t b1g) {
  EpiCtx c;
  const int64_t bb = g.gather_bias ? b1g : b1;
  c.bias = g.bias ? g.bias + bb * g.sBias1 + b0 * g.sBias0 : nullptr;
  c.res = g.res ? g.res + b1 * g.sRes1 + b0 * g.sRes0 : nullptr;
  c.C = (char*)g.C;
  c.c_off = b1 * g.sC1 + b0 * g.sC0;
  const int esz = (g.c_dtype == TRIBE_BF16) ? 2 : 4;
  bool v = ((g.ldc & 3) == 0) && ((c.c_off & 3) == 0) && ((((uintptr_t)g.C) & (uintptr_t)(4 * esz - 1)) == 0);
  if (g.bias_mode == TRIBE_BIAS_COL) v = v && ((((uintptr_t)c.bias) & 15) == 0);
  if (c.res) v = v && ((g.ldres & 3) == 0) && ((((uintptr_t)c.res) & 15) == 0);
  if (g.res_scale) v = v && ((((uintptr_t)g.res_scale) & 15) == 0);
  if (g.rowadd) v = v && ((g.ld_rowadd & 3) == 0) && ((((uintptr_t)g.rowadd) & 15) == 0);
  if (g.gadd) v = v && ((g.ld_gadd & 3) == 0) && ((((uintptr_t)g.gadd) & 15) == 0);
  if (g.aux) v = v && ((g.ld_aux & 3) == 0) && ((((uintptr_t)g.aux) & 7) == 0);
  c.vec = v;
  return c;
}

// One 16x16 accumulator tile (v_mfma_f32_16x16x32 C/D map: acc[r] = D[4*(lane>>4) + r][lane & 15]) ->
// epi(...) -> C.  A 4x4 transpose inside each quad of lanes turns "4 rows x 1 column" per lane into
// "1 row x 4 consecutive columns", so every lane issues ONE 16-byte (f32) / 8-byte (bf16) store and
// reads its residual / bias operands as float4.
// 4x4 transpose inside each quad of lanes: acc[r] = D[4*(lane>>4) + r][lane & 15] (MFMA C/D map) becomes
// "row 4*(lane>>4) + (lane & 3), columns 4*((lane & 15) >> 2) .. +3" -- 4 consecutive columns of ONE row per lane.
__device__ __forceinline__ void quad_transpose(f32x4_t acc, float alpha, int lane, float (&v)[4]) {
  const int a = lane & 3;
  float v0 = acc[0] * alpha, v1 = acc[1] * alpha, v2 = acc[2] * alpha, v3 = acc[3] * alpha;
  const bool b0 = a & 1, b1 = a & 2;
  const float t0 = quad_xor1(b0 ? v0 : v1), t1 = quad_xor1(b0 ? v2 : v3);
  if (b0) { v0 = t0; v2 = t1; } else { v1 = t0; v3 = t1; }
  const float u0 = quad_xor2(b1 ? v0 : v2), u1 = quad_xor2(b1 ? v1 : v3);
  if (b1) { v0 = u0; v1 = u1; } else { v2 = u0; v3 = u1; }
  v[0] = v0; v[1] = v1; v[2] = v2; v[3] = v3;
}

// epilogue of 4 consecutive columns n..n+3 of row m (values already scaled by alpha): bias, activation, residual,
// row / gathered adds, store.
// EXT = 0 compiles only the operators of the encode hot path (bias, GELU, scaled residual, row / gathered adds);
// EXT = 1 adds the rarely used ones (SwiGLU / GLU / SiLU, GELU' with the saved pre-activation, pre-activation store).
// Keeping them out of the hot kernels keeps their code (and SGPR pressure) small.
template <int OUT_BF16, int EXT>
__device__ __forceinline__ void epilogue_row4(const tribe_gemm_desc& g, const EpiCtx& c, float (&v)[4], int64_t m, int64_t n) {
  if (m >= g.M || n >= g.N) return;
  if (g.bias_mode == TRIBE_BIAS_ROW) {
    const float b = c.bias[m];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += b;
  }
  const float* rowadd = g.rowadd ? g.rowadd + (m % g.rowadd_period) * g.ld_rowadd : nullptr;
  const float* gadd = g.gadd ? g.gadd + g.gadd_index[m / g.gadd_div] * g.ld_gadd : nullptr;
  const int64_t idx = c.c_off + m * g.ldc + n;
  if (c.vec && n + 4 <= g.N) {
    if (g.bias_mode == TRIBE_BIAS_COL) {
      const float4 b = *(const float4*)(c.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (EXT && (g.act == TRIBE_ACT_SWIGLU || g.act == TRIBE_ACT_GLU)) {
      // column pairs -> two outputs at columns n/2, n/2 + 1 of a C that is N/2 wide:
      // SwiGLU silu(gate) * up (LlamaMLP, modeling_llama.py:177); GLU a * sigmoid(b) (nn.GLU in the conformer conv module)
      const bool glu = g.act == TRIBE_ACT_GLU;
      const float o0 = glu ? v[0] * sigmoid_f(v[1]) : silu_f(v[0]) * v[1];
      const float o1 = glu ? v[2] * sigmoid_f(v[3]) : silu_f(v[2]) * v[3];
      const int64_t oidx = c.c_off + m * g.ldc + (n >> 1);
      if (OUT_BF16) {
        const unsigned int pk = (unsigned int)f32_to_bf16(o0) | ((unsigned int)f32_to_bf16(o1) << 16);
        *(unsigned int*)((unsigned short*)c.C + oidx) = pk;
      } else {
        *(float2*)((float*)c.C + oidx) = make_float2(o0, o1);
      }
      return;
    }
    if (g.act == TRIBE_ACT_GELU) {
      if (EXT && g.aux) {  // training forward: keep the pre-activation for the backward pass
        u16x4_t pre;
#pragma unroll
        for (int k = 0; k < 4; ++k) pre[k] = f32_to_bf16(v[k]);
        *(u16x4_t*)((unsigned short*)g.aux + c.c_off + m * g.ld_aux + n) = pre;
      }
      if (OUT_BF16) {
        const f32x2_t lo = gelu_poly2(f32x2_t{v[0], v[1]}), hi = gelu_poly2(f32x2_t{v[2], v[3]});
        v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gelu_erf(v[k]);
      }
    } else if (EXT && g.act == TRIBE_ACT_GELU_BWD) {
      const u16x4_t pre = *(const u16x4_t*)((const unsigned short*)g.aux + c.c_off + m * g.ld_aux + n);
      gelu_grad_mul4(v, pre);
    } else if (EXT && g.act == TRIBE_ACT_SILU) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = silu_f(v[k]);
    } else if (EXT && g.act == TRIBE_ACT_EXP2) {   // attention backward: P = exp2(scaled score - lse2[row]) (the row bias)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_exp2f(v[k]);
    } else if (EXT && g.act == TRIBE_ACT_MUL_AUX) {   // attention backward: dS = (scale dP - scale D[row]) * P
      const u16x4_t p = *(const u16x4_t*)((const unsigned short*)g.aux + c.c_off + m * g.ld_aux + n);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= bf16_to_f32(p[k]);
    }
    if (c.res) {
      const float4 r = *(const float4*)(c.res + m * g.ldres + n);
      if (g.res_scale) {
        const float4 s = *(const float4*)(g.res_scale + n);
        v[0] += r.x * s.x; v[1] += r.y * s.y; v[2] += r.z * s.z; v[3] += r.w * s.w;
      } else {
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
    }
    if (rowadd) {
      const float4 r = *(const float4*)(rowadd + n);
      v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    if (gadd) {
      const float4 r = *(const float4*)(gadd + n);
      v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    }
    if (OUT_BF16) {
      u16x4_t o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = f32_to_bf16(v[k]);
      *(u16x4_t*)((unsigned short*)c.C + idx) = o;
    } else {
      *(float4*)((float*)c.C + idx) = make_float4(v[0], v[1], v[2], v[3]);
    }
    return;
  }
  if (EXT && (g.act == TRIBE_ACT_SWIGLU || g.act == TRIBE_ACT_GLU)) {
#pragma unroll
    for (int k = 0; k < 4; k += 2) {
      if (n + k + 1 >= g.N) break;
      float gate = v[k], up = v[k + 1];
      if (g.bias_mode == TRIBE_BIAS_COL) { gate += c.bias[n + k]; up += c.bias[n + k + 1]; }
      const float o = (g.act == TRIBE_ACT_GLU) ? gate * sigmoid_f(up) : silu_f(gate) * up;
      const int64_t oidx = c.c_off + m * g.ldc + ((n + k) >> 1);
      if (OUT_BF16) ((unsigned short*)c.C)[oidx] = f32_to_bf16(o);
      else ((float*)c.C)[oidx] = o;
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (n + k >= g.N) break;
    float x = v[k];
    if (g.bias_mode == TRIBE_BIAS_COL) x += c.bias[n + k];
    if (g.act == TRIBE_ACT_GELU) {
      if (EXT && g.aux) ((unsigned short*)g.aux)[c.c_off + m * g.ld_aux + n + k] = f32_to_bf16(x);
      x = OUT_BF16 ? gelu_poly2(f32x2_t{x, x}).x : gelu_erf(x);   // same approximation as the vector paths
    } else if (EXT && g.act == TRIBE_ACT_GELU_BWD) {
      x *= gelu_grad(bf16_to_f32(((const unsigned short*)g.aux)[c.c_off + m * g.ld_aux + n + k]));
    } else if (EXT && g.act == TRIBE_ACT_SILU) x = silu_f(x);
    else if (EXT && g.act == TRIBE_ACT_EXP2) x = __builtin_amdgcn_exp2f(x);
    else if (EXT && g.act == TRIBE_ACT_MUL_AUX) x *= bf16_to_f32(((const unsigned short*)g.aux)[c.c_off + m * g.ld_aux + n + k]);
    if (c.res) {
      const float r = c.res[m * g.ldres + n + k];
      x += g.res_scale ? r * g.res_scale[n + k] : r;
    }
    if (rowadd) x += rowadd[n + k];
    if (gadd) x += gadd[n + k];
    if (OUT_BF16) ((unsigned short*)c.C)[idx + k] = f32_to_bf16(x);
    else ((float*)c.C)[idx + k] = x;
  }
}

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The epilogues index the
// accumulator array with these constants, so the accumulators stay in registers whatever the unroller decides
// ("#pragma unroll" gives up above its size threshold; the accumulators then move to scratch and the epilogue
// becomes a loop over scratch loads -- that cost 528 B/lane of scratch and 20 % of the 256 x 256 kernel once).
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Whole-wave epilogue for INTERIOR tiles (every operand 16-byte accessible, no bounds): the accumulators of a wave's
// NI x 4 sub-tiles -> C.  The generic path below re-loads its operands per sub-tile; every one of those loads ends in an
// s_waitcnt vmcnt(0) that also waits for the STORES of the previous sub-tile (vmcnt counts loads and stores in order), so
// a 256 x 256 tile paid ~32 serial store-drain + load round trips (voxel head: ~50 us per tile, against a 63 us K loop).
// Here nothing inside the sub-tile loop waits on memory:
//   * operands that depend on the column only (column bias, residual scale) or the row only (row bias) are loaded once;
//   * the residual tile is fetched with LDS-DMA into the (now idle) GEMM staging buffers, 16 sub-tiles per round,
//     lane-linear so that each lane later reads back exactly its own 16 bytes -- no VGPRs held, one wait per round.
// Supported: bias none / column / row; act none / GELU (+ SwiGLU / GLU / SiLU in the EXT kernels); residual (+ scale); rows past M
// masked; in the EXT kernels also the training operators (pre-activation store, GELU' from the saved pre-activation).
// The periodic row add (pos_embed) takes the residual's place when there is no residual; gathered adds, a row add together
// with a residual, and tiles that cross N take the generic path.
// NI x NJ = the wave's accumulator grid of 16 x 16 sub-tiles; a round covers RI = min(NI, 4) sub-tile rows x NJ columns (<= 16 sub-tiles,
// 1 KiB of LDS each for the tile-shaped addend).  Row sums of squares go to slot nw / (16 NJ): one slot per wave column group.
// TACC = 1 (round 3; the 8-wave 256-row kernel, NT form): the K loop multiplied with the MFMA operands SWAPPED and the B fragment row quads
// permuted (0, 2, 1, 3), so a 16 x 16 accumulator holds its sub-tile transposed: register r of lane l is C[row l & 15][column 4 sigma(l >> 4) + r].
// A lane then owns four consecutive columns of one row WITHOUT the quad transpose (16 vector instructions per sub-tile), lanes l and l + 32
// own adjacent column quads, and bf16 outputs are written 16 bytes per lane: two v_permlane32_swap per PAIR of sub-tiles (j, j + 1) give the
// lower half-wave 8 consecutive columns of sub-tile j and the upper half those of j + 1 -- half the store instructions (an epilogue's
// stores cost a wave ~400 cycles each whatever their width).
template <int OUT_BF16, int NI, int NJ, int EXT, int TACC = 0>
__device__ __forceinline__ void epilogue_fast(const tribe_gemm_desc& g, const EpiCtx& c, f32x4_t (&acc)[NI][NJ], int64_t mw, int64_t nw, int lane,
                                              char* lds_wave) {
  constexpr int RI = NI >= 4 ? 4 : NI;
  static_assert(NI % RI == 0 && NJ >= 1 && NJ <= 4, "accumulator grid");
  const int tq = lane >> 4;
  const int64_t row0 = TACC ? mw + (lane & 15) : mw + ((lane >> 4) << 2) + (lane & 3);                       // + 16 i
  const int64_t col0 = TACC ? nw + 4 * ((tq & 1) * 2 + (tq >> 1)) : nw + (((lane & 15) >> 2) << 2);          // + 16 j
  const int64_t colp = nw + 16 * (lane >> 5) + 8 * (tq & 1);   // TACC: first of the 8 columns a paired bf16 store writes (+ 16 j, j even)
  const bool bias_col = g.bias_mode == TRIBE_BIAS_COL, bias_row = g.bias_mode == TRIBE_BIAS_ROW;
  const bool res_scaled = c.res && g.res_scale;
  const bool pair_act = EXT && (g.act == TRIBE_ACT_SWIGLU || g.act == TRIBE_ACT_GLU);
  float4 bcol[NJ], rsc[NJ];
  static_for<NJ>([&](auto jt) {
    constexpr int j = decltype(jt)::value;
    if (bias_col) bcol[j] = *(const float4*)(c.bias + col0 + j * 16);
    if (res_scaled) rsc[j] = *(const float4*)(g.res_scale + col0 + j * 16);
  });
  constexpr int ROUNDS = NI / RI;
  static_for<ROUNDS>([&](auto rt) {
    constexpr int round = decltype(rt)::value;
    float brow[RI], srow[RI], ssq[RI];
    static_for<RI>([&](auto it) { ssq[decltype(it)::value] = 0.f; });
    if (bias_row) {
      static_for<RI>([&](auto it) {
        const int64_t r = row0 + (round * RI + decltype(it)::value) * 16;
        brow[decltype(it)::value] = r < g.M ? c.bias[r] : 0.f;
      });
    }
    if (g.row_scale) {   // ScaleNorm of the A rows, applied to the product (see tribe_gemm_desc)
      static_for<RI>([&](auto it) {
        const int64_t r = row0 + (round * RI + decltype(it)::value) * 16;
        srow[decltype(it)::value] = r < g.M ? g.row_scale[r] : 0.f;
      });
    }
    // the tile-shaped addend: the residual, or (when there is none) the periodic row add -- pos_embed[t] of the projector
    const bool tile_add = c.res || g.rowadd;
    if (tile_add) {
      static_for<RI>([&](auto it) {
        constexpr int i4 = decltype(it)::value, i = round * RI + i4;
        const int64_t r = row0 + i * 16;
        if (r < g.M) {   // rows past M (bottom tile row): the lane neither fetches nor stores
          const float* src = c.res ? c.res + r * g.ldres + col0
                                   : g.rowadd + (int64_t)((unsigned)r % (unsigned)g.rowadd_period) * g.ld_rowadd + col0;
          static_for<NJ>([&](auto jt) {
            constexpr int j = decltype(jt)::value;
            __builtin_amdgcn_global_load_lds((gptr_t)(src + j * 16), (lptr_t)(lds_wave + (i4 * NJ + j) * 1024), 16, 0, 0);
          });
        }
      });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA's LDS writes are invisible to the compiler's own counters
    }
    // training dgrad through GELU: the saved bf16 pre-activations of the round, fetched together (32 VGPRs, EXT kernels only)
    u16x4_t pre[EXT ? RI * NJ : 1];
    const bool gelu_bwd = EXT && g.act == TRIBE_ACT_GELU_BWD, mul_aux = EXT && g.act == TRIBE_ACT_MUL_AUX;
    if (gelu_bwd || mul_aux) {
      static_for<RI * NJ>([&](auto st) {
        constexpr int s = decltype(st)::value, i = round * RI + s / NJ, j = s % NJ;
        const int64_t r = row0 + i * 16;
        pre[EXT ? s : 0] = r < g.M ? *(const u16x4_t*)((const unsigned short*)g.aux + c.c_off + r * g.ld_aux + col0 + j * 16) : u16x4_t{0, 0, 0, 0};
      });
    }
    uint2 pend = make_uint2(0u, 0u);   // TACC: the packed bf16 quad of a pair's first sub-tile
    static_for<RI * NJ>([&](auto st) {
      constexpr int s = decltype(st)::value, i4 = s / NJ, i = round * RI + i4, j = s % NJ;
      const int64_t row = row0 + i * 16;
      float v[4];
      if constexpr (TACC) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = acc[i][j][k] * g.alpha;
      } else {
        quad_transpose(acc[i][j], g.alpha, lane, v);   // all four lanes of a quad take part, including those whose row is past M
      }
      const bool live = row < g.M;
      if (!TACC && !live) return;   // (TACC: every lane stays for the half-wave exchanges of the paired stores; loads and stores are guarded)
      if (g.row_scale) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= srow[i4];
      }
      if (bias_row) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += brow[i4];
      }
      if (bias_col) { v[0] += bcol[j].x; v[1] += bcol[j].y; v[2] += bcol[j].z; v[3] += bcol[j].w; }
      if (pair_act) {
        // column pairs -> two outputs in a C that is N/2 wide: SwiGLU silu(gate) * up (modeling_llama.py:177), GLU a * sigmoid(b)
        const bool glu = g.act == TRIBE_ACT_GLU;
        const float o0 = glu ? v[0] * sigmoid_f(v[1]) : silu_f(v[0]) * v[1];
        const float o1 = glu ? v[2] * sigmoid_f(v[3]) : silu_f(v[2]) * v[3];
        const int64_t oidx = c.c_off + row * g.ldc + ((col0 + j * 16) >> 1);
        if (live) {
          if (OUT_BF16) *(unsigned int*)((unsigned short*)c.C + oidx) = (unsigned int)f32_to_bf16(o0) | ((unsigned int)f32_to_bf16(o1) << 16);
          else *(float2*)((float*)c.C + oidx) = make_float2(o0, o1);
        }
        return;
      }
      if (g.act == TRIBE_ACT_GELU) {
        if (EXT && g.aux) {   // training forward: keep the pre-activation for the backward pass (one more store, nothing to wait for)
          u16x4_t p;
#pragma unroll
          for (int k = 0; k < 4; ++k) p[k] = f32_to_bf16(v[k]);
          if (live) *(u16x4_t*)((unsigned short*)g.aux + c.c_off + row * g.ld_aux + col0 + j * 16) = p;
        }
        if (OUT_BF16) {
          const f32x2_t lo = gelu_poly2(f32x2_t{v[0], v[1]}), hi = gelu_poly2(f32x2_t{v[2], v[3]});
          v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = gelu_erf(v[k]);
        }
      } else if (gelu_bwd) {
        gelu_grad_mul4(v, pre[EXT ? s : 0]);
      } else if (mul_aux) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= bf16_to_f32(pre[EXT ? s : 0][k]);
      } else if (EXT && g.act == TRIBE_ACT_SILU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = silu_f(v[k]);
      } else if (EXT && g.act == TRIBE_ACT_EXP2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_exp2f(v[k]);
      }
      if (tile_add) {
        const float4 r = *(const float4*)(lds_wave + s * 1024 + lane * 16);
        if (res_scaled) { v[0] += r.x * rsc[j].x; v[1] += r.y * rsc[j].y; v[2] += r.z * rsc[j].z; v[3] += r.w * rsc[j].w; }
        else { v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
      }
      const int64_t idx = c.c_off + row * g.ldc + col0 + j * 16;
      // a bf16 row of 4 values: alone (8 bytes), or -- TACC, sub-tiles in pairs -- merged with the partner half-wave's into 16 bytes
      auto store_bf16 = [&](unsigned short* base, int64_t ld, int64_t boff) {
        u16x4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = f32_to_bf16(v[k]);
        constexpr bool paired = TACC && (NJ % 2 == 0 || j + 1 < NJ || (j & 1));
        if constexpr (!paired) {
          if (live) *(u16x4_t*)(base + boff + row * ld + col0 + j * 16) = o;
        } else {
          const uint2 pk = __builtin_bit_cast(uint2, o);
          if constexpr ((j & 1) == 0) {
            pend = pk;
          } else {
            const auto sx = __builtin_amdgcn_permlane32_swap(pend.x, pk.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pend.y, pk.y, false, false);
            if (live) *(uint4*)(base + boff + row * ld + colp + (j - 1) * 16) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        }
      };
      if (OUT_BF16) {
        store_bf16((unsigned short*)c.C, g.ldc, c.c_off);
      } else {
        if (live) *(float4*)((float*)c.C + idx) = make_float4(v[0], v[1], v[2], v[3]);
        if (g.c_bf16) store_bf16(g.c_bf16, g.ld_c_bf16, 0);   // the next GEMM's A operand, un-normalised (its ScaleNorm factor rides in that GEMM's row_scale)
        if (g.row_sumsq && live) ssq[i4] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      }
    });
    if (!OUT_BF16 && g.row_sumsq) {
      // the 4 NJ values a lane holds of row i (NJ sub-tiles x 4 columns) + the three other lanes of that row -> one slot per wave
      static_for<RI>([&](auto it) {
        constexpr int i4 = decltype(it)::value;
        float t = ssq[i4];
        t += __shfl_xor(t, TACC ? 16 : 4, 64);   // the four lanes that share a row
        t += __shfl_xor(t, TACC ? 32 : 8, 64);
        const int64_t r = row0 + (round * RI + i4) * 16;
        if ((TACC ? lane < 16 : (lane & 12) == 0) && r < g.M) g.row_sumsq[r * g.ld_row_sumsq + (int64_t)((unsigned)nw / (unsigned)(16 * NJ))] = t;
      });
    }
  });
}

// true when epilogue_fast covers this launch's operators; the caller checks that the tile does not cross N (rows past M
// are masked per lane, so the bottom tile row of an M that is not a multiple of the tile still takes the fast path)
template <int EXT>
__device__ __forceinline__ bool epilogue_fast_ok(const tribe_gemm_desc& g, const EpiCtx& c) {
  const bool act_ok = g.act == TRIBE_ACT_NONE || g.act == TRIBE_ACT_GELU ||
                      (EXT && (g.act == TRIBE_ACT_SWIGLU || g.act == TRIBE_ACT_GLU || g.act == TRIBE_ACT_SILU || g.act == TRIBE_ACT_GELU_BWD ||
                              g.act == TRIBE_ACT_EXP2 || g.act == TRIBE_ACT_MUL_AUX));
  const bool rowadd_ok = !g.rowadd || (!c.res && g.M < (1ll << 31) && g.rowadd_period < (1ll << 31));   // rides in the residual's LDS slot
  return c.vec && rowadd_ok && !g.gadd && (EXT || !g.aux) && act_ok;   // (the launcher refuses the fused-norm operands unless this holds)
}

// One 16x16 accumulator tile -> epi(...) -> C, straight from registers (one 16-/8-byte store per lane).
template <int OUT_BF16, int EXT, int TACC = 0>
__device__ __forceinline__ void epilogue_tile16(const tribe_gemm_desc& g, const EpiCtx& c, f32x4_t acc, int64_t mt0,
                                                int64_t nt0, int lane) {
  float v[4];
  if constexpr (TACC) {   // transposed accumulation (see epilogue_fast): four consecutive columns of one row per lane as they are
    const int tq = lane >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = acc[k] * g.alpha;
    epilogue_row4<OUT_BF16, EXT>(g, c, v, mt0 + (lane & 15), nt0 + 4 * ((tq & 1) * 2 + (tq >> 1)));
  } else {
    quad_transpose(acc, g.alpha, lane, v);
    epilogue_row4<OUT_BF16, EXT>(g, c, v, mt0 + ((lane >> 4) << 2) + (lane & 3), nt0 + (((lane & 15) >> 2) << 2));
  }
}
