// Shared by the attention kernels (attention.hip: every head size, 16 query rows per wave, and DH = 384; attention_d64.hip: DH = 64,
// 64 query rows per wave): launch arguments, the LDS-DMA helper, cross-half lane exchange.
#pragma once
#include "gemm_common.h"
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// max(a, b, c) in one instruction.  fmaxf() chains compile to v_max_f32 PLUS a canonicalising v_max_f32 x, x per MFMA-produced input
// (IEEE maxnum semantics): 56 instructions per 128-key tile at DH = 64 where 16 suffice.  Not volatile: the scheduler may move it.
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// One LDS-DMA wave-instruction: lane l copies the 16 bytes at gsrc (per lane) to LDS byte address lds_dst + 16 l (lds_dst
// wave-uniform).  Inline asm, NOT __builtin_amdgcn_global_load_lds: hipcc tracks the builtin as a VMEM write to LDS and, unable
// to prove that a later ds_read touches another ring slot, puts s_waitcnt vmcnt(0) in front of the first LDS read after it --
// every tile then drained ALL its LDS-DMA loads, the multi-slot rings of these kernels never had a tile in flight and each
// tile paid a full L2 / HBM round trip (seen in the ISA of the round-1 kernels, and in stamps as 5900 of 7600 cycles per tile
// when pieces were issued between MFMAs).  Unseen by the compiler, the loads are waited for by the kernels' own counted
// s_waitcnt vmcnt(N) + barrier; hipcc's counted waits for its OWN loads stay correct (hidden younger loads only make a counted
// wait cover more).  M0 is compiler-reserved: saved and restored inside the statement (guide 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(lptr_t)p; }

struct AttnArgs {
  const unsigned short* q; const unsigned short* k; const unsigned short* v;
  unsigned short* out;
  int64_t ld_q, ld_kv, ld_out;
  int T, heads_q, group;  // group = heads_q / heads_kv
  float scale_log2e;
  int qblocks;
  int n_bh;   // batch * heads_q: the (sequence, head) pairs
  // relative_key position bias (Wav2Vec2BertSelfAttention, modeling_wav2vec2_bert.py:308-320): score += q . E[clamp(j - i)]
  const float* qe; int64_t ld_qe; int qe_stride_h; int rel_left, rel_right;  // qe[row][h * stride + clamp(j-i, -left, right) + left]
  // partial rotary of Q applied while the Q fragments are loaded (DH = 384 one-wave kernel only): interleaved pairs (2 i, 2 i + 1),
  // tables f32 [T, rot_dim / 2]; nullptr = q arrives rotated.  Same arithmetic, same bf16 rounding as rotary_kernel.
  const float* q_cos; const float* q_sin; int q_rot_dim;
  float* lse;   // optional [B, heads_q, T] base-2 log-sum-exp of the scaled scores (DH = 384 one-wave kernel only; nullptr = not wanted)
};

// compile-time loop: the accumulator tile index selects literal registers (attn_acc_regs.h)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// The value lane ^ 32 holds, combined with this lane's.  v_permlane32_swap exchanges lanes 32..63 of its first operand with
// lanes 0..31 of its second: fed two COPIES of x it leaves x[lane & 31] in one register and x[32 + (lane & 31)] in the other, in
// every lane.  Inline asm with two read-write operands: the builtin handed both copies the same register (the swap then only
// rotates x by 32 lanes and a lane never sees its own value).  The s_nop pads the VALU-write -> permlane-read hazard.
__device__ __forceinline__ void pair_split(float x, float& lo, float& hi) {
  lo = x;
  hi = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
}
__device__ __forceinline__ float pair_max(float x) {
  float lo, hi;
  pair_split(x, lo, hi);
  return fmaxf(lo, hi);
}
__device__ __forceinline__ float pair_sum(float x) {
  float lo, hi;
  pair_split(x, lo, hi);
  return lo + hi;
}

}  // namespace

// Diagnostic build only (-DTRIBE_ATTN_STAMPS, scripts/attn_stamps.py): s_memtime stamps at the phase boundaries of a key loop, summed per
// wave into a side buffer whose pointer rides in desc.rel_qe.  Never quote the run time of that build; read the shares.
#ifdef TRIBE_ATTN_STAMPS
#define ATTN_STAMP(var)                                                                   \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#define ATTN_STAMP_ADD(i, a, b) stamp_acc[i] += (b) - (a)
#else
#define ATTN_STAMP(var) do {} while (0)
#define ATTN_STAMP_ADD(i, a, b) do {} while (0)
#endif
