// Fused bidirectional attention forward at DH = 64 (the V-JEPA2 ViT-g and Wav2Vec-BERT extractors, where it is half of the
// forward: profiles/r02_j_*): out = softmax((q k^T + rel) * scale) v per (sequence, head), q / k / v read in place from the
// projection buffers, the [T, T] scores never leave registers.  HF modeling_vjepa2.py (eager attention after 3-D rope) and
// modeling_wav2vec2_bert.py:308-320 (relative_key bias) are the semantics; data_utils/features/{video,audio}.py call it.
//
// Why a kernel of its own.  At DH = 64 a score costs 256 MFMA flops but a fixed ~6 vector instructions (max, scale-and-subtract,
// exp2, convert), so the vector ALU and not the matrix pipe is the roofline (MI355X_MICROARCH "vector-instruction ISSUE cost":
// v_exp_f32 8 cycles, the others 4, per wave-instruction; a 32x32x16 MFMA owns the matrix pipe for 32).  The 16-row kernel of
// attention.hip on top of that reads every K / V fragment from LDS for ONE 16x16x32 MFMA -- 256 B/clk/CU at full MFMA rate, the
// whole LDS bandwidth -- and stood at 28 % of the MFMA peak (V-JEPA2) or 5 % (Wav2Vec-BERT, which also gathered the relative
// bias from global memory for EVERY score).  Here:
//   * a wave owns 64 query rows = two 32-row tiles, v_mfma_f32_32x32x16_bf16: every K / V fragment read from LDS feeds two MFMAs
//     of 32 cycles (64 B/clk/CU at full rate); four waves = 256 query rows per workgroup, two workgroups per CU (<= 256 registers);
//   * S^T = K Q^T puts the query on the lane (col = lane & 31), so the softmax state is one scalar per lane and tile; the S^T
//     accumulator is re-used as the B operand of O^T += V^T P^T (element j of lane half h of k-step s is key 16 s + 8 (j >> 2) +
//     4 h + (j & 3)); V^T comes from the row-major V tile through ds_read_b64_tr_b16 in exactly that key order;
//   * the row sums of P come from one more MFMA per 16 keys (A = all ones: every accumulator row holds sum_key P^T[key][q] of the
//     bf16 P the numerator uses) instead of 16 adds per lane: the vector ALU is the scarcer unit;
//   * the relative_key bias q . E[clamp(j - i, -left, right)] is CONSTANT per query row outside the band -left < j - i < right:
//     a 32-key sub-tile wholly left (right) of the band of all 32 rows adds qe[row][0] (qe[row][left + right]) -- folded into the
//     exponent's fma, zero extra instructions per score -- and only the 3-4 sub-tiles per wave that cross the band gather;
//   * K / V tiles of 64 keys arrive by LDS-DMA into a 3-slot ring (48 KiB), issued two tiles ahead and waited for with a counted
//     vmcnt.
// Two schedules share the wave-level code (struct D64Wave):
//   attn_fwd_d64_kernel       4 waves, 256 query rows, two workgroups per CU, one barrier per tile.  The two waves of a SIMD belong
//                             to different workgroups and drift: stamps (scripts/attn64_stamps.py) show a tile costing the SUM of
//                             both waves' MFMA and softmax phases (S^T 20 %, softmax 44 %, P V 22 %, wait 14 % of 5255 ticks) --
//                             matrix beside matrix and vector beside vector half of the time.  Kept for small grids.
//   attn_fwd_d64_pair_kernel  8 waves, 512 query rows, one workgroup per CU.  Waves w and w + 4 share a SIMD and run in ANTI-PHASE,
//                             a barrier at every phase change: while one issues the 20 MFMAs of a 32-key step (P V of the previous
//                             sub-tile, S^T of the next), its partner runs the softmax of its own sub-tile on the vector ALU
//                             (MI355X_MICROARCH "Two waves per SIMD": matrix beside vector is the complementary pairing).
// LDS image: plain 128-byte rows, 16-byte chunks XOR-swizzled by f(row) = bit-reverse((row >> 1) & 7): the 16 rows of a
// ds_read_b128 group land on 16 distinct chunk positions of the 256-byte bank window, and so do the 4 keys x 64 bytes of a
// transposed V read (bit 2 of the chunk flips with bit 1 of the row).
#include "attn_common.h"

namespace {

struct D64Cfg {
  static constexpr int DH = 64, ROWB = 128, KV = 64, KS = 4;
  static constexpr int TILE_BYTES = KV * ROWB;              // 8 KiB per K or V tile
  static constexpr int QT = 2;                              // 32-row tiles per wave
  static constexpr int SLOTS = 3;
  static constexpr int V_BASE = SLOTS * TILE_BYTES;         // K ring first, V ring behind it
  static constexpr int SMEM = 2 * SLOTS * TILE_BYTES;       // 48 KiB
};

__device__ __forceinline__ int swz64(int chunk, int row) {
  return chunk ^ ((((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1));
}

// Everything one wave owns: 64 query rows of one (sequence, head) pair.  WAVES = waves per workgroup (4 or 8): only the staging split
// depends on it (8 K + 8 V pieces of 1 KiB per tile, 16 / WAVES each).
template <int RELKEY, int WAVES>
struct D64Wave {
  using C = D64Cfg;
  static constexpr int PPW = 8 / WAVES;   // K pieces (and V pieces) per wave and tile
  const AttnArgs& a;
  char* smem;
  int lane, wave, r31, h, T, q0, b, hd;
  float scale_log2e;
  const unsigned short* kbase; const unsigned short* vbase;
  int64_t ld;
  int qrow[C::QT];
  bf16x8_t qf[C::QT][C::KS];
  const float* qe_row[C::QT];
  float c_left[C::QT], c_right[C::QT];
  int st_row[PPW], st_src[PPW], st_off[PPW];
  int k_rd[C::KS], v_rd[2][2];
  f32x16_t o[C::QT][2], lacc[C::QT], s[C::QT];
  bf16x8_t pf[C::QT][2], ones;
  float m_run[C::QT];

  __device__ __forceinline__ D64Wave(const AttnArgs& a_, char* smem_) : a(a_), smem(smem_) {}

  // false: this workgroup is grid padding
  __device__ __forceinline__ bool init() {
    const int tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    r31 = lane & 31; h = lane >> 5;
    T = a.T;
    scale_log2e = a.scale_log2e;
    // all query blocks of one (sequence, head) pair run on one XCD, so its K / V come from HBM once (see attention.hip)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qb = slot % a.qblocks;
    const int pair = (slot / a.qblocks) * 8 + xcd;
    if (pair >= a.n_bh) return false;
    hd = pair % a.heads_q;
    b = pair / a.heads_q;
    const int hk = hd / a.group;
    ld = a.ld_kv;
    const unsigned short* qbase = a.q + (int64_t)b * T * a.ld_q + (int64_t)hd * C::DH;
    kbase = a.k + (int64_t)b * T * ld + (int64_t)hk * C::DH;
    vbase = a.v + (int64_t)b * T * ld + (int64_t)hk * C::DH;

    // ---- Q^T fragments (B operand of S^T): for row tile qt, lane (query r31, half h) holds Q[q][16 ks + 8 h .. +7] ----
    q0 = qb * (WAVES * C::QT * 32) + wave * (C::QT * 32);
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
      const int q = q0 + 32 * qt + r31;
      qrow[qt] = q < T ? q : T - 1;
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) qf[qt][ks] = *(const bf16x8_t*)(qbase + (int64_t)qrow[qt] * a.ld_q + ks * 16 + h * 8);
    }
    // relative_key: the two out-of-band constants of each row, and the row's table for the band
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
      qe_row[qt] = nullptr; c_left[qt] = 0.f; c_right[qt] = 0.f;
      if (RELKEY) {
        qe_row[qt] = a.qe + ((int64_t)b * T + qrow[qt]) * a.ld_qe + hd * a.qe_stride_h + a.rel_left;
        c_left[qt] = qe_row[qt][-a.rel_left];
        c_right[qt] = qe_row[qt][a.rel_right];
      }
    }
    // ---- staging plan: piece p = wave + WAVES i of a tile is its rows 8 p .. 8 p + 7; lane l fills LDS chunk l of the piece (row
    // 8 p + l / 8, physical chunk l % 8) from the source chunk the swizzle maps there ----
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      st_row[i] = 8 * (wave + WAVES * i) + (lane >> 3);
      st_src[i] = swz64(lane & 7, st_row[i]) * 8;   // element offset of the SOURCE chunk inside the row
      st_off[i] = st_row[i] * (int)ld + st_src[i];
    }
    // ---- per-lane LDS read offsets inside a slot ----
    // K row read (A operand of S^T), sub-tile u, k-step ks: row 32 u + r31, chunk 2 ks + h
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) k_rd[ks] = r31 * C::ROWB + swz64(2 * ks + h, r31) * 16;
    // V transposed read (A operand of O^T), d-tile dt, key half lh of a 16-key step: lane i of a 16-lane group supplies row i >> 2 of
    // a 4-key block, 4 columns; lanes 16-31 of a half the second 16 columns of the 32-column d-tile
    const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int lh = 0; lh < 2; ++lh) {
        const int row = 4 * h + tq + 8 * lh;
        v_rd[dt][lh] = C::V_BASE + row * C::ROWB + swz64(4 * dt + 2 * g1 + (tp >> 1), row) * 16 + (tp & 1) * 8;
      }
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[qt][0][r] = 0.f; o[qt][1][r] = 0.f; lacc[qt][r] = 0.f; }
      m_run[qt] = -INFINITY;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;
    // The compiler waits for its own global loads at their first use -- inside the key loop, where its s_waitcnt vmcnt(0) would also
    // drain the LDS-DMA loads it cannot see.  Using the loaded values here pins those waits in front of the loop.
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[qt][ks]));
      if (RELKEY) asm volatile("" : "+v"(c_left[qt]), "+v"(c_right[qt]));
    }
    return true;
  }

  // this wave's pieces of the K and V tiles that start at key0, into ring slot `ring_slot`: 2 PPW LDS-DMA loads
  __device__ __forceinline__ void stage(int ring_slot, int key0) {
    const unsigned short* kb = kbase + (int64_t)key0 * ld;
    const unsigned short* vb = vbase + (int64_t)key0 * ld;
    const bool full = key0 + C::KV <= T;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int off = st_off[i];
      if (!full) {   // tail keys re-read the last valid row; they are masked to -inf in the softmax
        const int row = (key0 + st_row[i] < T) ? st_row[i] : T - 1 - key0;
        off = row * (int)ld + st_src[i];
      }
      const unsigned dst = lds_addr(smem) + ring_slot * C::TILE_BYTES + (wave + WAVES * i) * 1024;
      glds16(kb + off, dst);
      glds16(vb + off, dst + C::V_BASE);
    }
  }

  // S^T[key][q] = sum_d K[key][d] Q[q][d] for both row tiles, keys 32 u .. 32 u + 31 of the tile at `tile`: each K fragment feeds two MFMAs
  __device__ __forceinline__ void s_mfmas(const char* tile, int u) {
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[qt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const bf16x8_t kf = *(const bf16x8_t*)(tile + k_rd[ks] + u * 32 * C::ROWB);
#pragma unroll
      for (int qt = 0; qt < C::QT; ++qt) s[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qt][ks], s[qt], 0, 0, 0);
    }
  }

  // online softmax of row tile qt, first part: s[qt][r] = S^T[key = sub_key0 + (r & 3) + 8 (r >> 2) + 4 h][q = r31], raw scores; adds the
  // relative_key bias, masks the keys past T, folds the sub-tile's maximum into the running one (rescaling O^T when it grew by > 2^8)
  // and returns the addend of the exponent's fma
  __device__ __forceinline__ float softmax_head(int qt, int sub_key0) {
    float cbias = 0.f;   // a per-row constant added to every score of the sub-tile
    if (RELKEY) {
      const int q_lo = q0 + 32 * qt, q_hi = q_lo + 31;   // wave-uniform classification over the tile's 32 rows
      if (sub_key0 + 31 - q_lo <= -a.rel_left) cbias = c_left[qt];
      else if (sub_key0 - q_hi >= a.rel_right) cbias = c_right[qt];
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int dist = sub_key0 + (r & 3) + 8 * (r >> 2) + 4 * h - qrow[qt];
          dist = dist < -a.rel_left ? -a.rel_left : (dist > a.rel_right ? a.rel_right : dist);
          s[qt][r] += qe_row[qt][dist];
        }
      }
    }
    if (sub_key0 + 32 > T) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (sub_key0 + (r & 3) + 8 * (r >> 2) + 4 * h >= T) s[qt][r] = -INFINITY;
    }
    // The FIRST read of the MFMA result must be an instruction the compiler sees (fmaxf): it pads the MFMA-write -> VALU-read hazard
    // in front of it (18 wait states after a 16-pass MFMA, no hardware interlock).  The inline-asm v_max3 chain gets no such padding --
    // begun with an asm it read the accumulator before the last MFMA had written it (run-to-run differences in the ViT-g encoder) --
    // so every link takes the running maximum as an operand and cannot be scheduled above the fmaxf.
    float pmax = fmaxf(s[qt][14], s[qt][15]);
#pragma unroll
    for (int r = 0; r < 14; r += 2) pmax = max3f(pmax, s[qt][r], s[qt][r + 1]);
    pmax = (pair_max(pmax) + cbias) * scale_log2e;
    if (!__all(pmax - m_run[qt] <= 8.0f)) {   // deferred max (guide T13): rescale only when some row's maximum grew by > 2^8
      const float m_new = fmaxf(m_run[qt], pmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
      m_run[qt] = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[qt][0][r] *= alpha; o[qt][1][r] *= alpha; lacc[qt][r] *= alpha; }
    }
    return fmaf(cbias, scale_log2e, -m_run[qt]);
  }
  // second part: P^T = exp2(s * scale + shift) as bf16 MFMA operands (40 vector instructions, no branch)
  __device__ __forceinline__ void softmax_exp(int qt, float shift) {
    // (v_pk_fma_f32 for pairs of scores was measured slower: 0.950 vs 0.930 ms on V-JEPA2 x 2, same box)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      pf[qt][r >> 3][r & 7] = (short)f32_to_bf16(__builtin_amdgcn_exp2f(fmaf(s[qt][r], scale_log2e, shift)));
  }
  __device__ __forceinline__ void softmax(int sub_key0) {
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) softmax_exp(qt, softmax_head(qt, sub_key0));
  }

  // One 32-key sub-tile with the matrix and the vector work of the two row tiles INTERLEAVED inside the wave (the two waves of a SIMD
  // only time-share it: counters show MFMA-busy + VALU-busy = 100 % of the run time in the phase-separated order):
  //   S^T(tile 0) | head(0) | S^T(tile 1) beside exp(0) | head(1) | P V(tile 0) beside exp(1) | P V(tile 1)
  // sched_group_barrier pins one MFMA per group of vector instructions inside the two mixed regions.
  __device__ __forceinline__ void subtile_interleaved(const char* tile, int u, int sub_key0) {
    bf16x8_t kf[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) kf[ks] = *(const bf16x8_t*)(tile + k_rd[ks] + u * 32 * C::ROWB);
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[0][ks], s[0], 0, 0, 0);
    bf16x8_t vf[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int imm = (u * 32 + st * 16) * C::ROWB;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + v_rd[dt][0] + imm));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + v_rd[dt][1] + imm));
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[st][dt][e] = lo[e]; vf[st][dt][4 + e] = hi[e]; }
      }
    const float shift0 = softmax_head(0, sub_key0);
    // ---- mixed region 1: 4 MFMAs of S^T(tile 1) beside the 40 vector instructions of exp(0) ----
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[1][ks], s[1], 0, 0, 0);
    softmax_exp(0, shift0);
    asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]));   // keeps exp(0) here: its first real use is two branches further down, and LLVM sinks it there
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 10, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float shift1 = softmax_head(1, sub_key0);
    // ---- mixed region 2: 6 MFMAs of P V(tile 0) beside exp(1) ----
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 2; ++st) lacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[0][st], lacc[0], 0, 0, 0);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st][dt], pf[0][st], o[0][dt], 0, 0, 0);
    softmax_exp(1, shift1);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 7, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 2; ++st) lacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[1][st], lacc[1], 0, 0, 0);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st][dt], pf[1][st], o[1][dt], 0, 0, 0);
  }

  // O^T[d][q] += sum_key V[key][d] P^T[key][q] for keys 32 u .. of the tile at `tile`, row sums by the all-ones MFMA (first: they need
  // no LDS operand and cover the latency of the V reads); each V fragment feeds two MFMAs
  __device__ __forceinline__ void pv_mfmas(const char* tile, int u) {
    bf16x8_t vf[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int imm = (u * 32 + st * 16) * C::ROWB;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + v_rd[dt][0] + imm));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + v_rd[dt][1] + imm));
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[st][dt][e] = lo[e]; vf[st][dt][4 + e] = hi[e]; }
      }
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int qt = 0; qt < C::QT; ++qt) lacc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[qt][st], lacc[qt], 0, 0, 0);
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st][dt], pf[qt][st], o[qt][dt], 0, 0, 0);
  }

  // normalise and write: O^T[d = 32 dt + (r & 3) + 8 (r >> 2) + 4 h][q = r31] -> out[q][hd * 64 + d], 4 consecutive d per store
  __device__ __forceinline__ void finish() {
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
      const int q = q0 + 32 * qt + r31;
      if (q >= T) continue;
      const float inv = 1.0f / lacc[qt][0];
      unsigned short* orow = a.out + ((int64_t)b * T + q) * a.ld_out + (int64_t)hd * C::DH + 4 * h;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u16x4_t pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = f32_to_bf16(o[qt][dt][4 * g + e] * inv);
          *(u16x4_t*)(orow + 32 * dt + 8 * g) = pk;
        }
    }
  }
};

template <int RELKEY, int INTERLEAVE>
__global__ __launch_bounds__(256, 2) void attn_fwd_d64_kernel(const AttnArgs a) {
  using C = D64Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  D64Wave<RELKEY, 4> w(a, smem);
  if (!w.init()) return;
  const int T = w.T;
  const int ntiles = (T + C::KV - 1) / C::KV;
  w.stage(0, 0);
  if (ntiles > 1) w.stage(1, C::KV);

#ifdef TRIBE_ATTN_STAMPS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
  unsigned long long stamp_acc[5] = {0, 0, 0, 0, 0};   // 0 wait + barrier + staging, 1 S^T MFMAs, 2 softmax, 3 P V
#endif
  for (int t = 0; t < ntiles; ++t) {
    ATTN_STAMP(ts0);
    // this wave's 4 loads of tile t + 1 may stay in flight; tile t has landed for every wave after the barrier, and every wave
    // has left tile t - 1, whose slot tile t + 2 takes
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < ntiles) w.stage((t + 2) % C::SLOTS, (t + 2) * C::KV);
    const char* tile = smem + (t % C::SLOTS) * C::TILE_BYTES;
    ATTN_STAMP(ts1);
    ATTN_STAMP_ADD(0, ts0, ts1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int sub_key0 = t * C::KV + 32 * u;
      if (sub_key0 >= T) break;   // wave-uniform: past the end of the sequence
      if (INTERLEAVE) {
        w.subtile_interleaved(tile, u, sub_key0);
      } else {
        w.s_mfmas(tile, u);
        ATTN_STAMP(ts2);
        w.softmax(sub_key0);
        ATTN_STAMP(ts3);
        w.pv_mfmas(tile, u);
      }
      ATTN_STAMP(ts4);
      ATTN_STAMP_ADD(1, ts1, ts2); ATTN_STAMP_ADD(2, ts2, ts3); ATTN_STAMP_ADD(3, ts3, ts4);
#ifdef TRIBE_ATTN_STAMPS
      ts1 = ts4;
#endif
    }
  }
#ifdef TRIBE_ATTN_STAMPS
  if (a.qe != nullptr && w.lane == 0) {
    unsigned long long* dbg = (unsigned long long*)a.qe + ((size_t)blockIdx.x * 4 + w.wave) * 8;
    for (int i = 0; i < 4; ++i) dbg[i] = stamp_acc[i];
    dbg[5] = (unsigned long long)ntiles;
  }
#endif
  w.finish();
}

// Anti-phase pairs.  Global phase p = 0, 1, ...; every phase ends at a workgroup barrier.  Half A (waves 0-3) runs its matrix phase
// at even p, half B (waves 4-7, the SIMD partners) at odd p, and each its softmax in the phase after:
//   matrix phase of a wave, sub-tile u = (p - half) / 2:   P V of sub-tile u - 1 (its P^T from the softmax just before), then S^T of u
//   softmax phase, sub-tile u = (p - half - 1) / 2:        S^T -> P^T, rescale of O^T when a maximum grew
// Tile t = sub-tiles 2 t, 2 t + 1 is first read in phase 4 t (A's S^T) and last in phase 4 t + 5 (B's P V of sub-tile 2 t + 1); every
// wave issues its two pieces of tile t + 2 at the start of phase 4 t + 2 into the slot of tile t - 1 (last read in phase 4 t + 1) and
// waits for its pieces of tile t + 1 before the barrier that ends phase 4 t + 3 (the pieces of tile t + 2 stay in flight).
template <int RELKEY>
__global__ __launch_bounds__(512, 1) void attn_fwd_d64_pair_kernel(const AttnArgs a) {
  using C = D64Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  D64Wave<RELKEY, 8> w(a, smem);
  if (!w.init()) return;
  const int T = w.T;
  const int ntiles = (T + C::KV - 1) / C::KV;
  const int nsub = (T + 31) / 32;
  const int half = w.wave >> 2;
  w.stage(0, 0);
  if (ntiles > 1) w.stage(1, C::KV);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  auto phase_begin = [&](int p) {
    if ((p & 3) == 2) {
      const int tn = (p >> 2) + 2;
      if (tn < ntiles) w.stage(tn % C::SLOTS, tn * C::KV);
    }
  };
  auto phase_end = [&](int p) {
    if ((p & 3) == 3) {
      // own pieces of tile (p >> 2) + 1 were issued in phase p - 5, those of the tile after it in phase p - 1
      if ((p >> 2) + 2 < ntiles) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  };
  // one loop per half (not one loop with the half tested inside: the accumulators then pass through a phi per phase and spill)
  int p = 0;
  if (half == 1) { phase_begin(0); phase_end(0); p = 1; }
#pragma unroll 1
  for (int u = 0; u <= nsub; ++u) {
    // ---- matrix phase: P V of sub-tile u - 1, S^T of sub-tile u ----
    phase_begin(p);
    if (u >= 1) w.pv_mfmas(smem + (((u - 1) >> 1) % C::SLOTS) * C::TILE_BYTES, (u - 1) & 1);
    if (u < nsub) w.s_mfmas(smem + ((u >> 1) % C::SLOTS) * C::TILE_BYTES, u & 1);
    phase_end(p);
    ++p;
    if (p >= 2 * nsub + 2) break;   // half B: its last matrix phase is the last phase of the workgroup
    // ---- softmax phase ----
    phase_begin(p);
    if (u < nsub) w.softmax(32 * u);
    phase_end(p);
    ++p;
  }
  w.finish();
}

}  // namespace

// Launch for `a` filled by tribe_attention_fwd_ex (attention.hip): B sequences, bidirectional, dim_head 64.
// variant: 0 = default, 1 = the 4-wave kernel in phase-separated order, 2 = the anti-phase 8-wave kernel, 3 = the 4-wave kernel with the
// two row tiles interleaved.
int tribe_internal_attn_d64_launch(const void* args, int64_t B, int relkey, int variant, hipStream_t s) {
  using C = D64Cfg;
  AttnArgs a = *(const AttnArgs*)args;
  a.n_bh = (int)(B * a.heads_q);
  TRIBE_REQUIRE((int64_t)a.T * a.ld_kv < (1ll << 31), "tribe_attention_fwd: sequence too long for 32-bit offsets");
  const int64_t pairs8 = (int64_t)((a.n_bh + 7) / 8) * 8;
  const int qb512 = (a.T + 511) / 512, qb256 = (a.T + 255) / 256;
  const bool pair_kernel = variant == 2;
  a.qblocks = pair_kernel ? qb512 : qb256;
  const int64_t nblocks = pairs8 * a.qblocks;
  TRIBE_REQUIRE(nblocks < (1ll << 31), "tribe_attention_fwd: grid too large");
  if (pair_kernel) {
    if (relkey) hipLaunchKernelGGL((attn_fwd_d64_pair_kernel<1>), dim3((unsigned)nblocks), dim3(512), C::SMEM, s, a);
    else hipLaunchKernelGGL((attn_fwd_d64_pair_kernel<0>), dim3((unsigned)nblocks), dim3(512), C::SMEM, s, a);
  } else if (variant == 1) {
    if (relkey) hipLaunchKernelGGL((attn_fwd_d64_kernel<1, 0>), dim3((unsigned)nblocks), dim3(256), C::SMEM, s, a);
    else hipLaunchKernelGGL((attn_fwd_d64_kernel<0, 0>), dim3((unsigned)nblocks), dim3(256), C::SMEM, s, a);
  } else {
    if (relkey) hipLaunchKernelGGL((attn_fwd_d64_kernel<1, 1>), dim3((unsigned)nblocks), dim3(256), C::SMEM, s, a);
    else hipLaunchKernelGGL((attn_fwd_d64_kernel<0, 1>), dim3((unsigned)nblocks), dim3(256), C::SMEM, s, a);
  }
  TRIBE_LAUNCH_CHECK();
  return 0;
}
