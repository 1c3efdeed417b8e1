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
//   * K / V tiles of 64 keys arrive by LDS-DMA into a 3-slot ring (48 KiB): tile t + 2 is issued right after the barrier that
//     opens tile t, waited for with a counted vmcnt two tiles later; one barrier per tile.
// LDS image: plain 128-byte rows, 16-byte chunks XOR-swizzled by f(row) = bit-reverse((row >> 1) & 7): the 16 rows of a
// ds_read_b128 group land on 16 distinct chunk positions of the 256-byte bank window, and so do the 4 keys x 64 bytes of a
// transposed V read (bit 2 of the chunk flips with bit 1 of the row).
#include "attn_common.h"

namespace {

struct D64Cfg {
  static constexpr int DH = 64, ROWB = 128, KV = 64, CHUNKS = 8, KS = 4;
  static constexpr int TILE_BYTES = KV * ROWB;              // 8 KiB per K or V tile
  static constexpr int WAVES = 4, QT = 2, ROWS = WAVES * QT * 32;   // 256 query rows per workgroup
  static constexpr int SLOTS = 3;
  static constexpr int V_BASE = SLOTS * TILE_BYTES;         // K ring first, V ring behind it
  static constexpr int SMEM = 2 * SLOTS * TILE_BYTES;       // 48 KiB
  static constexpr int PPW = 2;                             // 1-KiB LDS-DMA pieces per wave, operand and tile (8 pieces / 4 waves)
};

__device__ __forceinline__ int swz64(int chunk, int row) {
  return chunk ^ ((((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1));
}

template <int RELKEY>
__global__ __launch_bounds__(256, 2) void attn_fwd_d64_kernel(const AttnArgs a) {
  using C = D64Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;

  const int T = a.T;
  const float scale_log2e = a.scale_log2e;
  // all query blocks of one (sequence, head) pair run on one XCD, so its K / V come from HBM once (see attention.hip)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int qb = slot % a.qblocks;
  const int pair = (slot / a.qblocks) * 8 + xcd;
  if (pair >= a.n_bh) return;
  const int hd = pair % a.heads_q;
  const int b = pair / a.heads_q;
  const int hk = hd / a.group;
  const int64_t ld = a.ld_kv;
  const unsigned short* qbase = a.q + (int64_t)b * T * a.ld_q + (int64_t)hd * C::DH;
  const unsigned short* kbase = a.k + (int64_t)b * T * ld + (int64_t)hk * C::DH;
  const unsigned short* vbase = a.v + (int64_t)b * T * ld + (int64_t)hk * C::DH;

  // ---- Q^T fragments (B operand of S^T): for row tile qt, lane (query r31, half h) holds Q[q][16 ks + 8 h .. +7] ----
  const int q0 = qb * C::ROWS + wave * (C::QT * 32);
  int qrow[C::QT];
  bf16x8_t qf[C::QT][C::KS];
#pragma unroll
  for (int qt = 0; qt < C::QT; ++qt) {
    const int q = q0 + 32 * qt + r31;
    qrow[qt] = q < T ? q : T - 1;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) qf[qt][ks] = *(const bf16x8_t*)(qbase + (int64_t)qrow[qt] * a.ld_q + ks * 16 + h * 8);
  }
  // relative_key: the two out-of-band constants of each row, and the row's table for the band
  const float* qe_row[C::QT] = {nullptr, nullptr};
  float c_left[C::QT] = {0.f, 0.f}, c_right[C::QT] = {0.f, 0.f};
  if (RELKEY) {
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
      qe_row[qt] = a.qe + ((int64_t)b * T + qrow[qt]) * a.ld_qe + hd * a.qe_stride_h + a.rel_left;
      c_left[qt] = qe_row[qt][-a.rel_left];
      c_right[qt] = qe_row[qt][a.rel_right];
    }
  }

  // ---- staging plan: piece p = wave + 4 i of a tile is its rows 8 p .. 8 p + 7; lane l fills LDS chunk l of the piece (row 8 p + l / 8,
  // physical chunk l % 8) from the source chunk the swizzle maps there ----
  int st_row[C::PPW], st_src[C::PPW], st_off[C::PPW];
#pragma unroll
  for (int i = 0; i < C::PPW; ++i) {
    st_row[i] = 8 * (wave + C::WAVES * i) + (lane >> 3);
    st_src[i] = swz64(lane & 7, st_row[i]) * 8;   // element offset of the SOURCE chunk inside the row
    st_off[i] = st_row[i] * (int)ld + st_src[i];
  }
  auto stage = [&](int ring_slot, int key0) {
    const unsigned short* kb = kbase + (int64_t)key0 * ld;
    const unsigned short* vb = vbase + (int64_t)key0 * ld;
    const bool full = key0 + C::KV <= T;
#pragma unroll
    for (int i = 0; i < C::PPW; ++i) {
      int off = st_off[i];
      if (!full) {   // tail keys re-read the last valid row; they are masked to -inf below
        const int row = (key0 + st_row[i] < T) ? st_row[i] : T - 1 - key0;
        off = row * (int)ld + st_src[i];
      }
      const unsigned dst = lds_addr(smem) + ring_slot * C::TILE_BYTES + (wave + C::WAVES * i) * 1024;
      glds16(kb + off, dst);
      glds16(vb + off, dst + C::V_BASE);
    }
  };

  // ---- per-lane LDS read offsets inside a slot ----
  // K row read (A operand of S^T), sub-tile u, k-step ks: row 32 u + r31, chunk 2 ks + h
  int k_rd[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) k_rd[ks] = r31 * C::ROWB + swz64(2 * ks + h, r31) * 16;
  // V transposed read (A operand of O^T), d-tile dt, key half lh of a 16-key step: lane i of a 16-lane group supplies row i >> 2 of a
  // 4-key block, 4 columns; lanes 16-31 of a half the second 16 columns of the 32-column d-tile
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int v_rd[2][2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int lh = 0; lh < 2; ++lh) {
      const int row = 4 * h + tq + 8 * lh;
      v_rd[dt][lh] = C::V_BASE + row * C::ROWB + swz64(4 * dt + 2 * g1 + (tp >> 1), row) * 16 + (tp & 1) * 8;
    }

  f32x16_t o[C::QT][2], lacc[C::QT];
#pragma unroll
  for (int qt = 0; qt < C::QT; ++qt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[qt][0][r] = 0.f; o[qt][1][r] = 0.f; lacc[qt][r] = 0.f; }
  }
  float m_run[C::QT] = {-INFINITY, -INFINITY};
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

  // The compiler waits for its own global loads at their first use -- inside the key loop, where its s_waitcnt vmcnt(0) would also
  // drain the LDS-DMA loads of tile t + 2 it cannot see (issued a few instructions earlier): every tile then paid a full memory
  // round trip.  Using the loaded values here pins those waits in front of the loop.
#pragma unroll
  for (int qt = 0; qt < C::QT; ++qt) {
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) asm volatile("" : "+v"(qf[qt][ks]));
    if (RELKEY) asm volatile("" : "+v"(c_left[qt]), "+v"(c_right[qt]));
  }

  const int ntiles = (T + C::KV - 1) / C::KV;
  stage(0, 0);
  if (ntiles > 1) stage(1, C::KV);

  for (int t = 0; t < ntiles; ++t) {
    // this wave's 4 loads of tile t + 1 may stay in flight; tile t has landed for every wave after the barrier, and every wave
    // has left tile t - 1, whose slot tile t + 2 takes
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < ntiles) stage((t + 2) % C::SLOTS, (t + 2) * C::KV);
    const char* tile = smem + (t % C::SLOTS) * C::TILE_BYTES;

#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int sub_key0 = t * C::KV + 32 * u;
      if (sub_key0 >= T) break;   // wave-uniform: past the end of the sequence
      // ---- S^T[key][q] = sum_d K[key][d] Q[q][d] for both row tiles: each K fragment feeds two MFMAs ----
      f32x16_t s[C::QT];
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

      // ---- online softmax per row tile: s[qt][r] = S^T[key = sub_key0 + (r & 3) + 8 (r >> 2) + 4 h][q = r31], raw scores ----
      bf16x8_t pf[C::QT][2];
      const bool edge = sub_key0 + 32 > T;
#pragma unroll
      for (int qt = 0; qt < C::QT; ++qt) {
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
        if (edge) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (sub_key0 + (r & 3) + 8 * (r >> 2) + 4 * h >= T) s[qt][r] = -INFINITY;
        }
        float pmax = max3f(s[qt][0], s[qt][1], s[qt][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) pmax = max3f(pmax, s[qt][r], s[qt][r + 1]);
        pmax = fmaxf(pmax, s[qt][15]);
        pmax = (pair_max(pmax) + cbias) * scale_log2e;
        if (!__all(pmax - m_run[qt] <= 8.0f)) {   // deferred max (guide T13): rescale only when some row's maximum grew by > 2^8
          const float m_new = fmaxf(m_run[qt], pmax);
          const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
          m_run[qt] = m_new;
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[qt][0][r] *= alpha; o[qt][1][r] *= alpha; lacc[qt][r] *= alpha; }
        }
        const float shift = fmaf(cbias, scale_log2e, -m_run[qt]);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pf[qt][r >> 3][r & 7] = (short)f32_to_bf16(__builtin_amdgcn_exp2f(fmaf(s[qt][r], scale_log2e, shift)));
      }

      // ---- O^T[d][q] += sum_key V[key][d] P^T[key][q], row sums by the all-ones MFMA; each V fragment feeds two MFMAs ----
#pragma unroll
      for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) lacc[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[qt][st], lacc[qt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int imm = (u * 32 + st * 16) * C::ROWB;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + v_rd[dt][0] + imm));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + v_rd[dt][1] + imm));
          bf16x8_t vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
#pragma unroll
          for (int qt = 0; qt < C::QT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qt][st], o[qt][dt], 0, 0, 0);
        }
      }
    }
  }

  // ---- normalise and write: O^T[d = 32 dt + (r & 3) + 8 (r >> 2) + 4 h][q = r31] -> out[q][hd * 64 + d], 4 consecutive d per store ----
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

}  // namespace

// Launch for `a` filled by tribe_attention_fwd_ex (attention.hip): B sequences, bidirectional, dim_head 64.
int tribe_internal_attn_d64_launch(const void* args, int64_t B, int relkey, hipStream_t s) {
  using C = D64Cfg;
  AttnArgs a = *(const AttnArgs*)args;
  a.qblocks = (a.T + C::ROWS - 1) / C::ROWS;
  a.n_bh = (int)(B * a.heads_q);
  const int64_t nblocks = (int64_t)((a.n_bh + 7) / 8) * 8 * a.qblocks;
  TRIBE_REQUIRE(nblocks < (1ll << 31), "tribe_attention_fwd: grid too large");
  TRIBE_REQUIRE((int64_t)a.T * a.ld_kv < (1ll << 31), "tribe_attention_fwd: sequence too long for 32-bit offsets");
  if (relkey) hipLaunchKernelGGL((attn_fwd_d64_kernel<1>), dim3((unsigned)nblocks), dim3(256), C::SMEM, s, a);
  else hipLaunchKernelGGL((attn_fwd_d64_kernel<0>), dim3((unsigned)nblocks), dim3(256), C::SMEM, s, a);
  TRIBE_LAUNCH_CHECK();
  return 0;
}
