// Fused bidirectional attention forward for gfx950:  out = softmax(q k^T * scale) v  per (batch, head),
// reading q / k / v straight out of the fused qkv projection buffer [B*T, 3*heads*DH] (bf16, rotary
// already applied) and never materialising the [T, T] score matrix (x_transformers Attend, called via
// model.py:173; the reference runs it with attn_flash=False, i.e. materialised in HBM).
//
// Work split: one 512-thread workgroup = 128 query rows of one (batch, head); wave w owns 16 of them
// (two waves per SIMD, <= 256 registers each -- with DH = 384 a 32-row wave would need 192 accumulator
// + 96 Q registers and spills):
//   * Q^T fragments for all DH/32 k-steps live in registers (48 VGPRs at DH = 384);
//   * S^T = K . Q^T  with v_mfma_f32_16x16x32_bf16: A = K tile rows (keys) from LDS (ds_read_b128),
//     B = Q^T from registers.  The accumulator puts the query on the LANE (col = lane & 15), so the
//     online-softmax state (running max m, running sum l) is one scalar per lane;
//   * P^T (bf16) is the pair of S^T accumulators (keys 0-15 / 16-31 of the tile) re-used directly as the
//     B operand of the second product  O^T += V^T . P^T  (guide section 3 "An accumulator tile as the
//     next MFMA's operand"): k-slot 8*(lane>>4)+j of the MFMA is key 4*(lane>>4)+j (j < 4) or
//     16+4*(lane>>4)+j-4 (j >= 4), and the A operand V^T is read from the row-major V tile with
//     ds_read_b64_tr_b16 (hardware transpose) in exactly that key order;
//   * O^T [DH x 16 queries] stays in accumulators (96 registers at DH = 384) for the whole key loop; it is
//     rescaled only when some row maximum grew by more than 2^8 (deferred max, guide T13).
// K / V tiles of 32 keys are staged HBM -> LDS by global_load_lds_dwordx4 into two buffers; the 16-byte
// chunk swizzle  chunk ^= (row & 7) << 1  (applied on the SOURCE address, undone on the reads) keeps both
// the row reads of K and the transposed reads of V bank-conflict free.
// Roofline: MFMA (4*T*DH flop per query row and head); every MFMA streams a 1 KiB operand from LDS, so
// the LDS read rate (256 B/clk/CU) is the co-bound.
#include "attn_common.h"
#include "attn_acc_regs.h"

namespace {

// keys per staged tile = 32 * NSUB: small heads take several 32-key sub-tiles per barrier / softmax pass so that the
// fixed per-tile cost (barrier, max exchange, rescale test) is amortised over the same number of MFMAs as at DH = 384
template <int DH>
struct AttnCfg {
  static constexpr int NSUB = DH <= 64 ? 4 : (DH <= 128 ? 2 : 1);
  static constexpr int KS = DH / 32;            // k-steps of S^T = K Q^T (16x16x32)
  static constexpr int DT = DH / 16;            // 16-row tiles of O^T
  static constexpr int ROWB = DH * 2;           // bytes per K / V tile row
  static constexpr int KV = 32 * NSUB;          // keys per staged tile
  static constexpr int TILE_BYTES = KV * ROWB;  // one K or V tile
  static constexpr int CHUNKS = DH / 8;         // 16-byte chunks per row
  static constexpr int WAVES = 8;
  static constexpr int PIECES = KV * CHUNKS / 64;  // 1 KiB glds pieces per tile
  static constexpr int PPW = (PIECES + WAVES - 1) / WAVES;  // pieces per wave (last ones may be idle)
  static constexpr int SWZ_MASK = (CHUNKS % 16 == 0) ? 15 : 7;
  // K/V tile ring.  Large heads run one workgroup per CU anyway (registers), so they can afford three slots: tile t + 2 is
  // staged while tile t + 1 is still landing and tile t is consumed, behind a counted vmcnt -- one 32-key tile of work
  // (~1.5 us at DH = 384) is shorter than an L2-miss round trip.  Small heads keep two slots: a third would push their LDS
  // past half a CU and halve their occupancy (measured: V-JEPA2 +20 % time).
  static constexpr int NBUF = DH >= 192 ? 3 : 2;
  static constexpr int SMEM = NBUF * 2 * TILE_BYTES;
  static_assert(DH % 64 == 0, "dim_head must be a multiple of 64");
};

template <int DH>
__device__ __forceinline__ int swz_chunk(int chunk, int row) {
  constexpr int MASK = AttnCfg<DH>::SWZ_MASK;
  return (chunk & ~MASK) | ((chunk ^ ((row & 7) << 1)) & MASK);
}


// CAUSAL: key <= query (HF LlamaAttention.is_causal, modeling_llama.py); a wave skips key tiles that lie entirely
// above its 16 query rows, the workgroup stops at the last tile its 128 rows can see.
template <int DH, int CAUSAL, int RELKEY>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(const AttnArgs a) {
  using C = AttnCfg<DH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fq = lane >> 4, l15 = lane & 15;

  const int T = a.T;
  const float scale_log2e = a.scale_log2e;
  // Workgroups are dealt to the 8 XCDs round-robin by id, and each XCD has its own L2.  All query blocks of one (sequence,
  // head) pair read the same K / V (1.5 MiB at T = 1024, DH = 384): keep them on ONE XCD -- id % 8 picks the XCD, the ids that
  // follow each other on that XCD walk the query blocks of one pair -- so K / V come from HBM once per pair instead of once
  // per query block (the counters showed 6.8 GB fetched per launch for 0.8 GB of qkv).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int qb = slot % a.qblocks;
  const int pair = (slot / a.qblocks) * 8 + xcd;
  if (pair >= a.n_bh) return;   // grid is rounded up to whole groups of 8 pairs
  const int h = pair % a.heads_q;
  const int b = pair / a.heads_q;
  const int hk = h / a.group;
  const int64_t ld = a.ld_kv;
  const unsigned short* qbase = a.q + (int64_t)b * T * a.ld_q + (int64_t)h * DH;
  const unsigned short* kbase = a.k + (int64_t)b * T * ld + (int64_t)hk * DH;
  const unsigned short* vbase = a.v + (int64_t)b * T * ld + (int64_t)hk * DH;

  // ---- Q^T fragments: lane (q = l15, fq) holds Q[q][32 ks + 8 fq .. +7] for every k-step ----
  const int q0 = qb * 128 + wave * 16;
  const int qrow = (q0 + l15 < T) ? q0 + l15 : T - 1;
  bf16x8_t qf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) qf[ks] = *(const bf16x8_t*)(qbase + (int64_t)qrow * a.ld_q + ks * 32 + fq * 8);

  // ---- staging plan: piece i of this wave covers linear chunks [(wave + 8 i)*64, +64) of a tile.  The per-lane
  // part of the source address is a 32-bit element offset fixed for the whole kernel; the tile's first key
  // goes into the (scalar) base pointer. ----
  int st_row[C::PPW], st_src[C::PPW], st_off[C::PPW];
#pragma unroll
  for (int i = 0; i < C::PPW; ++i) {
    const int p = (wave + C::WAVES * i) * 64 + lane;
    st_row[i] = p / C::CHUNKS;
    st_src[i] = swz_chunk<DH>(p % C::CHUNKS, st_row[i]) * 8;  // element offset of the SOURCE chunk inside the row
    st_off[i] = st_row[i] * (int)ld + st_src[i];
  }
  auto stage = [&](int buf, int key0) {
    char* kdst = smem + buf * 2 * C::TILE_BYTES;
    const unsigned short* kb = kbase + (int64_t)key0 * ld;
    const unsigned short* vb = vbase + (int64_t)key0 * ld;
    const bool full = key0 + C::KV <= T;
#pragma unroll
    for (int i = 0; i < C::PPW; ++i) {
      const int piece = wave + C::WAVES * i;  // wave-uniform
      if (piece < C::PIECES) {
        int off = st_off[i];
        if (!full) {  // tail keys re-read a valid row; they are masked to -inf below
          const int row = (key0 + st_row[i] < T) ? st_row[i] : T - 1 - key0;
          off = row * (int)ld + st_src[i];
        }
        glds16(kb + off, lds_addr(kdst + piece * 1024));
        glds16(vb + off, lds_addr(kdst + C::TILE_BYTES + piece * 1024));
      }
    }
  };

  // ---- per-lane LDS read offsets.  The XOR swizzle only permutes chunks inside a segment of SEG chunks, so an
  // address is (one of a few per-lane registers) + (a compile-time offset that folds into the ds_read immediate) ----
  constexpr int SEG = C::SWZ_MASK + 1;
  // K row read (A operand of S^T): row = key = l15 (+16 for the second key tile: same row & 7), chunk = 4 ks + fq
  constexpr int KRD = SEG / 4;
  int k_rd[KRD];
#pragma unroll
  for (int i = 0; i < KRD; ++i) k_rd[i] = l15 * C::ROWB + swz_chunk<DH>(4 * i + fq, l15) * 16;
  // V transposed read (A operand of O^T): lane i of a 16-lane group supplies row (i >> 2) of a 4-key block,
  // columns 4 (i & 3) .. +3 of the 16-column d-tile; group fq reads key block 4 fq (and 16 + 4 fq).
  const int tq = l15 >> 2, tp = l15 & 3;
  constexpr int VRD = SEG / 2;
  int v_rd[VRD];
#pragma unroll
  for (int i = 0; i < VRD; ++i) {
    const int row = 4 * fq + tq;
    v_rd[i] = row * C::ROWB + swz_chunk<DH>(2 * i + (tp >> 1), row) * 16 + (tp & 1) * 8;
  }

  f32x4_t o[C::DT];
#pragma unroll
  for (int dt = 0; dt < C::DT; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  // Small heads are bound by the softmax's vector instructions, not by the matrix pipe: there the row sums of P come
  // from one more MFMA per sub-tile -- A = all-ones, so every accumulator row holds sum_key P^T[key][q] of the SAME bf16 P the
  // numerator uses -- instead of 8 adds per sub-tile and lane, and the final cross-lane reduction disappears as well.
  constexpr bool LSUM_MFMA = C::NSUB > 1;
  f32x4_t lacc = {0.f, 0.f, 0.f, 0.f};
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;

  int kv_end = T;  // keys this workgroup can see
  if (CAUSAL) { const int last_q = qb * 128 + 127; kv_end = (last_q + 1 < T) ? last_q + 1 : T; }
  const int ntiles = (kv_end + C::KV - 1) / C::KV;
  // LDS-DMA loads issued by this wave per staged tile (K and V pieces): the 3-slot pipeline leaves exactly one tile in flight
  int my_glds = 0;
#pragma unroll
  for (int i = 0; i < C::PPW; ++i) my_glds += (wave + C::WAVES * i < C::PIECES) ? 2 : 0;
  auto wait_keep_one_tile = [&]() {  // s_waitcnt needs an immediate: my_glds is 0, 2, 4 or 6 (wave-uniform)
    if (my_glds >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (my_glds == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (my_glds == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(C::NBUF == 2 || C::PPW <= 3, "wait_keep_one_tile handles up to 6 loads per tile and wave");
  stage(0, 0);
  if (C::NBUF == 3 && ntiles > 1) { stage(1, C::KV); wait_keep_one_tile(); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int cur = t % C::NBUF;
    if (C::NBUF == 2 && t + 1 < ntiles) stage(cur ^ 1, (t + 1) * C::KV);
    const char* kt = smem + cur * 2 * C::TILE_BYTES;
    const char* vt = kt + C::TILE_BYTES;
    if (!CAUSAL || t * C::KV <= q0 + 15) {  // wave-uniform: some key of this tile is visible to some row of this wave

    // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]: per 32-key sub-tile two accumulators (keys 0-15, 16-31) share each Q fragment ----
    float sv[C::NSUB][8];
    float pmax = -INFINITY;
    const int lim = CAUSAL ? ((q0 + l15 + 1 < T) ? q0 + l15 + 1 : T) : T;  // first masked key for this lane's query
#pragma unroll
    for (int u = 0; u < C::NSUB; ++u) {
      const int sub_key0 = t * C::KV + u * 32;
      if (sub_key0 >= T || (CAUSAL && sub_key0 > q0 + 15)) {  // wave-uniform: nothing visible in this sub-tile
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[u][j] = -INFINITY;
        continue;
      }
      f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        const int off = k_rd[ks % KRD] + (ks / KRD) * SEG * 16 + u * 32 * C::ROWB;
        const bf16x8_t ka = *(const bf16x8_t*)(kt + off);
        const bf16x8_t kb2 = *(const bf16x8_t*)(kt + off + 16 * C::ROWB);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb2, qf[ks], s1, 0, 0, 0);
      }
      // accumulator map: s0[r] = S^T[key = 4 fq + r][q = l15], s1[r] = S^T[key = 16 + 4 fq + r][q]
      const int key_base = sub_key0 + 4 * fq;
      if (RELKEY) {
        const float* qrow_e = a.qe + ((int64_t)b * T + qrow) * a.ld_qe + h * a.qe_stride_h + a.rel_left;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int d0 = key_base + r - qrow, d1 = key_base + 16 + r - qrow;
          d0 = d0 < -a.rel_left ? -a.rel_left : (d0 > a.rel_right ? a.rel_right : d0);
          d1 = d1 < -a.rel_left ? -a.rel_left : (d1 > a.rel_right ? a.rel_right : d1);
          s0[r] += qrow_e[d0];
          s1[r] += qrow_e[d1];
        }
      }
      // sv keeps the RAW scores: the softmax scale rides in the exponent's fma below (scale > 0, so the maximum commutes with it).
      // Only a sub-tile that crosses the end of the sequence or the causal diagonal of this wave's rows pays for masking
      // (wave-uniform test; the per-element compare + select was a quarter of the loop's vector instructions at DH = 64).
      const bool edge = sub_key0 + 32 > T || (CAUSAL && sub_key0 + 31 > q0);
      if (edge) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[u][r] = (key_base + r < lim) ? s0[r] : -INFINITY;
          sv[u][4 + r] = (key_base + 16 + r < lim) ? s1[r] : -INFINITY;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) { sv[u][r] = s0[r]; sv[u][4 + r] = s1[r]; }
      }
      // first read of this sub-tile's MFMA results by an instruction the compiler sees, so that it pads the MFMA-write -> VALU-read
      // hazard; the inline-asm v_max3 links depend on it through pmax (see attention_d64.hip softmax_head)
      pmax = fmaxf(pmax, fmaxf(sv[u][0], sv[u][4]));
#pragma unroll
      for (int r = 1; r < 4; ++r) pmax = max3f(pmax, sv[u][r], sv[u][4 + r]);
    }
    // the other three 16-lane groups hold the other keys of this query
    pmax = fmaxf(pmax, __shfl_xor(pmax, 16, 64));
    pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
    pmax *= scale_log2e;
    // deferred max: rescale only when some row's maximum moved by more than 2^8
    if (!__all(pmax - m_run <= 8.0f)) {
      const float m_new = fmaxf(m_run, pmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
      if (LSUM_MFMA) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lacc[r] *= alpha;
      }
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
    }
    bf16x8_t pf[C::NSUB];
    float psum = 0.f;
#pragma unroll
    for (int u = 0; u < C::NSUB; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sv[u][j], scale_log2e, -m_run));
        if (!LSUM_MFMA) psum += p;
        pf[u][j] = (short)f32_to_bf16(p);
      }
    l_run += psum;

    // ---- O^T[d][q] += sum_key V[key][d] P^T[key][q] ----
#pragma unroll
    for (int u = 0; u < C::NSUB; ++u) {
      const int sub_key0 = t * C::KV + u * 32;
      if (sub_key0 >= T || (CAUSAL && sub_key0 > q0 + 15)) continue;  // P is all zero there
      if (LSUM_MFMA) lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[u], lacc, 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        // d-tile dt covers columns 16 dt .. +15 = chunks 2 dt, 2 dt + 1
        const int off = v_rd[dt % VRD] + (dt / VRD) * SEG * 16 + u * 32 * C::ROWB;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vt + off));
        const s16x4_t hi =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vt + off + 16 * C::ROWB));
        bf16x8_t vf;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[u], o[dt], 0, 0, 0);
      }
    }
    }  // visible tile
    if (C::NBUF == 3) {
      // stage tile t + 2 into the slot tile t - 1 occupied (every wave left it at the previous barrier), then wait for tile
      // t + 1 only: the loads of tile t + 2 stay in flight across the barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS reads of tile t are complete
      if (t + 2 < ntiles) { stage((t + 2) % C::NBUF, (t + 2) * C::KV); wait_keep_one_tile(); }
      else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- normalise and write: O^T[d = 16 dt + 4 fq + r][q = l15] -> out[q][h*DH + d], 4 consecutive d (8 bytes) per store ----
  float l_tot;
  if (LSUM_MFMA) {
    l_tot = lacc[0];
  } else {
    l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  const int q = q0 + l15;
  if (q < T) {
    unsigned short* orow = a.out + ((int64_t)b * T + q) * a.ld_out + (int64_t)h * DH + 4 * fq;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      u16x4_t pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = f32_to_bf16(o[dt][e] * inv);
      *(u16x4_t*)(orow + 16 * dt) = pk;
    }
  }
}

// =====================================================================================================================
// DH = 384, bidirectional (the TRIBE encoder, 10 % of the step): ONE wave per SIMD, 32 query rows per wave.
//
// The 16-row kernel above reads every K / V fragment (1 KiB) from LDS for ONE 16x16x32 MFMA: LDS bytes per flop are at the
// array's limit and, at 242 VGPRs, the compiler keeps one or two fragments in flight, so every MFMA waits out most of an LDS
// round trip (24 % of the MFMA peak, profiles/r01_n_*).  Here a wave owns 32 query rows and v_mfma_f32_32x32x16_bf16: every
// fragment feeds twice the flops, and the wave has the whole 512-register file:
//   * O^T [384 x 32 queries] = 12 accumulator tiles of 32x32 = 192 registers, pinned to AGPRs by the "+a" operands of the
//     inline-asm MFMAs (the compiler would otherwise let them compete with Q for the 256 architectural VGPRs and spill);
//   * Q^T fragments for all 24 k-steps in 96 VGPRs; S^T (16) / P^T (8) / the fragment pipeline (DEPTH x 4) in the rest;
//   * the K / V fragment reads are software-pipelined DEPTH deep in SOURCE order, pinned with sched_barrier; the compiler
//     still counts them (plain loads), so every MFMA waits with the exact lgkmcnt;
//   * S^T = K Q^T puts the query on the lane (col = lane & 31): softmax state is one scalar per lane, the other 16 keys of a
//     query sit in lane ^ 32 (v_permlane32_swap); the S^T accumulator IS the B operand of O^T += V^T P^T (guide section 3:
//     element j of lane half h of k-step s is key 16 s + 8 (j >> 2) + 4 h + (j & 3)), V^T read with ds_read_b64_tr_b16 in
//     exactly that key order.
// LDS image: plain 768-byte rows, 16-byte chunks swizzled inside each 256-byte segment by
// chunk ^= ((row & 3) << 2) | ((row >> 2) & 3) (guide T10 image (b)): conflict-free for the 32-row ds_read_b128 of K (16 lanes
// of a read group see 16 distinct row & 15) and for the transposed reads of V (a 32-lane half reads 4 keys x 64 bytes, the
// keys' row & 3 spread them over four chunk quads).  K / V tiles of 32 keys arrive by LDS-DMA (inline asm, see stage_piece)
// into a 2-slot K ring and a 4-slot V ring, six 1-KiB pieces per wave and PHASE: the L1 -> LDS path moves ~37 B/clk/CU when
// bursts queue up (12 pieces in a row held the wave's issue for 1300 cycles, as long as the tile's 48 MFMAs).
// One barrier per tile, between softmax and the P V product: at that point K and V of tile t + 1 have landed (issued a phase
// and a tile earlier), so the first K fragments of tile t + 1 are prefetched under the last P V MFMAs of tile t, the first V
// fragments under the last S^T MFMAs, and the matrix pipe does not drain at the seams.
// Diagnostic build only (-DTRIBE_ATTN_STAMPS, scripts/attn_stamps.py): s_memtime stamps at the phase boundaries of the key loop,
// summed per wave into a side buffer whose pointer rides in desc.rel_qe.  Never quote the run time of that build; read the shares.

struct WideCfg {
  static constexpr int DH = 384, KS = 24, DT = 12, ROWB = 768, KV = 32, CHUNKS = 48;
  static constexpr int TILE_BYTES = KV * ROWB;       // 24 KiB per K or V tile
  static constexpr int WAVES = 4, PPW = 6;           // 24 LDS-DMA pieces of 1 KiB per K or V tile, 6 per wave
  static constexpr int K_SLOTS = 2, V_SLOTS = 4;     // see the ring schedule in the kernel
  static constexpr int V_BASE = K_SLOTS * TILE_BYTES;
  static constexpr int SMEM = (K_SLOTS + V_SLOTS) * TILE_BYTES;   // 144 KiB
  static constexpr int DEPTH = 6;                    // fragments in flight per wave
};

__device__ __forceinline__ int swz_wide(int chunk, int row) {
  return (chunk & ~15) | ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) & 15);
}


// S^T accumulation in VGPRs (asm, so that the compiler never allocates an accumulator register of its own: a0..a191 belong to
// O^T).  hipcc does not know these are MFMAs: the caller pads the result -> VALU hazard (s_nop after the last one).
__device__ __forceinline__ void mfma_s_first(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_s(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}


__global__ __launch_bounds__(256, 1) void attn_fwd_wide384_kernel(const AttnArgs a) {
  using C = WideCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r31 = lane & 31, h = lane >> 5;

  const int T = a.T;
  const float scale_log2e = a.scale_log2e;
  // same XCD-aware walk as the 16-row kernel: all query blocks of one (sequence, head) pair run on one XCD
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

  // ---- Q^T fragments (B operand of S^T): lane (query r31, half h) holds Q[q][16 ks + 8 h .. +7] ----
  const int q0 = qb * 128 + wave * 32;
  const int qrow = (q0 + r31 < T) ? q0 + r31 : T - 1;
  bf16x8_t qf[C::KS];
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) qf[ks] = *(const bf16x8_t*)(qbase + (int64_t)qrow * a.ld_q + ks * 16 + h * 8);
  if (a.q_cos) {   // wave-uniform: rotate the first q_rot_dim dims of this lane's query row (x_transformers partial rotary)
    const int half = a.q_rot_dim >> 1;
    const float* cr = a.q_cos + (int64_t)qrow * half;
    const float* sr = a.q_sin + (int64_t)qrow * half;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const int d0 = ks * 16 + h * 8;                 // this lane's 8 dims of k-step ks = pairs d0 / 2 .. d0 / 2 + 3
      if (d0 < a.q_rot_dim) {
        const float4 c4 = *(const float4*)(cr + (d0 >> 1)), s4 = *(const float4*)(sr + (d0 >> 1));
        const float cs[4] = {c4.x, c4.y, c4.z, c4.w}, sn[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const float qa = bf16_to_f32((unsigned short)qf[ks][2 * pp]), qb = bf16_to_f32((unsigned short)qf[ks][2 * pp + 1]);
          qf[ks][2 * pp] = (short)f32_to_bf16(qa * cs[pp] - qb * sn[pp]);
          qf[ks][2 * pp + 1] = (short)f32_to_bf16(qb * cs[pp] + qa * sn[pp]);
        }
      }
    }
  }

  // ---- staging plan: piece i of this wave covers linear 16-byte chunks [(wave + 4 i) * 64, +64) of a tile ----
  int st_row[C::PPW], st_src[C::PPW], st_off[C::PPW];
#pragma unroll
  for (int i = 0; i < C::PPW; ++i) {
    const int p = (wave + C::WAVES * i) * 64 + lane;
    st_row[i] = p / C::CHUNKS;
    st_src[i] = swz_wide(p % C::CHUNKS, st_row[i]) * 8;   // element offset of the SOURCE chunk inside the row
    st_off[i] = st_row[i] * (int)ld + st_src[i];
  }
  // One 1-KiB piece of the K (which = 0) or V (which = 1) tile that starts at key0, into ring slot `slot` of that operand.
  // Round 3: a lone wave per SIMD pays for every LDS-DMA piece with its own issue time, so the piece is as cheap as the ISA allows (the
  // lesson of the one-wave-per-SIMD GEMM, gemm.hip): SCALAR tile base + the lane's fixed 32-bit byte offset (no 64-bit vector add per
  // piece), M0 written and not saved / restored (nothing else in the kernel uses it).
  const unsigned lds0_u = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  auto stage_piece = [&](int which, int slot, int key0, int i) {
    const char* sbase = (const char*)((which ? vbase : kbase) + (int64_t)key0 * ld);
    const int piece = wave + C::WAVES * i;
    unsigned voff = (unsigned)st_off[i] * 2u;
    if (key0 + C::KV > T) {   // wave-uniform: tail keys (key0 + row >= T) re-read the last valid row; they are masked to -inf in the softmax
      const int row = (key0 + st_row[i] < T) ? st_row[i] : T - 1 - key0;
      voff = (unsigned)(row * (int)ld + st_src[i]) * 2u;
    }
    const unsigned dst = lds0_u + (which ? C::V_BASE : 0) + slot * C::TILE_BYTES + piece * 1024;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(voff), "s"(sbase) : "memory");
  };
  auto stage = [&](int which, int slot, int key0) {
#pragma unroll
    for (int i = 0; i < C::PPW; ++i) stage_piece(which, slot, key0, i);
  };

  // ---- per-lane LDS byte offsets inside a slot: 8 for the K row reads, 8 for the V transposed reads; everything else is an
  // immediate (segment, k-step half) or the slot base added once per tile ----
  int kb_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) kb_off[i] = r31 * C::ROWB + swz_wide(2 * i + h, r31) * 16;
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int vb_off[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int lh = 0; lh < 2; ++lh) {
      const int row = 4 * h + tq + 8 * lh;
      vb_off[i][lh] = C::V_BASE + row * C::ROWB + swz_wide(4 * i + 2 * g1 + (tp >> 1), row) * 16 + (tp & 1) * 8;
    }
  auto ld_k = [&](const int (&ka)[8], int ks) -> bf16x8_t {
    return *(const bf16x8_t*)(smem + ka[ks & 7] + (ks >> 3) * 256);
  };
  auto ld_v = [&](const int (&va)[4][2], int i) -> bf16x8_t {   // fragment i = 12 s + dt
    const int s = i / C::DT, dt = i % C::DT;
    const int imm = (dt >> 2) * 256 + s * 16 * C::ROWB;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + va[dt & 3][0] + imm));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + va[dt & 3][1] + imm));
    bf16x8_t vf;
#pragma unroll
    for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
    return vf;
  };

  static_for<0, C::DT>([&](auto dt) { AccTile<decltype(dt)::value>::zero(); });   // O^T = 0 in a[0:191]
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (T + C::KV - 1) / C::KV;
  // Ring schedule (6 LDS-DMA loads per wave and PHASE, so the L1 -> LDS path sees an even 32 B/clk instead of bursts):
  //   S^T phase of tile t issues V(t + 2) into V slot (t + 2) % 4  (V(t - 2) was consumed before barrier t - 1);
  //   P V phase of tile t issues K(t + 2) into K slot t % 2        (K(t) was consumed before barrier t);
  //   the wait before barrier t leaves the 6 youngest loads (V(t + 2)) in flight: K(t + 1) and V(t + 1), issued a phase and a
  //   tile earlier, have landed and become visible to every wave -- K(t + 1) for the prefetch under the last P V MFMAs of tile
  //   t, V(t + 1) for the prefetch under the last S^T MFMAs of tile t + 1.
  stage(0, 0, 0); stage(1, 0, 0);
  if (ntiles > 1) { stage(0, 1, C::KV); stage(1, 1, C::KV); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
  else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __builtin_amdgcn_s_barrier();

  bf16x8_t kf[C::KS], vf[2 * C::DT];
  // The 16 read addresses live in registers for the whole kernel and are STEPPED from slot to slot by a scalar (opaque asm: the
  // compiler otherwise precomputes one copy per slot, 48 registers, and spills the Q fragments to make room).
  int ka[8], va[4][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) ka[i] = kb_off[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) { va[i][0] = vb_off[i][0]; va[i][1] = vb_off[i][1]; }
#pragma unroll
  for (int d = 0; d < C::DEPTH; ++d) kf[d] = ld_k(ka, d);

#ifdef TRIBE_ATTN_STAMPS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;
  unsigned long long stamp_acc[5] = {0, 0, 0, 0, 0};   // 0 S^T phase, 1 softmax, 2 vmcnt wait, 3 barrier, 4 P V phase
#endif
  int vcur = 0;            // V slot of tile t
  for (int t = 0; t < ntiles; ++t) {
    ATTN_STAMP(ts0);
    const bool more = t + 2 < ntiles;
    const int vnext2 = (vcur + 2) & 3;

    // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]: 24 k-steps of 16, K fragments DEPTH ahead, then the first V fragments ----
    f32x16_t s;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      if (ks + C::DEPTH < C::KS) kf[ks + C::DEPTH] = ld_k(ka, ks + C::DEPTH);
      else vf[ks + C::DEPTH - C::KS] = ld_v(va, ks + C::DEPTH - C::KS);
      if (ks % 4 == 1) { if (more) stage_piece(1, vnext2, (t + 2) * C::KV, ks / 4); }
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0) mfma_s_first(s, kf[ks], qf[ks]);
      else mfma_s(s, kf[ks], qf[ks]);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 11" : "+v"(s));   // 8-pass MFMA result -> first VALU read (hipcc pads its own MFMAs with 12 states here)
    ATTN_STAMP(ts1);

    // ---- online softmax: s[r] = S^T[key = 32 t + (r & 3) + 8 (r >> 2) + 4 h][q = r31], raw scores ----
    if ((t + 1) * C::KV > T) {   // wave-uniform: the last tile of a sequence whose length is not a multiple of 32
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (t * C::KV + (r & 3) + 8 * (r >> 2) + 4 * h >= T) s[r] = -INFINITY;
    }
    float pmax = max3f(s[0], s[1], s[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) pmax = max3f(pmax, s[r], s[r + 1]);
    pmax = fmaxf(pmax, s[15]);
    pmax = pair_max(pmax) * scale_log2e;
    if (!__all(pmax - m_run <= 8.0f)) {   // deferred max (guide T13): all P V products of earlier tiles are complete here
      const float m_new = fmaxf(m_run, pmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
      if (t > 0) {   // at t = 0 O^T is still zero
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last P V MFMA -> v_accvgpr_read (24 S^T MFMAs ran in between; belt and braces)
        static_for<0, C::DT>([&](auto dt) { AccTile<decltype(dt)::value>::scale(alpha); });
        asm volatile("s_nop 7" ::: "memory");               // v_accvgpr_write -> MFMA C operand
      }
    }
    bf16x8_t pf[2];
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_run));
      psum += p;
      pf[r >> 3][r & 7] = (short)f32_to_bf16(p);
    }
    l_run += psum;

    // ---- K(t + 1) and V(t + 1) have landed for everyone after this barrier; every wave is done with K(t) and V(t - 1) ----
    ATTN_STAMP(ts2);
    if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // the 6 youngest = this wave's V(t + 2) pieces, just issued
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(ts3);
    __builtin_amdgcn_s_barrier();
    ATTN_STAMP(ts4);
    // K read addresses flip to the other K slot (tile t + 1)
    const int kslot = t & 1;
    const int kstep = kslot ? -C::TILE_BYTES : C::TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(ka[i]) : "s"(kstep));
    asm volatile("s_nop 1" : "+v"(pf[0]), "+v"(pf[1]));   // VALU-written P fragments -> MFMA operands inside asm

    // ---- O^T[d][q] += sum_key V[key][d] P^T[key][q]: 2 key halves x 12 d-tiles; the first K fragments of tile t + 1 ride
    // under the last MFMAs (harmless stale reads after the last tile) ----
    static_for<0, 2 * C::DT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i + C::DEPTH < 2 * C::DT) vf[i + C::DEPTH] = ld_v(va, i + C::DEPTH);
      else kf[i + C::DEPTH - 2 * C::DT] = ld_k(ka, i + C::DEPTH - 2 * C::DT);
      if constexpr (i % 4 == 1) { if (more) stage_piece(0, kslot, (t + 2) * C::KV, i / 4); }
      __builtin_amdgcn_sched_barrier(0);
      AccTile<i % C::DT>::mfma(vf[i], pf[i / C::DT]);
      __builtin_amdgcn_sched_barrier(0);
    });
    // V read addresses move on to the next of the four V slots
    const int vstep = (vcur == 3) ? -3 * C::TILE_BYTES : C::TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm volatile("v_add_u32 %0, %1, %0" : "+v"(va[i][0]) : "s"(vstep));
      asm volatile("v_add_u32 %0, %1, %0" : "+v"(va[i][1]) : "s"(vstep));
    }
    vcur = (vcur + 1) & 3;
    ATTN_STAMP(ts5);
    ATTN_STAMP_ADD(0, ts0, ts1); ATTN_STAMP_ADD(1, ts1, ts2); ATTN_STAMP_ADD(2, ts2, ts3); ATTN_STAMP_ADD(3, ts3, ts4); ATTN_STAMP_ADD(4, ts4, ts5);
  }
#ifdef TRIBE_ATTN_STAMPS
  if (a.qe != nullptr && lane == 0) {
    unsigned long long* dbg = (unsigned long long*)a.qe + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 5; ++i) dbg[i] = stamp_acc[i];
    dbg[5] = (unsigned long long)ntiles;
  }
#endif

  // ---- normalise and write: O^T[d = 32 dt + (r & 3) + 8 (r >> 2) + 4 h][q = r31] -> out[q][hd * DH + d] ----
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  const float l_tot = pair_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r31;
  // what a backward pass needs to rebuild P = exp2(s * scale * log2 e - lse2) without a softmax (TRIBE_ACT_EXP2): one float per query
  if (a.lse != nullptr && h == 0 && q < T) a.lse[((int64_t)b * a.heads_q + hd) * T + q] = m_run + __builtin_amdgcn_logf(l_tot);
  unsigned short* orow = a.out + ((int64_t)b * T + (q < T ? q : T - 1)) * a.ld_out + (int64_t)hd * C::DH + 4 * h;
  static_for<0, C::DT>([&](auto dtc) {
    constexpr int dt = decltype(dtc)::value;
    float ov[16];
    AccTile<dt>::read(ov);
    if (q < T) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u16x4_t pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = f32_to_bf16(ov[4 * g + e] * inv);
        *(u16x4_t*)(orow + 32 * dt + 8 * g) = pk;
      }
    }
  });
}

// =====================================================================================================================
// DH = 384, bidirectional, KEY-SPLIT variant: 8 waves = two per SIMD.
//
// The one-wave-per-SIMD kernel above pays for every LDS-DMA piece with ~105 cycles of its single in-order instruction stream
// (stamps: 12 pieces per tile = 1260 of 3770 cycles; bursts, even spreading, early issue, per-wave skew and a buffer descriptor
// all cost the same) because nothing else can issue on the SIMD meanwhile.  Two waves per SIMD hide that under the partner's
// MFMAs, but a wave then has 256 registers, too few for 32 query rows x 384 (O^T alone is 192).  So the head dimension is split
// between the two waves of a pair: wave (j, c) owns query block j (32 rows) and the half c of the head dimension whose 128-byte
// chunk index ((d >> 6) & 1) equals c:
//   * S^T partial = K[:, half c] . Q^T[half c]: 12 MFMAs; the partials are exchanged through LDS (4 KiB per wave, f32) and
//     summed, so both waves hold the same full S^T (a + b == b + a in IEEE) and run the same softmax;
//   * O^T[half c] += V^T[half c] . P^T: 12 MFMAs into 6 accumulator tiles (96 AGPRs); Q^T[half c] sits in 32 more AGPRs (MFMA B
//     operands may be accumulator registers; 8 of the 12 fragments, a[96:127], the other 4 in VGPRs), 128 VGPRs for the rest.
// MEASURED (profiles/r02_g_attn_bench.json): 1.14-1.22 ms at B = 64 across three ring designs, the same as the
// one-wave kernel (1.13-1.17 ms).  With every MFMA and fragment read compiled out (-DTRIBE_ATTN_DMA_ONLY) it runs 0.83 ms: the
// K / V stream of a 128-query workgroup (48 KiB per 32-key tile, ~30 GB/s per CU) is the floor, and the matrix work adds its
// 0.3 ms on top of it instead of under it.  Kept as the selectable variant (tribe_attention_set_mode(3)).
// Same LDS image, operand maps and AGPR ownership rules as the kernel above.  Unified K / V ring of 5 slots + exchange buffer
// 32 KiB: 152 KiB.  Two barriers per tile: after the partial S^T (publishes the partials and the half-tiles waited for) and after P V.
struct KSplitCfg {
  static constexpr int DH = 384, ROWB = 768, KV = 32, CHUNKS = 48;
  static constexpr int KS = 12, NT = 6, Q_AGPR = 8;    // k-steps of 16 and 32-row O^T tiles PER WAVE; Q^T fragments kept in AGPRs
  static constexpr int TILE_BYTES = KV * ROWB;
  static constexpr int WAVES = 8, PPW = 6;             // 24 pieces per K or V half-tile, 6 per wave of the half that moves it
  static constexpr int SLOTS = 5;                      // unified K / V ring (see the kernel)
  static constexpr int X_BASE = SLOTS * TILE_BYTES;    // exchange buffer: 8 waves x 4 KiB
  static constexpr int SMEM = X_BASE + WAVES * 4096;   // 152 KiB
  static constexpr int DEPTH = 4;
};

__global__ __launch_bounds__(512, 2) void attn_fwd_ksplit384_kernel(const AttnArgs a) {
  using C = KSplitCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qj = wave & 3, c = wave >> 2;              // query block, head-dimension half
  const int r31 = lane & 31, h = lane >> 5;

  const int T = a.T;
  const float scale_log2e = a.scale_log2e;
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

  // ---- Q^T fragments of this wave's half: local k-step k' = 4 s + i is global k-step 8 s + 4 c + i (d = 16 ks + 8 h ..) ----
  const int q0 = qb * 128 + qj * 32;
  const int qrow = (q0 + r31 < T) ? q0 + r31 : T - 1;
  bf16x8_t qv[C::KS - C::Q_AGPR];   // a wave at two per SIMD addresses 128 accumulator registers: 8 fragments there, 4 in VGPRs
  static_for<0, C::KS>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int ksg = 8 * (k >> 2) + 4 * c + (k & 3);
    const bf16x8_t qfrag = *(const bf16x8_t*)(qbase + (int64_t)qrow * a.ld_q + ksg * 16 + h * 8);
    if constexpr (k < C::Q_AGPR) QFrag<k>::set(qfrag);
    else qv[k - C::Q_AGPR] = qfrag;
  });
  static_for<0, C::NT>([&](auto n) { AccTile<decltype(n)::value>::zero(); });
  asm volatile("s_nop 7" ::: "memory");   // v_accvgpr_write -> MFMA operand

  // ---- staging plan.  The waves of half c = 0 move ALL K half-tiles, those of c = 1 all V half-tiles, 6 pieces of 1 KiB per wave
  // (piece = qj + 4 i): the two waves of a SIMD (same qj, c = 0 / 1) then never issue LDS-DMA in the same phase -- K pieces go out
  // during S^T phases, V pieces during P V phases -- so one wave's ~100-cycle LDS-DMA issues overlap the other's MFMAs instead of
  // colliding with its own partner's (both waves run the same code in lockstep; with pieces split evenly the kernel ran
  // 0.83 ms WITHOUT any MFMA and 1.16 ms with them). ----
  int st_off[C::PPW];
#pragma unroll
  for (int i = 0; i < C::PPW; ++i) {
    const int p = (qj + 4 * i) * 64 + lane;
    st_off[i] = (p / C::CHUNKS) * (int)ld + swz_wide(p % C::CHUNKS, p / C::CHUNKS) * 8;
  }
  const unsigned short* my_base = c ? vbase : kbase;
  auto stage_piece = [&](int slot_i, int key0, int i) {
    const unsigned short* src = my_base + (int64_t)key0 * ld;
    const int piece = qj + 4 * i;
    int off = st_off[i];
    if (key0 + C::KV > T) {   // tail keys re-read the last valid row; they are masked to -inf in the softmax
      const int p = piece * 64 + lane;
      const int row0 = p / C::CHUNKS;
      const int row = (key0 + row0 < T) ? row0 : T - 1 - key0;
      off = row * (int)ld + swz_wide(p % C::CHUNKS, row0) * 8;
    }
    glds16(src + off, lds_addr(smem) + slot_i * C::TILE_BYTES + piece * 1024);
  };

  // ---- per-lane LDS read offsets: K row reads (4 registers: chunk = 16 s + 8 c + 2 i + h) and V transposed reads (4 registers:
  // d-tile dt = 4 u + 2 c + e, chunk = 4 (2 c + e) + 2 g1 + pp inside segment u) ----
  int ka[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ka[i] = r31 * C::ROWB + swz_wide(8 * c + 2 * i + h, r31) * 16;
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int va[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int lh = 0; lh < 2; ++lh) {
      const int row = 4 * h + tq + 8 * lh;
      va[e][lh] = C::TILE_BYTES + row * C::ROWB + swz_wide(4 * (2 * c + e) + 2 * g1 + (tp >> 1), row) * 16 + (tp & 1) * 8;
    }
  auto ld_k = [&](int k) -> bf16x8_t { return *(const bf16x8_t*)(smem + ka[k & 3] + (k >> 2) * 256); };
  auto ld_v = [&](int i) -> bf16x8_t {   // fragment i = 6 s_key + n, tile n = 2 u + e
    const int sk = i / C::NT, n = i % C::NT;
    const int imm = (n >> 1) * 256 + sk * 16 * C::ROWB;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + va[n & 1][0] + imm));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + va[n & 1][1] + imm));
    bf16x8_t vf;
#pragma unroll
    for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
    return vf;
  };
  const int x_mine = C::X_BASE + wave * 4096 + lane * 16, x_peer = C::X_BASE + (wave ^ 4) * 4096 + lane * 16;

  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (T + C::KV - 1) / C::KV;
  // ---- unified ring of five 24-KiB slots: half-tile hh (even: K(hh / 2), odd: V(hh / 2)) lives in slot hh % 5 and is consumed in
  // phase hh (S^T of tile t = phase 2 t, P V = phase 2 t + 1).  Two barriers per tile: B1 after the partial S^T (publishes the
  // partials, frees K(t)'s slot), B2 after P V (frees V(t)'s slot); a freed slot is refilled at once with the half-tile five
  // ahead -- V(t + 2) after B1(t), K(t + 3) after B2(t) -- and waited for just before the B1 that precedes its first read, three
  // to four phases later.  (With one barrier per tile either K or V had ~1.2 phases between issue and wait, less than the loaded
  // L2 latency, and the wait showed up in every tile.) ----
  auto stage_half = [&](int hh) {   // the six pieces this wave owns of half-tile hh (K half-tiles: c = 0 waves, V: c = 1)
    if ((hh & 1) != c) return;
#pragma unroll
    for (int i = 0; i < C::PPW; ++i) stage_piece(hh % C::SLOTS, (hh >> 1) * C::KV, i);
  };
  const int nhalf = 2 * ntiles;
  for (int hh = 0; hh < C::SLOTS && hh < nhalf; ++hh) stage_half(hh);            // K0 V0 K1 V1 K2
  // K(0) must have landed: a c = 0 wave has K(1), K(2) behind it in its queue (V waves wait for nothing here: V(0) is waited
  // for before the first B1)
  if (c == 0) {
    if (ntiles >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (ntiles == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  bf16x8_t kf[C::KS], vf[2 * C::NT];
#pragma unroll
  for (int d = 0; d < C::DEPTH; ++d) kf[d] = ld_k(d);

  int kslot = 0, vslot = 1;   // ring slots of K(t) and V(t)
  for (int t = 0; t < ntiles; ++t) {
    // ---- partial S^T over this wave's half of the head dimension ----
    f32x16_t s;
    static_for<0, C::KS>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
#ifndef TRIBE_ATTN_DMA_ONLY
      if constexpr (k + C::DEPTH < C::KS) kf[k + C::DEPTH] = ld_k(k + C::DEPTH);
#endif
      // K(t + 2) into the slot V(t - 1) left (B2 of the previous tile freed it): the K waves, one piece every second MFMA
      if constexpr (k % 2 == 1) { if (c == 0 && t >= 1 && t + 2 < ntiles) stage_piece((2 * t + 4) % C::SLOTS, (t + 2) * C::KV, k / 2); }
      __builtin_amdgcn_sched_barrier(0);
#ifndef TRIBE_ATTN_DMA_ONLY
      if constexpr (k == 0) QFrag<k>::mfma_first(s, kf[k]);
      else if constexpr (k < C::Q_AGPR) QFrag<k>::mfma(s, kf[k]);
      else mfma_s(s, kf[k], qv[k - C::Q_AGPR]);
#else
      if constexpr (k == 0) { for (int r = 0; r < 16; ++r) s[r] = (float)r31; }
#endif
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("s_nop 11" : "+v"(s));
    // ---- hand the partial to the partner wave (the other half of the head dimension, same query block) ----
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) *(f32x4_t*)(smem + x_mine + q4 * 1024) = f32x4_t{s[4 * q4], s[4 * q4 + 1], s[4 * q4 + 2], s[4 * q4 + 3]};
    // V(t) and K(t + 1) must have landed (first read after this barrier).  A K wave has K(t + 2) behind K(t + 1) in its queue (when
    // it exists), a V wave V(t + 1) behind V(t)
    if (c == 0 ? (t + 2 < ntiles) : (t + 1 < ntiles)) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // B1(t)
#pragma unroll
    for (int d = 0; d < C::DEPTH; ++d) vf[d] = ld_v(d);            // the first V fragments fly under the softmax
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x4_t o4 = *(const f32x4_t*)(smem + x_peer + q4 * 1024);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[4 * q4 + e] += o4[e];
    }

    // ---- online softmax (identical in both waves of the pair) ----
    if ((t + 1) * C::KV > T) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (t * C::KV + (r & 3) + 8 * (r >> 2) + 4 * h >= T) s[r] = -INFINITY;
    }
    float pmax = max3f(s[0], s[1], s[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) pmax = max3f(pmax, s[r], s[r + 1]);
    pmax = fmaxf(pmax, s[15]);
    pmax = pair_max(pmax) * scale_log2e;
    if (!__all(pmax - m_run <= 8.0f)) {
      const float m_new = fmaxf(m_run, pmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
      if (t > 0) {
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        static_for<0, C::NT>([&](auto n) { AccTile<decltype(n)::value>::scale(alpha); });
        asm volatile("s_nop 7" ::: "memory");
      }
    }
    bf16x8_t pf[2];
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, -m_run));
      psum += p;
      pf[r >> 3][r & 7] = (short)f32_to_bf16(p);
    }
    l_run += psum;

    // K read addresses move on to K(t + 1)'s slot
    const int knext = kslot + 2 >= C::SLOTS ? kslot + 2 - C::SLOTS : kslot + 2;
    const int kstep = (knext - kslot) * C::TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(ka[i]) : "s"(kstep));
    asm volatile("s_nop 1" : "+v"(pf[0]), "+v"(pf[1]));

    // ---- O^T[half c] += V^T[half c] . P^T; V(t + 2) goes into the slot K(t) left (the partner wave's MFMAs cover the issue
    // time of the pieces), the first K fragments of tile t + 1 ride under the last MFMAs ----
    static_for<0, 2 * C::NT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#ifndef TRIBE_ATTN_DMA_ONLY
      if constexpr (i + C::DEPTH < 2 * C::NT) vf[i + C::DEPTH] = ld_v(i + C::DEPTH);
      else kf[i + C::DEPTH - 2 * C::NT] = ld_k(i + C::DEPTH - 2 * C::NT);
#endif
      if constexpr (i % 2 == 1) { if (c == 1 && t + 2 < ntiles) stage_piece(kslot, (t + 2) * C::KV, i / 2); }
      __builtin_amdgcn_sched_barrier(0);
#ifndef TRIBE_ATTN_DMA_ONLY
      AccTile<i % C::NT>::mfma(vf[i], pf[i / C::NT]);
#endif
      __builtin_amdgcn_sched_barrier(0);
    });
    const int vnext = vslot + 2 >= C::SLOTS ? vslot + 2 - C::SLOTS : vslot + 2;
    const int vstep = (vnext - vslot) * C::TILE_BYTES;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      asm volatile("v_add_u32 %0, %1, %0" : "+v"(va[e][0]) : "s"(vstep));
      asm volatile("v_add_u32 %0, %1, %0" : "+v"(va[e][1]) : "s"(vstep));
    }
    kslot = knext;
    vslot = vnext;
    // B2(t): every wave is done with V(t) -- its slot takes K(t + 3) during the next S^T phase.  The V fragment reads were consumed
    // by the MFMAs above; the K(t + 1) prefetch reads may stay in flight across the barrier.
    __builtin_amdgcn_s_barrier();
  }

  // ---- normalise and write this wave's half: tile n = 2 u + e holds d = 32 (4 u + 2 c + e) + (r & 3) + 8 (r >> 2) + 4 h ----
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  const float inv = 1.0f / pair_sum(l_run);
  const int q = q0 + r31;
  unsigned short* orow = a.out + ((int64_t)b * T + (q < T ? q : T - 1)) * a.ld_out + (int64_t)hd * C::DH + 4 * h + 64 * c;
  static_for<0, C::NT>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    float ov[16];
    AccTile<n>::read(ov);
    if (q < T) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u16x4_t pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = f32_to_bf16(ov[4 * g + e] * inv);
        *(u16x4_t*)(orow + 128 * (n >> 1) + 32 * (n & 1) + 8 * g) = pk;
      }
    }
  });
}

int launch_attn_ksplit384(const AttnArgs& a, int64_t B, hipStream_t s) {
  using C = KSplitCfg;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_ksplit384_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    attr_done = true;
  }
  const int64_t nblocks = (B * a.heads_q + 7) / 8 * 8 * a.qblocks;
  if (nblocks >= (1ll << 31)) { tribe_set_error("tribe_attention_fwd: grid too large"); return -1; }
  if ((int64_t)a.T * a.ld_kv >= (1ll << 31)) { tribe_set_error("tribe_attention_fwd: sequence too long for 32-bit offsets"); return -1; }
  hipLaunchKernelGGL(attn_fwd_ksplit384_kernel, dim3((unsigned)nblocks), dim3(512), C::SMEM, s, a);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

int launch_attn_wide384(const AttnArgs& a, int64_t B, hipStream_t s) {
  using C = WideCfg;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_wide384_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    attr_done = true;
  }
  const int64_t nblocks = (B * a.heads_q + 7) / 8 * 8 * a.qblocks;
  if (nblocks >= (1ll << 31)) { tribe_set_error("tribe_attention_fwd: grid too large"); return -1; }
  if ((int64_t)a.T * a.ld_kv >= (1ll << 31)) { tribe_set_error("tribe_attention_fwd: sequence too long for 32-bit offsets"); return -1; }
  hipLaunchKernelGGL(attn_fwd_wide384_kernel, dim3((unsigned)nblocks), dim3(256), C::SMEM, s, a);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

template <int DH, int CAUSAL, int RELKEY>
int launch_attn(const AttnArgs& a, int64_t B, hipStream_t s) {
  using C = AttnCfg<DH>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<DH, CAUSAL, RELKEY>, hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
    attr_done = true;
  }
  const int64_t nblocks = (B * a.heads_q + 7) / 8 * 8 * a.qblocks;   // whole groups of 8 (sequence, head) pairs, one per XCD
  if (nblocks >= (1ll << 31)) { tribe_set_error("tribe_attention_fwd: grid too large"); return -1; }
  if ((int64_t)a.T * a.ld_kv >= (1ll << 31)) { tribe_set_error("tribe_attention_fwd: sequence too long for 32-bit offsets"); return -1; }
  hipLaunchKernelGGL((attn_fwd_kernel<DH, CAUSAL, RELKEY>), dim3((unsigned)nblocks), dim3(512), C::SMEM, s, a);
  TRIBE_LAUNCH_CHECK();
  return 0;
}

}  // namespace
int tribe_internal_attn_d64_launch(const void* args, int64_t B, int relkey, int variant, hipStream_t s);   // attention_d64.hip
namespace {

int g_attn_d64_variant = 0;   // DH = 64: 0 = by grid size, 1 = 4-wave kernel, 2 = anti-phase 8-wave kernel (attention_d64.hip)
int g_attn_wide384 = 1;   // DH = 384 variant: 1 = one wave per SIMD (32 rows x 384), 2 = key-split pairs, 0 = the 16-row kernel (tribe_attention_set_mode)

template <int DH>
int launch_attn_dh(const AttnArgs& a, int64_t B, int causal, hipStream_t s) {
  if (DH == 384 && !causal && g_attn_wide384 == 1) return launch_attn_wide384(a, B, s);
  if (DH == 384 && !causal && g_attn_wide384 == 2) return launch_attn_ksplit384(a, B, s);
  return causal ? launch_attn<DH, 1, 0>(a, B, s) : launch_attn<DH, 0, 0>(a, B, s);
}

}  // namespace

void tribe_internal_attention_set_wide384(int on) { g_attn_wide384 = on; }
void tribe_internal_attention_set_d64_variant(int v) { g_attn_d64_variant = v; }

// returns 1 if a fused kernel exists for this head size, else 0 (caller falls back to the 3-kernel path)
int tribe_internal_attention_fused_supported(int dim_head) {
  return dim_head == 64 || dim_head == 128 || dim_head == 192 || dim_head == 384;
}

// the work behind tribe_attention_fwd_ex; q_cos / q_sin / q_rot_dim: partial rotary of Q applied by the kernel itself (nullptr: q arrives
// rotated; only tribe_internal_attention_fused_qrot passes tables, and only where tribe_internal_attention_rotates_q holds)
static int attention_dispatch_impl(const tribe_attention_desc* d, const float* q_cos, const float* q_sin, int q_rot_dim, void* stream);
static int attention_dispatch(const tribe_attention_desc* d, const float* q_cos, const float* q_sin, int q_rot_dim, void* stream) {
  // profile slot of the "attention" role (bench.py's by_kernel table): QK^T and PV, 4 * T * dim_head flop per query row and head
  const double flops = d ? 4.0 * (double)d->T * (double)d->dim_head * (double)d->T * (double)d->heads_q * (double)d->B * (d->causal ? 0.5 : 1.0) : 0.0;
  const int slot = tribe_internal_prof_before(TRIBE_ROLE_ATTENTION, flops, (hipStream_t)stream);
  const int rc = attention_dispatch_impl(d, q_cos, q_sin, q_rot_dim, stream);
  tribe_internal_prof_after(slot, (hipStream_t)stream);
  return rc;
}
static int attention_dispatch_impl(const tribe_attention_desc* d, const float* q_cos, const float* q_sin, int q_rot_dim, void* stream) {
  TRIBE_REQUIRE(d && d->q && d->k && d->v && d->out, "tribe_attention_fwd_ex: null pointer");
  TRIBE_REQUIRE(d->B > 0 && d->T > 0 && d->heads_q > 0 && d->heads_kv > 0 && d->heads_q % d->heads_kv == 0,
                "tribe_attention_fwd_ex: bad shape (B=%lld T=%lld heads %d/%d)", (long long)d->B, (long long)d->T, d->heads_q, d->heads_kv);
  TRIBE_REQUIRE(tribe_internal_attention_fused_supported(d->dim_head), "tribe_attention_fwd_ex: dim_head=%d not in {64,128,192,384}",
                d->dim_head);
  TRIBE_REQUIRE(d->ld_k == d->ld_v, "tribe_attention_fwd_ex: k and v must share their row stride");
  TRIBE_REQUIRE(d->ld_q % 8 == 0 && d->ld_k % 8 == 0 && d->ld_out % 4 == 0 && ((uintptr_t)d->q % 16) == 0 && ((uintptr_t)d->k % 16) == 0 &&
                    ((uintptr_t)d->v % 16) == 0 && ((uintptr_t)d->out % 8) == 0,
                "tribe_attention_fwd_ex: operands must be 16-byte aligned with row strides that are multiples of 8");
  AttnArgs a;
  a.q = d->q; a.k = d->k; a.v = d->v; a.out = d->out;
  a.ld_q = d->ld_q; a.ld_kv = d->ld_k; a.ld_out = d->ld_out;
  a.T = (int)d->T; a.heads_q = d->heads_q; a.group = d->heads_q / d->heads_kv;
  a.scale_log2e = d->scale * 1.4426950408889634f;
  a.qblocks = (int)((d->T + 127) / 128);
  a.n_bh = (int)(d->B * d->heads_q);
  a.qe = d->rel_qe; a.ld_qe = d->ld_rel_qe; a.qe_stride_h = d->rel_stride_h; a.rel_left = d->rel_left; a.rel_right = d->rel_right;
  a.q_cos = q_cos; a.q_sin = q_sin; a.q_rot_dim = q_rot_dim;
  a.lse = d->lse;
  TRIBE_REQUIRE(!d->lse || (tribe_attention_lse_supported(d->dim_head, d->causal) && !d->rel_qe),
                "tribe_attention_fwd_ex: the log-sum-exp output exists for dim_head 384, bidirectional, default mode only");
  hipStream_t s = (hipStream_t)stream;
#ifdef TRIBE_ATTN_STAMPS
  if (d->rel_qe && d->dim_head == 384) return launch_attn_dh<384>(a, d->B, d->causal, s);   // rel_qe carries the stamp buffer
  if (d->rel_qe && d->dim_head == 64 && d->rel_left < 0) return tribe_internal_attn_d64_launch(&a, d->B, 0, g_attn_d64_variant, s);
#endif
  if (d->rel_qe) {
    TRIBE_REQUIRE(d->dim_head == 64 && !d->causal, "tribe_attention_fwd_ex: the relative_key bias is built for dim_head 64, non-causal");
    TRIBE_REQUIRE(d->rel_left >= 0 && d->rel_right >= 0 && d->rel_stride_h >= d->rel_left + d->rel_right + 1 &&
                      d->ld_rel_qe >= (int64_t)d->heads_q * d->rel_stride_h,
                  "tribe_attention_fwd_ex: bad relative_key table geometry");
    return g_attn_wide384 ? tribe_internal_attn_d64_launch(&a, d->B, 1, g_attn_d64_variant, s) : launch_attn<64, 0, 1>(a, d->B, s);
  }
  switch (d->dim_head) {
    case 64: return (!d->causal && g_attn_wide384) ? tribe_internal_attn_d64_launch(&a, d->B, 0, g_attn_d64_variant, s) : launch_attn_dh<64>(a, d->B, d->causal, s);
    case 128: return launch_attn_dh<128>(a, d->B, d->causal, s);
    case 192: return launch_attn_dh<192>(a, d->B, d->causal, s);
    default: return launch_attn_dh<384>(a, d->B, d->causal, s);
  }
}

extern "C" int tribe_attention_fwd_ex(const tribe_attention_desc* d, void* stream) { return attention_dispatch(d, nullptr, nullptr, 0, stream); }

// 1 when the next fused launch at this head size rotates Q itself if given the tables (the DH = 384 one-wave kernel)
extern "C" int tribe_attention_lse_supported(int32_t dim_head, int32_t causal) { return dim_head == 384 && !causal && g_attn_wide384 == 1; }

int tribe_internal_attention_rotates_q(int dim_head) { return dim_head == 384 && g_attn_wide384 == 1; }

static tribe_attention_desc fused_desc(const uint16_t* qkv, int64_t B, int64_t T, int heads, int dim_head, float scale, uint16_t* out) {
  const int64_t inner = (int64_t)heads * dim_head;
  tribe_attention_desc d;
  d.q = qkv; d.k = qkv + inner; d.v = qkv + 2 * inner;
  d.ld_q = d.ld_k = d.ld_v = 3 * inner;
  d.out = out; d.ld_out = inner;
  d.B = B; d.T = T; d.heads_q = heads; d.heads_kv = heads; d.dim_head = dim_head; d.causal = 0; d.scale = scale;
  d.rel_qe = nullptr; d.ld_rel_qe = 0; d.rel_stride_h = 0; d.rel_left = d.rel_right = 0;
  d.lse = nullptr;
  return d;
}

// as tribe_internal_attention_fused, with the partial rotary of Q (interleaved pairs) applied inside the kernel: q in `qkv` is NOT
// rotated, k is.  Only valid while tribe_internal_attention_rotates_q(dim_head).
int tribe_internal_attention_fused_qrot(const uint16_t* qkv, int64_t B, int64_t T, int heads, int dim_head, float scale, uint16_t* out,
                                        const float* cos_tab, const float* sin_tab, int rot_dim, hipStream_t s) {
  TRIBE_REQUIRE(tribe_internal_attention_rotates_q(dim_head) && cos_tab && sin_tab && rot_dim > 0 && rot_dim % 16 == 0 && rot_dim <= dim_head,
                "tribe_internal_attention_fused_qrot: unsupported configuration");
  const tribe_attention_desc d = fused_desc(qkv, B, T, heads, dim_head, scale, out);
  return attention_dispatch(&d, cos_tab, sin_tab, rot_dim, (void*)s);
}

int tribe_internal_attention_fused(const uint16_t* qkv, int64_t B, int64_t T, int heads, int dim_head, float scale, uint16_t* out,
                                   hipStream_t s) {
  const tribe_attention_desc d = fused_desc(qkv, B, T, heads, dim_head, scale, out);
  return attention_dispatch(&d, nullptr, nullptr, 0, (void*)s);
}
