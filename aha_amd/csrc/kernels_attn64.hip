// Prefill attention, the one-wave-per-SIMD form (round 6; round-5 verdict, next-round item 1): a workgroup = 4 waves = one 256-row
// Q block, each wave 64 q rows (two 32-row q blocks) on v_mfma_f32_32x32x16_bf16 with the whole 512-register file of its SIMD.
// Same semantics, same page layout and the same rounding points as attn_prefill_kernel's f32 score chain (kernels_attn.hip, SMX 3;
// reference op sequence: eager_attention_forward, /root/reference/src/models/common/modules.rs:757-813 -- scores = q . k^T * scale,
// + causal mask, softmax, . v): the scores stay the f32 QK^T accumulators through scale, mask, maximum and exponential, P is rounded
// to bf16 once for the P.V MFMA, the row sum is the f32 sum of the un-rounded probabilities (text) or rides a ones row of the V^T pad
// through the matrix pipe (ViT, LSUM).  What differs is the work per flop:
//   * a 1-KB fragment read from LDS feeds TWO 32x32x16 MFMAs (both q blocks of the wave) = 64 Ki MACs, where the 16-row kernel's read
//     feeds one 16x16x32 MFMA = 8 Ki MACs: 1/8 of the LDS read traffic per flop;
//   * a staged K / V^T tile serves 256 q rows instead of 128 (or 64): half the LDS-DMA volume per flop;
//   * no partner wave on the SIMD: nothing to arbitrate the matrix pipe or the VALU issue with.
//
// Fragment scheme (wave64; q = lane & 31, hi = lane >> 5; crow(r, hi) = (r & 3) + 8 * (r >> 2) + 4 * hi):
//   S^T block (32 tokens x 32 q rows) = K . Q^T :  A = K   (row = token q,  k = dims ks*16 + 8*hi .. +8)
//                                                  B = Q^T (col = q row q,  k = dims ks*16 + 8*hi .. +8: 16 B straight from a q row)
//                                                  C[r]  = S[token crow(r, hi)][q row q]     -> softmax statistics are per lane column
//   O^T block (32 dims x 32 q rows)   = V^T . P^T: A = V^T (row = dim q,    k = 8 token slots)
//                                                  B = P^T (col = q row q,  k = the tokens of C[4s'..4s'+3] and C[8+4s'..8+4s'+3])
//                                                  C[r]  = O[q row q][dim crow(r, hi)]
// The fragment-major KV pages (common.h kpage_elem / vpage_elem, written for v_mfma_f32_16x16x32_bf16) feed this unchanged: lanes 0-15 /
// 16-31 read the same 256-byte window of two neighbouring 1-KB fragments (tokens or dims 0-15 / 16-31 of the 32-row block), lanes 32-63
// the next window -- every 16-lane group of a ds_read_b128 covers 256 contiguous bytes: conflict-free.  The V slot permutation (v_slot)
// is exactly the token order the 32x32 accumulator hands a lane: slot G'*8 + i of a 32-token run holds token (i >> 2) * 16 + G' * 4 +
// (i & 3), and with G' = 2 s' + hi that is crow(4 s' + (i & 3) + 8 (i >> 2), hi) -- so P goes from the S^T accumulators through
// v_cvt_pk_bf16_f32 straight into the next MFMA's B operand, no permlane, no LDS.
//
// Staging: K / V^T tiles by LDS-DMA (buffer_load_dwordx4 ... lds, one 1-KB piece per wave-instruction, buffer resources built from the
// scalar page pointer: out-of-range pieces -- the ViT's 10-KB V^T block read as 12 KB -- write zeros and fetch nothing) into a ring of
// three stages, two tiles ahead; one counted s_waitcnt vmcnt + one s_barrier per tile.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "attn_common.h"

namespace aha {

namespace {

extern __shared__ __attribute__((aligned(16))) char attn64_smem[];   // the block's dynamic LDS (the only __shared__ object of the unit)

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// v_permlane32_swap of a value with itself: one of the two results is the lane's own value, the other the value lane ^ 32 holds
__device__ __forceinline__ float xhi_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhi_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xhi_low(float v) {   // the value of the column's hi = 0 lane, in both lanes
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int ATTN64_ROWS = 256;   // q rows per workgroup

// PIPE: 0 = a tile's four parts in program order (QK^T, maximum, probabilities, P.V); 1 = software-pipelined inside the wave: QK^T of
// tile t+1 beside the exponentials of tile t, P.V of tile t beside the maximum of tile t+1 (two S^T register sets)
template <int DQK, int DV, bool LSUM, int PIPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_prefill64_kernel(AttnPrefillArgs a) {
  constexpr int KS = DQK / 32;          // 32-dim fragment columns of a K block
  constexpr int NKS = DQK / 16;         // k-steps of QK^T
  constexpr int NDB = (DV + 31) / 32;   // 32-dim output blocks
  constexpr int KBYTES = KV_PAGE_TOKENS * DQK * 2, VBYTES = DV * KV_PAGE_TOKENS * 2;
  constexpr int VSPAN = NDB * 4096;     // bytes of V^T the fragment reads cover (ViT: 12 KB of a 10-KB block; the rest arrives as zeros)
  constexpr int KPW = KBYTES / 4096, VPW = VSPAN / 4096;   // 1-KB DMA pieces per wave and tile: 4 + 4 (text), 3 + 3 (ViT)
  constexpr int NP = KPW + VPW;
  constexpr int STAGE = KBYTES + VSPAN;
  constexpr int NSTAGE = 3;
  static_assert(KBYTES % 4096 == 0 && NP < 16, "piece counts");
  char* const smem = attn64_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q32 = lane & 31, hi = lane >> 5;

  // ---- block -> (head, q block, segment): the XCD-aware order of attn_prefill_kernel with 256-row blocks ----
  int head, qblk, seg_rows = a.S, seg_off = a.kv_offset, seg_tot = a.kv_total, seg_row0 = 0;
  if (a.nqb > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = a.nh / a.kvh, hpx = a.nh >> 3;
    const int hq = slot % hpx, qi = slot / hpx;
    head = (xcd + 8 * (hq / g)) * g + hq % g;
    qblk = a.causal ? a.nqb - 1 - qi : qi;
    if (a.S2 > 0) {
      const int nqb2 = (a.S2 + ATTN64_ROWS - 1) / ATTN64_ROWS;
      if (qi < nqb2) {
        qblk = nqb2 - 1 - qi;
        seg_rows = a.S2, seg_off = a.kv_offset2, seg_tot = a.kv_total2, seg_row0 = a.S;
      }
    }
  } else {
    head = blockIdx.y;
    qblk = blockIdx.x;
  }
  const int kvhd = head / (a.nh / a.kvh);
  const int qb0 = qblk * ATTN64_ROWS;   // first q row of the block
  const int q0 = qb0 + wave * 64;       // first q row of the wave

  // ---- Q^T operand fragments, straight from the q rows ----
  bf16x8_t qf[2][NKS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = seg_row0 + min(q0 + qb * 32 + q32, seg_rows - 1);
    const bf16_t* qp = (const bf16_t*)a.q + (int64_t)qrow * (a.q_ld ? a.q_ld : (int64_t)a.nh * DQK) + (int64_t)head * DQK;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[qb][ks] = as_frag(ld16(qp + ks * 16 + hi * 8));
  }
  const int blk_last_q = min(qb0 + ATTN64_ROWS - 1, seg_rows - 1);
  const int last_tok = a.causal ? min(seg_off + blk_last_q, seg_tot - 1) : seg_tot - 1;
  const int ntiles = last_tok / KV_PAGE_TOKENS + 1;
  // tiles this WAVE needs (causal: up to its own last row); past them it only stages and meets the barriers
  const int wv_last_q = min(q0 + 63, seg_rows - 1);
  const int wtiles = (q0 < seg_rows) ? ((a.causal ? min(seg_off + wv_last_q, seg_tot - 1) : seg_tot - 1) / KV_PAGE_TOKENS + 1) : 0;

  // ---- staging ----
  typedef const __attribute__((address_space(4))) uint64_t* cptr64_t;
  const cptr64_t ptab = (cptr64_t)(uintptr_t)a.kv.page_ptrs;
  int voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) voff[i] = (wave + 4 * i) * 1024 + lane * 16;
  const uint64_t koff = a.kv.layer_off + (uint64_t)kvhd * KBYTES;
  const uint64_t voffb = a.kv.layer_off + (uint64_t)a.kvh * KBYTES + (uint64_t)kvhd * VBYTES;
  auto dma_tile = [&](uint64_t page, int stage_off) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(page + koff), 0, KBYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(page + voffb), 0, VBYTES, 0x00020000);
    char* dst = smem + stage_off;
#pragma unroll
    for (int i = 0; i < KPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)(dst + (wave + 4 * i) * 1024), 16, voff[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < VPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dst + KBYTES + (wave + 4 * i) * 1024), 16, voff[i], 0, 0, 0);
  };
#define ATTN64_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))
#define ATTN64_BAR()                             \
  do {                                           \
    __builtin_amdgcn_sched_barrier(0);           \
    __builtin_amdgcn_s_barrier();                \
    __builtin_amdgcn_sched_barrier(0);           \
  } while (0)

  // ---- per-lane fragment read bases (everything else of a fragment address is an instruction offset) ----
  const int lk = (((q32 >> 4) * KS * 64) + hi * 16 + (q32 & 15)) * 16;
  const int lv = KBYTES + (((q32 >> 4) * 2 * 64) + hi * 16 + (q32 & 15)) * 16;
  auto kfrag = [&](int base, int tb, int ks) __attribute__((always_inline)) {
    return as_frag(*reinterpret_cast<const u32x4_t*>(smem + base + ((tb * 2 * KS + (ks >> 1)) * 1024 + (ks & 1) * 512)));
  };
  auto vfrag = [&](int base, int db, int kstep) __attribute__((always_inline)) {   // kstep = kk * 2 + s'
    return as_frag(*reinterpret_cast<const u32x4_t*>(smem + base + ((db * 4 + (kstep >> 1)) * 1024 + (kstep & 1) * 512)));
  };

  // ---- online-softmax state ----
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f32x16_t o[NDB][2];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][qb][r] = 0.f;
  const float c2 = a.scale * 1.4426950408889634f;

  // QK^T of one tile: S^T[tb][qb]
  auto qk_part = [&](int kbase, f32x16_t (&s)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[tb][qb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        const bf16x8_t kf = kfrag(kbase, tb, ks);
        s[tb][0] = mfma32(kf, qf[0][ks], s[tb][0]);
        s[tb][1] = mfma32(kf, qf[1][ks], s[tb][1]);
      }
  };
  // mask, tile maximum, new running maximum, rescale factor (the score side of softmax_scores<3>, attn_common.h)
  auto max_part = [&](int t0, f32x16_t (&s)[2][2], float (&alpha)[2], float (&m2)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qpos = seg_off + q0 + qb * 32 + q32;   // cache position of this lane's q row
      const int lim = a.causal ? min(qpos, seg_tot - 1) : seg_tot - 1;
      const int lim_min = a.causal ? min(seg_off + q0 + qb * 32, seg_tot - 1) : seg_tot - 1;
      if (t0 + KV_PAGE_TOKENS - 1 > lim_min) {   // wave-uniform: only diagonal / last tiles carry the per-element predicate
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (t0 + tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi > lim) s[tb][qb][r] = -INFINITY;
      }
      auto sv = [&](int i) { return s[i >> 4][qb][i & 15]; };
      float tmax = max3(sv(0), sv(1), sv(2));
#pragma unroll
      for (int i = 3; i < 31; i += 2) tmax = max3(tmax, sv(i), sv(i + 1));
      tmax = max3(tmax, sv(31), sv(31));
      tmax = xhi_max(tmax);
      const float m_new = fmaxf(m[qb], tmax);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      alpha[qb] = __builtin_amdgcn_exp2f((m[qb] - m_use) * c2);   // m = -inf -> 0
      m2[qb] = m_use * c2;
      m[qb] = m_new;
    }
  };
  // p = exp2(s c2 - m c2), the row sum, the bf16 P^T operand fragments
  auto prob_part = [&](f32x16_t (&s)[2][2], const float (&alpha)[2], const float (&m2)[2], bf16x8_t (&pf)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tb][qb][r], c2, -m2[qb]));
        if (!LSUM) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) sum0 += p[r], sum1 += p[r + 1];
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const u32x4_t u = {pack_bf(p[4 * sp + 0], p[4 * sp + 1]), pack_bf(p[4 * sp + 2], p[4 * sp + 3]),
                             pack_bf(p[8 + 4 * sp + 0], p[8 + 4 * sp + 1]), pack_bf(p[8 + 4 * sp + 2], p[8 + 4 * sp + 3])};
          pf[qb][tb * 2 + sp] = as_frag(u);
        }
      }
      if (!LSUM) l[qb] = l[qb] * alpha[qb] + (sum0 + sum1);
    }
  };
  // once the running maximum has settled alpha is exactly 1 in every lane: skip the multiplies (x * 1 == x)
  auto rescale_part = [&](const float (&alpha)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
      if (__builtin_amdgcn_ballot_w64(alpha[qb] != 1.f) != 0) {
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][qb][r] *= alpha[qb];
      }
  };
  auto pv_part = [&](int vbase, const bf16x8_t (&pf)[2][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int kstep = 0; kstep < 4; ++kstep)
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const bf16x8_t vf = vfrag(vbase, db, kstep);
        o[db][0] = mfma32(vf, pf[0][kstep], o[db][0]);
        o[db][1] = mfma32(vf, pf[1][kstep], o[db][1]);
      }
  };

  // ---- prologue: tiles 0 and 1 requested, tile 0 landed ----
  dma_tile(ptab[0], 0);
  if (ntiles > 1) dma_tile(ptab[1], STAGE);
  uint64_t pg_next = ptab[__builtin_amdgcn_readfirstlane(min(2, ntiles - 1))];
  if (ntiles > 1) ATTN64_WAIT_VM(NP);
  else ATTN64_WAIT_VM(0);
  ATTN64_BAR();

  if (PIPE == 0) {
    int stage_off = 0, pre_off = 2 * STAGE;   // LDS offsets of tile t's stage and of the stage tile t + 2 goes to
    for (int tile = 0; tile < ntiles; ++tile) {
      if (tile + 2 < ntiles) dma_tile(pg_next, pre_off);
      pg_next = ptab[__builtin_amdgcn_readfirstlane(min(tile + 3, ntiles - 1))];
      if (tile < wtiles) {
        f32x16_t s[2][2];
        float alpha[2], m2[2];
        bf16x8_t pf[2][4];
        qk_part(lk + stage_off, s);
        max_part(tile * KV_PAGE_TOKENS, s, alpha, m2);
        rescale_part(alpha);
        prob_part(s, alpha, m2, pf);
        pv_part(lv + stage_off, pf);
      }
      if (tile + 2 < ntiles) ATTN64_WAIT_VM(NP);   // tile t + 1 has landed, t + 2 may still fly
      else ATTN64_WAIT_VM(0);
      ATTN64_BAR();
      pre_off = stage_off;
      stage_off = (stage_off == 2 * STAGE) ? 0 : stage_off + STAGE;
    }
  } else {
    // Software pipeline inside the wave.  Iteration `tile` runs  QK^T(tile + 1) beside the exponentials of `tile`  and then
    // P.V(tile) beside the maximum of tile + 1; the rescale by alpha(tile + 1) follows P.V(tile).  Two S^T register sets.
    f32x16_t sA[2][2], sB[2][2];
    float alphaA[2], m2A[2], alphaB[2], m2B[2];
    bf16x8_t pf[2][4];
    int stage_off = 0, nxt_off = STAGE, pre_off = 2 * STAGE;
    if (wtiles > 0) {
      qk_part(lk, sA);
      max_part(0, sA, alphaA, m2A);   // alpha = 0 on nothing accumulated yet: no rescale needed
    }
    auto iter = [&](int tile, f32x16_t (&sc)[2][2], float (&ac)[2], float (&mc)[2], f32x16_t (&sn)[2][2], float (&an)[2], float (&mn)[2])
        __attribute__((always_inline)) {
      if (tile + 2 < ntiles) dma_tile(pg_next, pre_off);
      pg_next = ptab[__builtin_amdgcn_readfirstlane(min(tile + 3, ntiles - 1))];
      if (tile < wtiles) {
        const bool more = tile + 1 < wtiles;   // (tile + 1 landed: the barrier at the end of the previous iteration)
        if (more) qk_part(lk + nxt_off, sn);
        prob_part(sc, ac, mc, pf);
        pv_part(lv + stage_off, pf);
        if (more) {
          max_part((tile + 1) * KV_PAGE_TOKENS, sn, an, mn);
          rescale_part(an);
        }
      }
      // the next iteration reads K(tile + 2) and V^T(tile + 1): everything but the newest VPW pieces (V^T of tile + 2)
      if (tile + 2 < ntiles) ATTN64_WAIT_VM(VPW);
      else ATTN64_WAIT_VM(0);
      ATTN64_BAR();
      pre_off = stage_off;
      stage_off = nxt_off;
      nxt_off = (nxt_off == 2 * STAGE) ? 0 : nxt_off + STAGE;
    };
    // iteration 0 reads tile 1's K: landed once only tile 1's V^T pieces (the newest VPW requests) may still fly
    if (ntiles > 1) {
      ATTN64_WAIT_VM(VPW);
      ATTN64_BAR();
    }
    int tile = 0;
    for (; tile + 1 < ntiles; tile += 2) {
      iter(tile, sA, alphaA, m2A, sB, alphaB, m2B);
      iter(tile + 1, sB, alphaB, m2B, sA, alphaA, m2A);
    }
    if (tile < ntiles) iter(tile, sA, alphaA, m2A, sB, alphaB, m2B);
  }

  // ---- epilogue: 1 / row sum, bf16, whole rows through this wave's slice of the (now free) staging LDS, 16 B per lane ----
  constexpr int EPITCH = DV * 2 + 16;   // bytes per LDS row
  char* wb = smem + wave * (64 * EPITCH);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    // LSUM: output row 72 = block db 2, crow(4, 0) = 8 -> register 4 of the hi = 0 lane of the column
    const float lt = LSUM ? xhi_low(o[NDB - 1][qb][4]) : xhi_sum(l[qb]);
    const float inv = 1.0f / lt;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dim = db * 32 + 8 * j + 4 * hi;
        if (dim < DV) {
          uint2 w;
          w.x = pack_bf(o[db][qb][4 * j + 0] * inv, o[db][qb][4 * j + 1] * inv);
          w.y = pack_bf(o[db][qb][4 * j + 2] * inv, o[db][qb][4 * j + 3] * inv);
          *reinterpret_cast<uint2*>(wb + (qb * 32 + q32) * EPITCH + dim * 2) = w;
        }
      }
  }
  // (same wave wrote and reads: no barrier, the LDS queue is in order)
  const int chunk = lane & 15;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4), qr = q0 + row;
    if (qr < seg_rows && chunk * 8 < a.d) {   // a.d = real head dim of the output rows: a multiple of 8 on this path (launcher)
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wb + row * EPITCH + chunk * 16);
      *reinterpret_cast<u32x4_t*>((bf16_t*)a.o + ((int64_t)(seg_row0 + qr) * a.nh + head) * a.d + chunk * 8) = v;
    }
  }
#undef ATTN64_WAIT_VM
#undef ATTN64_BAR
}

template <typename K>
void set_max_lds(K kernel, size_t lds) {
  // > 64 KiB of dynamic LDS needs the attribute, per device; set first, publish afterwards, under a mutex (a second thread must not
  // see "done" before the attribute is in place)
  static std::mutex mu;
  static uint64_t done[4] = {0, 0, 0, 0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  if (dev >= 0 && dev < 256 && (done[dev >> 6] >> (dev & 63) & 1)) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (dev >= 0 && dev < 256) done[dev >> 6] |= 1ull << (dev & 63);
}

}  // namespace

// The 64-rows-per-wave form of launch_attn_prefill (kernels_attn.hip decides when): f32 score chain only, head_dim 128 or the ViT's 72
// (96 / 80 padded pages), output rows in multiples of 8 dims at 16-byte aligned addresses.  Returns false when the shape is not its own.
bool launch_attn_prefill64(const AttnPrefillArgs& a_in, hipStream_t st, int pipe) {
  AttnPrefillArgs a = a_in;
  if (!(a.d == 128 || a.d == 72) || a.d % 8 != 0 || ((uintptr_t)a.o & 15) != 0 || !(a.scale > 0.f)) return false;
  if (a.S2 > 0 && !(a.causal && a.kvh % 8 == 0 && a.nh % a.kvh == 0)) return false;
  const bool xcd_order = a.kvh % 8 == 0 && a.nh % a.kvh == 0;
  const int nqb = (a.S + ATTN64_ROWS - 1) / ATTN64_ROWS + (a.S2 + ATTN64_ROWS - 1) / ATTN64_ROWS;
  a.nqb = xcd_order ? nqb : 0;
  a.epi_rows = 1;
  const dim3 grid = a.nqb ? dim3(nqb * a.nh) : dim3(nqb, a.nh), block(256);
#define ATTN64_LAUNCH(DQK_, DV_, LSUM_)                                                                     \
  do {                                                                                                      \
    constexpr size_t lds = 3 * (KV_PAGE_TOKENS * DQK_ * 2 + ((DV_ + 31) / 32) * 4096);                      \
    if (pipe) {                                                                                             \
      set_max_lds(attn_prefill64_kernel<DQK_, DV_, LSUM_, 1>, lds);                                         \
      hipLaunchKernelGGL((attn_prefill64_kernel<DQK_, DV_, LSUM_, 1>), grid, block, lds, st, a);            \
    } else {                                                                                                \
      set_max_lds(attn_prefill64_kernel<DQK_, DV_, LSUM_, 0>, lds);                                         \
      hipLaunchKernelGGL((attn_prefill64_kernel<DQK_, DV_, LSUM_, 0>), grid, block, lds, st, a);            \
    }                                                                                                       \
  } while (0)
  if (a.d == 128) ATTN64_LAUNCH(128, 128, false);
  else if (a.v_ones_row) ATTN64_LAUNCH(96, 80, true);
  else ATTN64_LAUNCH(96, 80, false);
#undef ATTN64_LAUNCH
  return true;
}

}  // namespace aha
