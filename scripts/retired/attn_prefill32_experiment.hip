// Prefill attention, the 32-row form (round 5): the same contract as attn_prefill_kernel (kernels_attn.hip: AttnPrefillArgs, the
// fragment-major KV pages of common.h, flash-style online softmax, causal predicate in-kernel, no repeat_kv, no mask tensor) with
// TWO changes of what work is done per score:
//   * every wave owns 32 q rows and runs v_mfma_f32_32x32x16_bf16: a 1-KB fragment read from LDS now feeds 32 x 32 x 16 MACs
//     instead of 16 x 16 x 32 -- half the LDS read traffic per flop (the 16-row kernel reads 32 KB of K / V^T per wave and 64-token
//     tile: with 16 waves per CU its LDS reads alone are as long as its MFMAs, and the parts add up, profiles/r04_attn_prefill.md);
//   * the f32 score chain (attn_common.h SMX 3): the scores stay the f32 QK^T accumulators through mask, maximum and exponential,
//     P is rounded to bf16 once for the P.V MFMA -- the reference's eager path rounds the scores twice (modules.rs:782-783), which
//     was ~40 of the ~92 vector instructions per tile and is not needed for "fp tensors within stated tol" (DESIGN.md section 2).
// Reference semantics: eager_attention_forward, /root/reference/src/models/common/modules.rs:757-813; repeat_kv
// /root/reference/src/utils/tensor_utils.rs:108-124; causal mask tensor_utils.rs:78-106.
//
// Fragment scheme (wave64; h = lane >> 5, c = lane & 31, c16 = lane & 15, hi = (lane >> 4) & 1):
//   S^T tile (32 tokens x 32 q rows) = K . Q^T:
//       A = K   (row = token jt*32 + c, k = dims ks*16 + h*8 .. +8): token sub-tile jt*2 + hi, piece (ks&1)*2 + h of the page's
//                fragment (sub, ks >> 1) -- lanes 0-15 / 16-31 read the same 256-byte window of two fragments 1 KB * KSF apart,
//                lanes 32-63 the next window: every 16-lane ds_read_b128 group covers the 64 banks once (conflict-free)
//       B = Q^T (col = q row c, k = dims ks*16 + h*8 .. +8: 16 B straight from the q row)
//       C[r]   = S[token (r/4)*8 + h*4 + r%4][q c]
//   O^T tile (32 dims x 32 q rows) = V^T . P^T, k = 16 token slots per MFMA (kk = 32-token half, j = 0 / 1):
//       A = V^T (row = dim dt*32 + c, k = 8 token slots): piece 2j + h of fragment (ds = dt*2 + hi, kk); by v_slot (common.h) that
//                piece holds tokens kk*32 + {(2j+h)*4 .. +3} and kk*32 + 16 + {(2j+h)*4 .. +3}
//       B = P^T : lane (h, c) holds exactly those tokens of q row c in S^T registers 4j .. 4j+3 and 8+4j .. 8+4j+3 of tile kk
//   so, as in the 16-row kernel, no operand needs a transpose or a cross-lane move; the only cross-lane step of a tile is ONE
//   v_permlane32_swap for the row maximum (the row sum stays per lane until the epilogue).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <utility>

#include "attn_common.h"

namespace aha {

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// DQK / DV: padded head dims of the page layout (K fragments per token sub-tile = DQK / 32, V fragments per page = 2 * DV / 16);
// KST: 16-dim k steps of QK^T that hold non-zero dims (text 8, ViT head_dim 72: 5, audio 4); NWV waves x 32 q rows per block.
// ABL (debug, results wrong by construction): 2 = no softmax arithmetic, 3 = no staging of the next tile, 4 = no MFMAs.
// STG: 0 = the next tile travels global -> LDS by LDS-DMA (global_load_lds_dwordx4); 1 = through registers (global_load_dwordx4 at the top
// of the tile, ds_write_b128 in front of the barrier that ends it)
// PF: prefetch distance in tiles (LDS stages = PF + 1): 2 = the tile after next is requested at the top of a tile, the end-of-tile wait
// leaves its pieces in flight (counted vmcnt)
template <int DQK, int DV, int KST, int NWV, int ABL = 0, int STG = 0, int PF = 1>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(DQK >= 128 ? 2 : 3, DQK >= 128 ? 2 : 3))) void attn_prefill32_kernel(AttnPrefillArgs a) {
  constexpr int KSF = DQK / 32, DSF = DV / 16, DT = (DV + 31) / 32;
  constexpr int RING = 6;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x [K tile | V^T tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31, c16 = lane & 15, hi = (lane >> 4) & 1;
  constexpr int BR = 32 * NWV;  // q rows per block
  int head, qblk;
  if (a.nqb > 0) {  // XCD-aware order (kernels_attn.hip): XCD x works on kv heads x, x+8, ..; causal launches hand out long blocks first
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = a.nh / a.kvh, hpx = a.nh >> 3;
    const int hq = slot % hpx, qi = slot / hpx;
    head = (xcd + 8 * (hq / g)) * g + hq % g;
    qblk = a.causal ? a.nqb - 1 - qi : qi;
  } else {
    head = blockIdx.y;
    qblk = blockIdx.x;
  }
  const int kvhd = head / (a.nh / a.kvh);
  const int qb = qblk * BR, q0 = qb + wave * 32;
  bf16x8_t qf[KST];
  {
    const int qrow = min(q0 + c, a.S - 1);
    const bf16_t* qp = (const bf16_t*)a.q + (int64_t)qrow * (a.q_ld ? a.q_ld : (int64_t)a.nh * DQK) + (int64_t)head * DQK;
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) qf[ks] = as_frag(ld16(qp + ks * 16 + h * 8));
  }
  const int blk_last_q = min(qb + BR - 1, a.S - 1);
  const int last_tok = a.causal ? min(a.kv_offset + blk_last_q, a.kv_total - 1) : a.kv_total - 1;
  const int ntiles = last_tok / KV_PAGE_TOKENS + 1;

  float m = -INFINITY, l = 0.f;   // running maximum in RAW score units (shared by lanes c and c + 32), this lane's share of the row sum
  f32x16_t o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

  // staging: the LDS image of a tile is a byte copy of the kv head's K and V blocks of the page (LDS-DMA, 1 KB per wave instruction)
  constexpr int KP = KV_PAGE_TOKENS * DQK / 8, VP = DV * KV_PAGE_TOKENS / 8;  // 16-byte pieces
  constexpr int KB = KP / 64, VB = VP / 64;                                    // 1-KB fragment blocks
  constexpr int STAGE_BYTES = (KP + VP) * 16;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  auto gload = [&](uint64_t page, int stage) __attribute__((always_inline)) {
    const uint64_t kb = page + a.kv.layer_off + (uint64_t)kvhd * KV_PAGE_TOKENS * (DQK * 2);
    const uint64_t vb = page + a.kv.layer_off + (uint64_t)a.kvh * KV_PAGE_TOKENS * (DQK * 2) + (uint64_t)kvhd * DV * (KV_PAGE_TOKENS * 2);
    char* dst = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < (KB + NWV - 1) / NWV; ++i) {
      const int blk = wave + i * NWV;   // wave-uniform
      if (KB % NWV == 0 || blk < KB)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(kb + blk * 1024 + lane * 16), (lds_ptr_t)(dst + blk * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < (VB + NWV - 1) / NWV; ++i) {
      const int blk = wave + i * NWV;
      if (VB % NWV == 0 || blk < VB)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(vb + blk * 1024 + lane * 16), (lds_ptr_t)(dst + KP * 16 + blk * 1024), 16, 0, 0);
    }
  };
  typedef const __attribute__((address_space(4))) uint64_t* cptr64_t;
  const cptr64_t ptab = (cptr64_t)(uintptr_t)a.kv.page_ptrs;
  gload(ptab[0], 0);
  if (PF == 2) gload(ptab[__builtin_amdgcn_readfirstlane(min(1, ntiles - 1))], 1);
  uint64_t pg_next = ptab[__builtin_amdgcn_readfirstlane(min(PF, ntiles - 1))];
  if (PF == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((KB + VB + NWV - 1) / NWV));   // vmcnt(pieces of tile 1): tile 0 and the q fragments have landed
  else __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the q fragments and the first tile (kernels_attn.hip: why once, here)
  __syncthreads();

  // lane parts of the fragment addresses (bytes inside a K / V^T tile image)
  const int k_lane = hi * (KSF * 1024) + (h * 16 + c16) * 16;
  const int v_lane = hi * 2048 + (h * 16 + c16) * 16;
  const int v_lane_last = (h * 16 + c16) * 16;   // last dim tile of an odd fragment count (ViT: 5): both lane halves read fragment ds = DSF - 1
  const float c2 = a.scale * 1.4426950408889634f;
  const int qpos = a.kv_offset + q0 + c;
  const int lim = a.causal ? min(qpos, a.kv_total - 1) : a.kv_total - 1;
  const int lim_min = a.causal ? min(a.kv_offset + q0, a.kv_total - 1) : a.kv_total - 1;   // wave-uniform

  for (int tile = 0; tile < ntiles; ++tile) {
    // unconditional prefetch of the next tile into the other stage (the last iteration re-requests its own tile; nobody reads it)
    constexpr int NPK = (KB + NWV - 1) / NWV, NPV = (VB + NWV - 1) / NWV;
    u32x4_t stg[NPK + NPV];
    if (STG == 0) {
      if (ABL != 3) gload(pg_next, PF == 2 ? (tile + 2) % 3 : (tile + 1) & 1);
    } else {
      const uint64_t kb = pg_next + a.kv.layer_off + (uint64_t)kvhd * KV_PAGE_TOKENS * (DQK * 2);
      const uint64_t vb = pg_next + a.kv.layer_off + (uint64_t)a.kvh * KV_PAGE_TOKENS * (DQK * 2) + (uint64_t)kvhd * DV * (KV_PAGE_TOKENS * 2);
#pragma unroll
      for (int i = 0; i < NPK; ++i) stg[i] = ld16_global(kb + (uint64_t)(min(wave + i * NWV, KB - 1) * 1024 + lane * 16));
#pragma unroll
      for (int i = 0; i < NPV; ++i) stg[NPK + i] = ld16_global(vb + (uint64_t)(min(wave + i * NWV, VB - 1) * 1024 + lane * 16));
    }
    pg_next = ptab[__builtin_amdgcn_readfirstlane(min(tile + 1 + PF, ntiles - 1))];
    const int t0 = tile * KV_PAGE_TOKENS;
    const bool act = !a.causal || t0 <= a.kv_offset + q0 + 31;   // wave-uniform: the wave's rows see something of the tile
    if (act) {
      const int stage_cur = PF == 2 ? tile % 3 : (tile & 1);
      const char* ks_base = smem + stage_cur * STAGE_BYTES + k_lane;
      const char* vs_base = smem + stage_cur * STAGE_BYTES + KP * 16;
      // ---- S^T = K . Q^T: 2 token halves x KST k steps; consecutive MFMAs alternate between the two accumulators ----------------
      f32x16_t st[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[jt][r] = 0.f;
      {
        constexpr int NF = 2 * KST;
        auto kread = [&](int f) {  // f = ks * 2 + jt
          const int ks = f >> 1, jt = f & 1;
          return *reinterpret_cast<const u32x4_t*>(ks_base + (jt * 2 * KSF + (ks >> 1)) * 1024 + (ks & 1) * 512);
        };
        u32x4_t ring[RING];
#pragma unroll
        for (int f = 0; f < RING && f < NF; ++f) ring[f] = kread(f);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bf16x8_t kf = as_frag(ring[f % RING]);
          if (ABL != 4) st[f & 1] = mfma32(kf, qf[f >> 1], st[f & 1]);
          else asm volatile("" :: "v"(kf));
          if (f + RING < NF) ring[f % RING] = kread(f + RING);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the first V^T fragments travel while the softmax runs
      auto vread = [&](int f) {  // f = (kk * 2 + j) * DT + dt
        const int dt = f % DT, kj = f / DT, kk = kj >> 1, j = kj & 1;
        const bool last_odd = (DSF & 1) && dt == DT - 1;
        return *reinterpret_cast<const u32x4_t*>(vs_base + (last_odd ? v_lane_last : v_lane) + (dt * 4 + kk) * 1024 + j * 512);
      };
      constexpr int NFV = 4 * DT;
      u32x4_t vring[RING];
#pragma unroll
      for (int f = 0; f < RING; ++f) vring[f] = vread(f);
      __builtin_amdgcn_sched_barrier(0);
      // ---- softmax, f32 chain -----------------------------------------------------------------------------------------------------
      bf16x8_t pf[2][2];
      if (ABL == 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            u32x4_t u = {__float_as_uint(st[kk][4 * j]), __float_as_uint(st[kk][4 * j + 1]), __float_as_uint(st[kk][8 + 4 * j]), __float_as_uint(st[kk][9 + 4 * j])};
            pf[kk][j] = as_frag(u);
          }
        l += 1.f;
      } else {
        if (t0 + KV_PAGE_TOKENS - 1 > lim_min) {   // diagonal / last tile: the per-element predicate (interior tiles skip it)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (t0 + jt * 32 + (r >> 2) * 8 + h * 4 + (r & 3) > lim) st[jt][r] = -INFINITY;
        }
        // v_max3_f32 by hand: fmaxf on an MFMA output makes clang canonicalise it first (v_max_f32 x, x), 3 instructions per pair of
        // scores instead of 1; the accumulators hold no signalling NaNs
        auto sv = [&](int i) { return st[i >> 4][i & 15]; };   // the tile's 32 scores of this lane
        float tmax = max3(sv(0), sv(1), sv(2));
#pragma unroll
        for (int i = 3; i < 31; i += 2) tmax = max3(tmax, sv(i), sv(i + 1));
        tmax = max3(tmax, sv(31), sv(31));
        {
          const unsigned u = __float_as_uint(tmax);
          const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
          tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const float m_new = fmaxf(m, tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;   // fully masked so far: keep everything at zero
        const float alpha = __builtin_amdgcn_exp2f((m - m_use) * c2);   // m = -inf -> 0
        const float m2 = m_use * c2;
        m = m_new;
        const f32x2_t k2 = {c2, c2}, nm2 = {-m2, -m2};
        f32x2_t psum2 = {0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          uint32_t pk[8];
#pragma unroll
          for (int r2 = 0; r2 < 8; ++r2) {
            const f32x2_t s2 = {st[kk][2 * r2], st[kk][2 * r2 + 1]};
            const f32x2_t e2 = __builtin_elementwise_fma(s2, k2, nm2);
            const f32x2_t p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
            psum2 += p2;
            pk[r2] = pack_bf(p2[0], p2[1]);
          }
          // registers 4j .. 4j+3 and 8+4j .. 8+4j+3 = packed pairs 2j, 2j+1 and 4+2j, 5+2j
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            u32x4_t u = {pk[2 * j], pk[2 * j + 1], pk[4 + 2 * j], pk[5 + 2 * j]};
            pf[kk][j] = as_frag(u);
          }
        }
        l = l * alpha + (psum2[0] + psum2[1]);
        // once the running maximum has settled alpha is exactly 1 in every lane: skip the rescale (x * 1 == x)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- O^T += V^T . P^T: (kk, j)-major, DT independent accumulators in a row ---------------------------------------------------
#pragma unroll
      for (int f = 0; f < NFV; ++f) {
        const bf16x8_t vf = as_frag(vring[f % RING]);
        const int kj = f / DT;
        if (ABL != 4) o[f % DT] = mfma32(vf, pf[kj >> 1][kj & 1], o[f % DT]);
        else asm volatile("" :: "v"(vf), "v"(pf[kj >> 1][kj & 1]));
        if (f + RING < NFV) vring[f % RING] = vread(f + RING);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (STG == 1) {
      char* dst = smem + ((tile + 1) & 1) * STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < NPK; ++i) {
        const int blk = wave + i * NWV;
        if (KB % NWV == 0 || blk < KB) *reinterpret_cast<u32x4_t*>(dst + blk * 1024 + lane * 16) = stg[i];
      }
#pragma unroll
      for (int i = 0; i < NPV; ++i) {
        const int blk = wave + i * NWV;
        if (VB % NWV == 0 || blk < VB) *reinterpret_cast<u32x4_t*>(dst + KP * 16 + blk * 1024 + lane * 16) = stg[NPK + i];
      }
    } else if (PF == 2) {
      __builtin_amdgcn_s_waitcnt(0x0F70 | ((KB + VB + NWV - 1) / NWV));   // the next tile has landed, the one after it may stay in flight
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of the next tile has landed
    }
    __syncthreads();
  }

  // Epilogue: the wave's 32 x DV tile through its own slice of the (now free) staging LDS, stored as whole rows, 16 bytes per lane
  // (kernels_attn.hip: why row order).  A lane holds 4 consecutive dims of q row c per 4 accumulator registers.
  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = 1.0f / lt;
  constexpr int EPITCH = DT * 64 + 16;   // bytes per LDS row
  if (a.epi_rows) {
    char* wb = smem + wave * (32 * EPITCH);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        uint2 w;
        w.x = pack_bf(o[dt][r4 * 4 + 0] * inv, o[dt][r4 * 4 + 1] * inv);
        w.y = pack_bf(o[dt][r4 * 4 + 2] * inv, o[dt][r4 * 4 + 3] * inv);
        *reinterpret_cast<uint2*>(wb + c * EPITCH + (dt * 32 + r4 * 8 + h * 4) * 2) = w;
      }
    // (same wave wrote and reads: no barrier, the LDS queue is in order)
    const int chunk = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4), qr = q0 + row;
      if (qr < a.S && chunk * 8 < a.d) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wb + row * EPITCH + chunk * 16);
        *reinterpret_cast<u32x4_t*>((bf16_t*)a.o + ((int64_t)qr * a.nh + head) * a.d + chunk * 8) = v;
      }
    }
    return;
  }
  const int qr = q0 + c;
  if (qr < a.S) {
    bf16_t* op = (bf16_t*)a.o + ((int64_t)qr * a.nh + head) * a.d;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dim = dt * 32 + r4 * 8 + h * 4;
        if (dim < a.d) {  // head dims come in multiples of 4
          uint2 w;
          w.x = pack_bf(o[dt][r4 * 4 + 0] * inv, o[dt][r4 * 4 + 1] * inv);
          w.y = pack_bf(o[dt][r4 * 4 + 2] * inv, o[dt][r4 * 4 + 3] * inv);
          *reinterpret_cast<uint2*>(op + dim) = w;
        }
      }
  }
}


// ---- the software-pipelined form (AHA_ATTN_SMX=5) ---------------------------------------------------------------------------------
// PMC of the kernel above at S = 8192 (profiles/r05_attn_prefill.md): matrix pipe 49 % busy, the waves spend 50 % of their cycles unable to
// issue (SQ_WAIT_INST_ANY) and 19 % in s_waitcnt / s_barrier: with two waves per SIMD each running QK^T -> softmax -> P.V one after the
// other, whole-phase overlap ACROSS waves does not happen by itself.  Here every wave overlaps the two pipes inside its own instruction
// stream: while the matrix pipe computes S(t+1) = K(t+1) . Q^T the vector ALU turns S(t) into P(t) (one packed fma, two v_exp, one packed
// add and one v_cvt_pk per MFMA -- ~28 issue cycles in a 32-cycle MFMA slot), and while it computes O += V(t)^T . P(t)^T the vector ALU
// takes the row maximum of S(t+1).  K runs one tile ahead of V (K(t+1) and V(t) are read in iteration t; both double-buffered).
// Fragment reads are inline-asm ds_read_b128 with COUNTED lgkmcnt waits: hipcc batches the reads of a ring and waits lgkmcnt(0), a full
// LDS round trip in which nothing of the wave issues (cdna_hip_programming.md section 5.7 form (ii)).
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_read16_asm(u32x4_t& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait_asm(u32x4_t& d) {   // at most N LDS reads still outstanding; `d` is usable afterwards
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N));
}
// Order pins for the software pipeline.  hipcc places plain arithmetic wherever its DAG scheduler likes inside a basic block (the first
// build of this kernel came out with all 16 MFMAs of a phase first and the whole softmax behind them); volatile asm statements keep
// their order among themselves, so an EMPTY volatile asm that takes a value "+v" pins its producer before, and its consumers after,
// that point of the volatile sequence (the fragment reads and waits) -- no instruction is emitted.
template <typename T>
__device__ __forceinline__ void pin(T& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ float max3v(float a, float b, float c) {   // v_max3_f32 at its place in the volatile sequence
  float r;
  asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max3_after_mfma(float a, float b, float c) {   // first reader of fresh MFMA results inside an asm string: its own wait states
  float r;
  asm volatile("s_nop 15\n\tv_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int DQK, int DV, int KST, int NWV>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_prefill32p_kernel(AttnPrefillArgs a) {
  constexpr int KSF = DQK / 32, DSF = DV / 16, DT = (DV + 31) / 32;
  constexpr int RING = 4;
  constexpr int NF = 2 * KST, NFV = 4 * DT;
  static_assert(NF >= RING && NFV >= RING, "ring deeper than a phase");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K[2] | V^T[2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31, c16 = lane & 15, hi = (lane >> 4) & 1;
  constexpr int BR = 32 * NWV;
  int head, qblk;
  if (a.nqb > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = a.nh / a.kvh, hpx = a.nh >> 3;
    const int hq = slot % hpx, qi = slot / hpx;
    head = (xcd + 8 * (hq / g)) * g + hq % g;
    qblk = a.causal ? a.nqb - 1 - qi : qi;
  } else {
    head = blockIdx.y;
    qblk = blockIdx.x;
  }
  const int kvhd = head / (a.nh / a.kvh);
  const int qb = qblk * BR, q0 = qb + wave * 32;
  bf16x8_t qf[KST];
  {
    const int qrow = min(q0 + c, a.S - 1);
    const bf16_t* qp = (const bf16_t*)a.q + (int64_t)qrow * (a.q_ld ? a.q_ld : (int64_t)a.nh * DQK) + (int64_t)head * DQK;
#pragma unroll
    for (int ks = 0; ks < KST; ++ks) qf[ks] = as_frag(ld16(qp + ks * 16 + h * 8));
  }
  const int blk_last_q = min(qb + BR - 1, a.S - 1);
  const int last_tok = a.causal ? min(a.kv_offset + blk_last_q, a.kv_total - 1) : a.kv_total - 1;
  const int ntiles = last_tok / KV_PAGE_TOKENS + 1;

  float m = -INFINITY, l = 0.f;
  f32x16_t o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

  constexpr int KP = KV_PAGE_TOKENS * DQK / 8, VP = DV * KV_PAGE_TOKENS / 8;
  constexpr int KB = KP / 64, VB = VP / 64;
  constexpr int KBYTES = KP * 16, VBYTES = VP * 16;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  auto gload_k = [&](uint64_t page, int buf) __attribute__((always_inline)) {
    const uint64_t kb = page + a.kv.layer_off + (uint64_t)kvhd * KV_PAGE_TOKENS * (DQK * 2);
    char* dst = smem + buf * KBYTES;
#pragma unroll
    for (int i = 0; i < (KB + NWV - 1) / NWV; ++i) {
      const int blk = wave + i * NWV;
      if (KB % NWV == 0 || blk < KB)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(kb + blk * 1024 + lane * 16), (lds_ptr_t)(dst + blk * 1024), 16, 0, 0);
    }
  };
  auto gload_v = [&](uint64_t page, int buf) __attribute__((always_inline)) {
    const uint64_t vb = page + a.kv.layer_off + (uint64_t)a.kvh * KV_PAGE_TOKENS * (DQK * 2) + (uint64_t)kvhd * DV * (KV_PAGE_TOKENS * 2);
    char* dst = smem + 2 * KBYTES + buf * VBYTES;
#pragma unroll
    for (int i = 0; i < (VB + NWV - 1) / NWV; ++i) {
      const int blk = wave + i * NWV;
      if (VB % NWV == 0 || blk < VB)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(vb + blk * 1024 + lane * 16), (lds_ptr_t)(dst + blk * 1024), 16, 0, 0);
    }
  };
  typedef const __attribute__((address_space(4))) uint64_t* cptr64_t;
  const cptr64_t ptab = (cptr64_t)(uintptr_t)a.kv.page_ptrs;
  auto page_at = [&](int t) { return ptab[__builtin_amdgcn_readfirstlane(min(t, ntiles - 1))]; };

  const uint32_t lds0 = lds_addr_of(smem);
  const uint32_t k_lane = lds0 + hi * (KSF * 1024) + (h * 16 + c16) * 16;
  const uint32_t v_lane = lds0 + 2 * KBYTES + hi * 2048 + (h * 16 + c16) * 16;
  const uint32_t v_lane_last = lds0 + 2 * KBYTES + (h * 16 + c16) * 16;
  const float c2 = a.scale * 1.4426950408889634f;
  const int qpos = a.kv_offset + q0 + c;
  const int lim = a.causal ? min(qpos, a.kv_total - 1) : a.kv_total - 1;
  const int lim_min = a.causal ? min(a.kv_offset + q0, a.kv_total - 1) : a.kv_total - 1;
  auto active = [&](int t) { return !a.causal || t * KV_PAGE_TOKENS <= a.kv_offset + q0 + 31; };        // wave-uniform
  auto needs_mask = [&](int t) { return t * KV_PAGE_TOKENS + KV_PAGE_TOKENS - 1 > lim_min; };             // wave-uniform

  f32x16_t S[2][2];        // S[p] = raw S^T of the tile in flight with parity p (two 32-token halves)
  float alpha = 0.f, m2 = 0.f;   // of the tile whose scores wait in S[cur]

  // ---- building blocks ---------------------------------------------------------------------------------------------------------------
  // K fragment f = ks * 2 + jt of buffer kb; V^T fragment f = (kk * 2 + j) * DT + dt of buffer vb
#define KREAD(dst, f, kaddr) lds_read16_asm<(((f) & 1) * 2 * KSF + ((f) >> 2)) * 1024 + (((f) >> 1) & 1) * 512>(dst, kaddr)
#define VOFF(f) ((((f) % DT) * 4 + (((f) / DT) >> 1)) * 1024 + (((f) / DT) & 1) * 512)
  auto qk_plain = [&](f32x16_t (&sn)[2], int kbuf) __attribute__((always_inline)) {   // S^T = K . Q^T, nothing interleaved
    const uint32_t ka = k_lane + kbuf * KBYTES;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[jt][r] = 0.f;
    u32x4_t ring[RING];
    [&]<int... F>(std::integer_sequence<int, F...>) { (KREAD(ring[F], F, ka), ...); }(std::make_integer_sequence<int, RING>{});
    [&]<int... F>(std::integer_sequence<int, F...>) {
      ([&] {
        lds_wait_asm<(NF - 1 - F < RING - 1 ? NF - 1 - F : RING - 1)>(ring[F % RING]);
        sn[F & 1] = mfma32(as_frag(ring[F % RING]), qf[F >> 1], sn[F & 1]);
        if constexpr (F + RING < NF) KREAD(ring[F % RING], F + RING, ka);
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, NF>{});
  };
  auto mask_max = [&](f32x16_t (&sn)[2], int t, bool fresh) __attribute__((always_inline)) {   // mask (diagonal / last tile), row maximum, alpha, m2
    const int t0 = t * KV_PAGE_TOKENS;
    if (needs_mask(t)) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + jt * 32 + (r >> 2) * 8 + h * 4 + (r & 3) > lim) sn[jt][r] = -INFINITY;
    }
    auto sv = [&](int i) { return sn[i >> 4][i & 15]; };
    float tmax = max3_after_mfma(sv(0), sv(1), sv(2));
#pragma unroll
    for (int i = 3; i < 31; i += 2) tmax = max3(tmax, sv(i), sv(i + 1));
    tmax = max3(tmax, sv(31), sv(31));
    (void)fresh;
    const unsigned u = __float_as_uint(tmax);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    const float m_new = fmaxf(m, tmax);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    alpha = __builtin_amdgcn_exp2f((m - m_use) * c2);
    m2 = m_use * c2;
    m = m_new;
  };
  uint32_t pk[16];   // packed bf16 P of the current tile: pk[kk * 8 + r2] = registers 2 r2, 2 r2 + 1 of half kk
  f32x2_t psum2;
  auto prob_pair = [&](const f32x16_t (&sc)[2], int i) __attribute__((always_inline)) {   // scores 2i, 2i+1 of the 32: one packed fma, two v_exp, packed add, v_cvt_pk
    f32x2_t s2 = {sc[i >> 3][2 * (i & 7)], sc[i >> 3][2 * (i & 7) + 1]};
    pin(s2);    // not before this point of the volatile sequence ...
    const f32x2_t k2 = {c2, c2}, nm2 = {-m2, -m2};
    const f32x2_t e2 = __builtin_elementwise_fma(s2, k2, nm2);
    const f32x2_t p2 = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
    psum2 += p2;
    pin(psum2);
    pk[i] = pack_bf(p2[0], p2[1]);
    pin(pk[i]);   // ... and not after this one
  };
  auto pfrag = [&](int kk, int j) __attribute__((always_inline)) {   // registers 4j .. 4j+3 and 8+4j .. 8+4j+3 of half kk
    u32x4_t u = {pk[kk * 8 + 2 * j], pk[kk * 8 + 2 * j + 1], pk[kk * 8 + 4 + 2 * j], pk[kk * 8 + 5 + 2 * j]};
    return as_frag(u);
  };
  auto rescale = [&]() __attribute__((always_inline)) {
    l = l * alpha + (psum2[0] + psum2[1]);
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
  };
  auto pv_plain = [&](int vbuf) __attribute__((always_inline)) {
    const uint32_t va = v_lane + vbuf * VBYTES, val = v_lane_last + vbuf * VBYTES;
    u32x4_t ring[RING];
    [&]<int... F>(std::integer_sequence<int, F...>) {
      ((((DSF & 1) && F % DT == DT - 1) ? lds_read16_asm<VOFF(F)>(ring[F], val) : lds_read16_asm<VOFF(F)>(ring[F], va)), ...);
    }(std::make_integer_sequence<int, RING>{});
    [&]<int... F>(std::integer_sequence<int, F...>) {
      ([&] {
        lds_wait_asm<(NFV - 1 - F < RING - 1 ? NFV - 1 - F : RING - 1)>(ring[F % RING]);
        o[F % DT] = mfma32(as_frag(ring[F % RING]), pfrag((F / DT) >> 1, (F / DT) & 1), o[F % DT]);
        if constexpr (F + RING < NFV) {
          constexpr int G = F + RING;
          if ((DSF & 1) && G % DT == DT - 1) lds_read16_asm<VOFF(G)>(ring[F % RING], val);
          else lds_read16_asm<VOFF(G)>(ring[F % RING], va);
        }
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, NFV>{});
  };

  // ---- prologue: K(0), V(0), K(1) staged; S(0), its mask and maximum ------------------------------------------------------------------
  gload_k(page_at(0), 0);
  gload_v(page_at(0), 0);
  gload_k(page_at(1), 1);
  uint64_t pg1 = page_at(1), pg2 = page_at(2);   // pages of tiles t + 1 (V) and t + 2 (K) at the top of iteration t
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  qk_plain(S[0], 0);
  mask_max(S[0], 0, true);
  __syncthreads();   // K[0] is restaged by iteration 0

  auto step = [&]<int P>(std::integral_constant<int, P>, int t) __attribute__((always_inline)) {
    gload_k(pg2, t & 1);
    gload_v(pg1, (t + 1) & 1);
    pg1 = pg2;
    pg2 = page_at(t + 3);
    const bool act_t = active(t), act_n = t + 1 < ntiles && active(t + 1);
    if (act_t) {
      psum2 = f32x2_t{0.f, 0.f};
      if (act_n && !needs_mask(t + 1)) {
        // ---- phase A: S(t+1) = K(t+1) . Q^T on the matrix pipe, P(t) on the vector ALU -----------------------------------------------
        {
          const uint32_t ka = k_lane + ((t + 1) & 1) * KBYTES;
          f32x16_t (&sn)[2] = S[1 - P];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[jt][r] = 0.f;
          u32x4_t ring[RING];
          [&]<int... F>(std::integer_sequence<int, F...>) { (KREAD(ring[F], F, ka), ...); }(std::make_integer_sequence<int, RING>{});
          [&]<int... F>(std::integer_sequence<int, F...>) {
            ([&] {
              lds_wait_asm<(NF - 1 - F < RING - 1 ? NF - 1 - F : RING - 1)>(ring[F % RING]);
              sn[F & 1] = mfma32(as_frag(ring[F % RING]), qf[F >> 1], sn[F & 1]);
              pin(sn[F & 1]);
              if constexpr (F + RING < NF) KREAD(ring[F % RING], F + RING, ka);
              // the 16 score pairs of P(t) spread over the NF MFMA slots
              if constexpr (16 * F / NF != 16 * (F + 1) / NF) {
#pragma unroll
                for (int i = 16 * F / NF; i < 16 * (F + 1) / NF; ++i) prob_pair(S[P], i);
              }
              __builtin_amdgcn_sched_barrier(0);
            }(), ...);
          }(std::make_integer_sequence<int, NF>{});
        }
        rescale();
        // ---- phase B: O += V(t)^T . P(t)^T on the matrix pipe, the row maximum of S(t+1) on the vector ALU --------------------------
        {
          const uint32_t va = v_lane + (t & 1) * VBYTES, val = v_lane_last + (t & 1) * VBYTES;
          f32x16_t (&sn)[2] = S[1 - P];
          auto sv = [&](int i) { return sn[i >> 4][i & 15]; };
          float tmax = 0.f;
          u32x4_t ring[RING];
          [&]<int... F>(std::integer_sequence<int, F...>) {
            ((((DSF & 1) && F % DT == DT - 1) ? lds_read16_asm<VOFF(F)>(ring[F], val) : lds_read16_asm<VOFF(F)>(ring[F], va)), ...);
          }(std::make_integer_sequence<int, RING>{});
          [&]<int... F>(std::integer_sequence<int, F...>) {
            ([&] {
              lds_wait_asm<(NFV - 1 - F < RING - 1 ? NFV - 1 - F : RING - 1)>(ring[F % RING]);
              o[F % DT] = mfma32(as_frag(ring[F % RING]), pfrag((F / DT) >> 1, (F / DT) & 1), o[F % DT]);
              pin(o[F % DT]);
              if constexpr (F + RING < NFV) {
                constexpr int G = F + RING;
                if ((DSF & 1) && G % DT == DT - 1) lds_read16_asm<VOFF(G)>(ring[F % RING], val);
                else lds_read16_asm<VOFF(G)>(ring[F % RING], va);
              }
              // the 16 v_max3 of the 32 scores of S(t+1) spread over the first slots (the QK^T MFMAs finished a phase ago)
              if constexpr (F == 0) tmax = max3_after_mfma(sv(0), sv(1), sv(2));
              else if constexpr (F <= 14) tmax = max3v(tmax, sv(2 * F + 1), sv(2 * F + 2));
              else if constexpr (F == 15) tmax = max3v(tmax, sv(31), sv(31));
              __builtin_amdgcn_sched_barrier(0);
            }(), ...);
          }(std::make_integer_sequence<int, NFV>{});
          if constexpr (NFV < 16) {   // fewer MFMA slots than v_max3 (ViT 12, audio 8): the rest of the chain behind them
#pragma unroll
            for (int F = NFV; F < 16; ++F) tmax = F <= 14 ? max3(tmax, sv(2 * F + 1), sv(2 * F + 2)) : max3(tmax, sv(31), sv(31));
          }
          const unsigned u = __float_as_uint(tmax);
          const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
          tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
          const float m_new = fmaxf(m, tmax);
          const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
          alpha = __builtin_amdgcn_exp2f((m - m_use) * c2);
          m2 = m_use * c2;
          m = m_new;
        }
      } else {
        // the wave's last tiles (diagonal, or the launch's last page): the phases one after the other
#pragma unroll
        for (int i = 0; i < 16; ++i) prob_pair(S[P], i);
        rescale();
        pv_plain(t & 1);
        if (act_n) {
          qk_plain(S[1 - P], (t + 1) & 1);
          mask_max(S[1 - P], t + 1, true);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of K(t+2) and V(t+1) has landed
    __syncthreads();
  };
  for (int t = 0; t < ntiles; t += 2) {
    step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) step(std::integral_constant<int, 1>{}, t + 1);
  }
#undef KREAD
#undef VOFF

  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = 1.0f / lt;
  constexpr int EPITCH = DT * 64 + 16;
  if (a.epi_rows) {
    char* wb = smem + wave * (32 * EPITCH);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        uint2 w;
        w.x = pack_bf(o[dt][r4 * 4 + 0] * inv, o[dt][r4 * 4 + 1] * inv);
        w.y = pack_bf(o[dt][r4 * 4 + 2] * inv, o[dt][r4 * 4 + 3] * inv);
        *reinterpret_cast<uint2*>(wb + c * EPITCH + (dt * 32 + r4 * 8 + h * 4) * 2) = w;
      }
    const int chunk = lane & 15;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4), qr = q0 + row;
      if (qr < a.S && chunk * 8 < a.d) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wb + row * EPITCH + chunk * 16);
        *reinterpret_cast<u32x4_t*>((bf16_t*)a.o + ((int64_t)qr * a.nh + head) * a.d + chunk * 8) = v;
      }
    }
    return;
  }
  const int qr = q0 + c;
  if (qr < a.S) {
    bf16_t* op = (bf16_t*)a.o + ((int64_t)qr * a.nh + head) * a.d;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int dim = dt * 32 + r4 * 8 + h * 4;
        if (dim < a.d) {
          uint2 w;
          w.x = pack_bf(o[dt][r4 * 4 + 0] * inv, o[dt][r4 * 4 + 1] * inv);
          w.y = pack_bf(o[dt][r4 * 4 + 2] * inv, o[dt][r4 * 4 + 3] * inv);
          *reinterpret_cast<uint2*>(op + dim) = w;
        }
      }
  }
}

}  // namespace

// Launcher of the 32-row forms; `a.epi_rows` is already decided by launch_attn_prefill (kernels_attn.hip).  pipelined = the software-
// pipelined kernel (AHA_ATTN_SMX=5), else the phase-by-phase one (4).
void launch_attn_prefill32(const AttnPrefillArgs& a_in, hipStream_t st, bool pipelined) {
  AttnPrefillArgs a = a_in;
  static const int nw_env = [] { const char* e = getenv("AHA_ATTN32_WAVES"); return e ? atoi(e) : 0; }();
  static const int sched_env = [] { const char* e = getenv("AHA_ATTN_SCHED"); return e ? atoi(e) : 1; }();
  // 4 waves = 128 q rows per staged tile, two blocks per CU (the two waves of a SIMD belong to DIFFERENT blocks)
  int nwv = 4;
  if (a.d == 128 && (nw_env == 8 || (nw_env == 2 && !pipelined))) nwv = nw_env;
  const int br = 32 * nwv, nqb = (a.S + br - 1) / br;
  a.nqb = (sched_env && a.kvh % 8 == 0 && a.nh % a.kvh == 0) ? nqb : 0;
  dim3 grid = a.nqb ? dim3(nqb * a.nh) : dim3(nqb, a.nh), block(nwv * 64);
  static const int abl = [] { const char* e = getenv("AHA_ATTN_ABL"); return e ? atoi(e) : 0; }();
  static const int stg_env = [] { const char* e = getenv("AHA_ATTN32_STG"); return e ? atoi(e) : 0; }();
  static const int pf_env = [] { const char* e = getenv("AHA_ATTN32_PF"); return e ? atoi(e) : 1; }();
  if (a.d == 128) {
    // two stages, or the epilogue's row buffers (32 rows x 272 B per wave) where those are larger (8 waves)
    const size_t lds = std::max<size_t>(2 * KV_PAGE_TOKENS * 2 * (128 + 128), (size_t)nwv * 32 * (4 * 64 + 16));
    if (pipelined) {
      if (nwv == 8) hipLaunchKernelGGL((attn_prefill32p_kernel<128, 128, 8, 8>), grid, block, lds, st, a);
      else hipLaunchKernelGGL((attn_prefill32p_kernel<128, 128, 8, 4>), grid, block, lds, st, a);
    } else if (abl == 2 && nwv == 4) hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 4, 2>), grid, block, lds, st, a);
    else if (abl == 3 && nwv == 4) hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 4, 3>), grid, block, lds, st, a);
    else if (abl == 4 && nwv == 4) hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 4, 4>), grid, block, lds, st, a);
    else if (pf_env == 2 && nwv == 8) {
      const size_t lds3 = 3 * KV_PAGE_TOKENS * 2 * (128 + 128);
      static bool once = false;
      if (!once) { hipFuncSetAttribute((const void*)attn_prefill32_kernel<128, 128, 8, 8, 0, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); once = true; }
      hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 8, 0, 0, 2>), grid, block, lds3, st, a);
    }
    else if (stg_env && nwv == 8) hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 8, 0, 1>), grid, block, lds, st, a);
    else if (stg_env && nwv == 4) hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 4, 0, 1>), grid, block, lds, st, a);
    else if (nwv == 8) hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 8>), grid, block, lds, st, a);
    else if (nwv == 2) hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 2>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((attn_prefill32_kernel<128, 128, 8, 4>), grid, block, lds, st, a);
  } else if (a.d == 64) {   // Qwen3-ASR audio encoder
    const size_t lds = 2 * KV_PAGE_TOKENS * 2 * (64 + 64);
    if (pipelined) hipLaunchKernelGGL((attn_prefill32p_kernel<64, 64, 4, 4>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((attn_prefill32_kernel<64, 64, 4, 4>), grid, block, lds, st, a);
  } else {                  // head_dim 72 (Qwen3-VL ViT): Q / K rows padded to 96, V block to 80
    const size_t lds = 2 * KV_PAGE_TOKENS * 2 * (96 + 80);
    if (pipelined) hipLaunchKernelGGL((attn_prefill32p_kernel<96, 80, 5, 4>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((attn_prefill32_kernel<96, 80, 5, 4>), grid, block, lds, st, a);
  }
}

}  // namespace aha
