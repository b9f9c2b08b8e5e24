// GQA attention over the paged KV cache: flash-style, no S x S score tensor, no repeat_kv
// copies, no mask tensor.  Reference semantics being reproduced (eager_attention_forward,
// /root/reference/src/models/common/modules.rs:757-813; repeat_kv /root/reference/src/utils/tensor_utils.rs:108-124;
// causal mask tensor_utils.rs:78-106):
//     scores = bf16(q . k^T) ; scores = bf16(scores * scale) ; (+ -inf above the diagonal) ; softmax ; P . v ; -> bf16
// q head i reads kv head i / g.  Softmax runs online in f32; P feeds the MFMA as bf16.  The two bf16 roundings of the scores are what
// the prefill kernel's score chains 0 / 1 reproduce (the bit-faithful forms, opt-in); the default chain of the prefill kernels since
// round 5 (SMX 3) and of the decode kernels since round 6 keeps the scores in f32 through scale, mask, maximum and exponential
// (attn_common.h; DESIGN.md section 2, deviation (iii)): one convention for a cache position, however it arrived.
//
// Fragment scheme (v_mfma_f32_16x16x32_bf16, wave64; G = lane>>4, c = lane&15):
//   S^T tile = K . Q^T :  A = K   (row = token c, k = dims G*8..+8  -> 16 B piece `lane` of a fragment-major K fragment)
//                         B = Q^T (col = q row c, k = dims G*8..+8  -> 16 B straight from a q row)
//                         C[reg] = S[token G*4+reg][q c]            -> softmax statistics are per lane column
//   O^T tile = V^T . P^T: A = V^T (row = dim c,  k = 8 token slots  -> 16 B piece `lane` of a fragment-major V fragment)
//                         B = P^T (col = q c,    k = the two S^T fragments of token sub-tiles 2kk, 2kk+1, in registers)
//                         C[reg] = O[q c][dim G*4+reg]
// so no operand ever needs a transpose or a cross-lane shuffle; page layout and the V slot permutation: common.h.
#include <stdio.h>
#include <stdlib.h>

#include "attn_decode_body.h"

namespace aha {

namespace {

// ---- prefill: NWV waves x QT*16 q rows per block, K / V^T page staged in LDS and shared by the waves ----------------
// DQK = padded head dim of Q/K rows (multiple of 32), DV = padded head dim of the V block (multiple of 16).
// Text decoder: 128/128.  ViT (head_dim 72): 96/80, pad lanes are zero in Q, K and V (csrc/kernels_vit.hip).
// QT q sub-tiles per wave share every K / V^T fragment read from LDS (LDS reads per MFMA 1 -> 1/QT) and every
// global->LDS staging pass is amortised over 64*QT query rows.
// NWV waves per block (4 or 8) share every staged K / V^T tile (8 waves = 128 q rows per tile halve the staging traffic).
// The waves of a block move in lockstep (one barrier per tile), so MFMA, softmax VALU and LDS phases only overlap ACROSS
// blocks: the kernel is held to 128 VGPRs (amdgpu_waves_per_eu 4) so that two 8-wave blocks are resident per CU.
// TRACE (debug, AHA_ATTN_PTRACE=1): waves 0 and 4 of the middle block add up the shader cycles they spend in each part of the loop body
// ABL (debug, AHA_ATTN_ABL, results wrong by construction): 2 = no softmax arithmetic, 3 = no staging of the next tile, 4 = no MFMAs
// SMX: the score chain (attn_common.h softmax_scores): 0 / 1 the reference's two roundings (1: scale multiply on the matrix pipe), 3 f32
template <int DQK, int DV, int QT, int NWV, bool TRACE = false, int ABL = 0, int SMX = 0>
__global__ __launch_bounds__(NWV * 64) __attribute__((amdgpu_waves_per_eu((NWV == 8 || DQK < 128) ? 4 : 3, 4))) void attn_prefill_kernel(AttnPrefillArgs a, unsigned long long* trace = nullptr) {
  constexpr int KS = DQK / 32, DS = DV / 16;
  // LSUM (ViT, f32 chain): the softmax row sum comes out of the P.V MFMAs -- V^T row 72 (a pad dim: head_dim 72 in an 80-row block) is 1.0
  // on every real token (kernels_vit.hip), so output row 72 = sum of the bf16 probabilities the numerator uses, rescaled with it: 8 packed
  // adds and the l update less per tile.  Same box, A B A B: 120.3 / 120.4 against 128.9 / 128.5 us for the cfg 3 ViT launch.
  // (Also tried there: the third 32-dim step of QK^T -- 8 real dims + 24 zeros -- as ONE v_mfma_f32_16x16x16_bf16 over dims 64..79 with
  // 8-byte operands: 126.4 against 116.2 us, slower -- four dependent MFMAs behind 8-byte LDS reads outside the fragment ring.)
  // Only the packer that wrote the pages can promise the ones row: a.v_ones_row (set by vision_tower.hip); without it the kernel keeps its own sum.
  constexpr bool LSUM = SMX == 3 && DQK == 96 && DV == 80;
  constexpr int RING = (DQK >= 128) ? 6 : 4;  // LDS fragment reads in flight ahead of the MFMA that consumes them
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x [K tile | V^T tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, c = lane & 15;
  // Block order.  Workgroups go round-robin over the 8 XCDs by linear id, and each XCD has its own 4 MB L2: in the XCD-aware
  // order (a.nqb > 0; kv heads a multiple of 8) XCD x only ever works on kv heads x, x+8, ..., all q heads of a kv head ride
  // together, and consecutive slots of one XCD take consecutive q blocks -- so the ~64 blocks resident on an XCD stream the same
  // K / V^T pages of one kv head in near lockstep and each page crosses the fabric once per XCD instead of once per block
  // (at 41 k tokens the per-kv-head K/V is 21 MB: in grid order the blocks drift apart and every tile comes from Infinity Cache).
  // Causal launches hand out the long (late) q blocks first.
  // Two causal segments in one launch (a.S2 > 0: a context-parallel rank's early and late chunk, csrc/model.hip): q / o rows
  // [0, S) see the cache through kv_offset, rows [S, S + S2) through kv_offset2; a block belongs to ONE segment, the late segment's
  // (longer) blocks go first and the early segment's fill the tail of the late one's last round.
  int head, qblk, seg_rows = a.S, seg_off = a.kv_offset, seg_tot = a.kv_total, seg_row0 = 0;
  if (a.nqb > 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = a.nh / a.kvh, hpx = a.nh >> 3;  // q heads per kv head / q heads per XCD
    const int hq = slot % hpx, qi = slot / hpx;
    head = (xcd + 8 * (hq / g)) * g + hq % g;
    qblk = a.causal ? a.nqb - 1 - qi : qi;
    if (a.S2 > 0) {  // launcher: causal, nqb = q blocks of both segments
      const int nqb2 = (a.S2 + 16 * QT * NWV - 1) / (16 * QT * NWV);
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
  const s16x4_t dg = diag_frag(a.scale, G, c);
  // (a static `s_setprio 1` for the second-dispatched half of an 8-wave block -- the age-arbitration loser -- measured as noise:
  // profiles/r04_attn_prefill.md)
  const int qb = qblk * (16 * QT * NWV);  // first q row of the block
  const int q0 = qb + wave * (16 * QT);        // first q row of the wave
  bf16x8_t qf[QT][KS];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qrow = seg_row0 + min(q0 + t * 16 + c, seg_rows - 1);
    const bf16_t* qp = (const bf16_t*)a.q + (int64_t)qrow * (a.q_ld ? a.q_ld : (int64_t)a.nh * DQK) + (int64_t)head * DQK;
#pragma unroll
    for (int k4 = 0; k4 < KS; ++k4) qf[t][k4] = as_frag(ld16(qp + k4 * 32 + G * 8));
    // q-norm + RoPE of Q in the Q load (round 6; AttnPrefillArgs::q_norm_w): `q` is the raw q heads of the qkv GEMM.  The lane holds dims
    // k4 * 32 + G * 8 + j of its row: the rotate_half partner (dims +- 64) is fragment k4 +- 2 of the SAME lane, and the sum of squares
    // follows qknorm_rope_rows_kernel's butterfly over the 64 partial terms fma(x[l], x[l], x[l + 64]^2), l = k4 * 32 + G * 8 + j -- partner
    // l ^ 32 = the other fragment, l ^ 16 / l ^ 8 = lane ^ 32 / lane ^ 16, l ^ 4, 2, 1 in registers -- so the sum, and with it every later
    // expression (kernels_elem.hip), has the same bits: tests/test_model_gpu.py (digest with AHA_ATTN_QFUSE=0).
    if constexpr (DQK == 128) {
      if (a.q_norm_w != nullptr) {
        float x[4][8], w[4][8], cs[2][8], sn[2][8];
        const bf16_t* tp = (const bf16_t*)a.q_rope_tab + (int64_t)qrow * 128;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          union { bf16x8_t b; u32x4_t u; } xv;
          xv.b = qf[t][k4];
          const u32x4_t wv = ld16((const bf16_t*)a.q_norm_w + k4 * 32 + G * 8);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            x[k4][2 * j] = lo_bf(xv.u[j]); x[k4][2 * j + 1] = hi_bf(xv.u[j]);
            w[k4][2 * j] = lo_bf(wv[j]); w[k4][2 * j + 1] = hi_bf(wv[j]);
          }
          if (k4 < 2) {
            const u32x4_t c4 = ld16(tp + k4 * 32 + G * 8), s4 = ld16(tp + 64 + k4 * 32 + G * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              cs[k4][2 * j] = lo_bf(c4[j]); cs[k4][2 * j + 1] = hi_bf(c4[j]);
              sn[k4][2 * j] = lo_bf(s4[j]); sn[k4][2 * j + 1] = hi_bf(s4[j]);
            }
          }
        }
        float sq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sq[j] = fmaf(x[0][j], x[0][j], x[2][j] * x[2][j]) + fmaf(x[1][j], x[1][j], x[3][j] * x[3][j]);   // l ^ 32
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // l ^ 16
          const unsigned u = __float_as_uint(sq[j]);
          const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
          sq[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {   // l ^ 8
          const unsigned u = __float_as_uint(sq[j]);
          const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
          sq[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        const float u0 = sq[0] + sq[4], u1 = sq[1] + sq[5], u2 = sq[2] + sq[6], u3 = sq[3] + sq[7];   // l ^ 4
        const float v0 = u0 + u2, v1 = u1 + u3;                                                        // l ^ 2
        const float ss = v0 + v1;                                                                      // l ^ 1
        const float rinv = 1.0f / sqrtf(ss / 128.0f + a.q_eps);
        float xn[4][8];
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
          for (int j = 0; j < 8; ++j) xn[k4][j] = rbf(x[k4][j] * rinv * w[k4][j]);
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4) {
          u32x4_t lo4, hi4;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float yl[2], yh[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int jj = 2 * j + e;
              yl[e] = rbf(rbf(xn[k4][jj] * cs[k4][jj]) + rbf(-xn[k4 + 2][jj] * sn[k4][jj]));
              yh[e] = rbf(rbf(xn[k4 + 2][jj] * cs[k4][jj]) + rbf(xn[k4][jj] * sn[k4][jj]));
            }
            lo4[j] = pack_bf(yl[0], yl[1]);
            hi4[j] = pack_bf(yh[0], yh[1]);
          }
          qf[t][k4] = as_frag(lo4);
          qf[t][k4 + 2] = as_frag(hi4);
        }
      }
    }
  }
  const int blk_last_q = min(qb + 16 * QT * NWV - 1, seg_rows - 1);
  const int last_tok = a.causal ? min(seg_off + blk_last_q, seg_tot - 1) : seg_tot - 1;
  const int ntiles = last_tok / KV_PAGE_TOKENS + 1;

  float m[QT], l[QT];
  f32x4_t o[QT][DS];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -INFINITY;
    l[t] = 0.f;
#pragma unroll
    for (int i = 0; i < DS; ++i) o[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  // Software pipeline over KV pages: the next page travels global -> LDS by LDS-DMA (global_load_lds_dwordx4: no register round
  // trip, no ds_write pass) into the other stage while the MFMAs of the current one run (one vmcnt(0) + barrier per page).
  // LDS layout: fragment-major.  Every K (16 tokens x 32 dims) and V^T (16 dims x 32 token slots) MFMA operand is one
  // contiguous 1 KB block whose 16-byte piece l belongs to lane l (row c = l & 15, 8-element chunk G = l >> 4), so a fragment
  // read is ds_read_b128 at base + lane * 16: conflict-free.  The KV pages in HBM have the same fragment-major layout (common.h),
  // so the LDS image of a tile is a byte copy of the kv head's K and V blocks of the page and a wave-wide DMA piece is 1 KB
  // contiguous on both sides.
  // (A padded row-major tile cannot be conflict-free: the b128 lane groups {0-3, 12-15, 20-27}, ... put rows 0-3,12-15 of chunk
  // G and rows 4-11 of chunk G+1 in one LDS cycle, which collide for every row pitch.)
  // The staging registers this frees (4 x 16 bytes per lane) pay for a deeper fragment-read ring: the MFMA phases were bound by
  // the LDS round trip times the reads in flight per wave (3: ~23 cycles per 16-cycle MFMA).
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
      const int blk = wave + i * NWV;   // wave-uniform: the surplus waves of a ragged count (ViT: 12 + 10 blocks, 8 waves) skip
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
  // the page table is read through the constant address space with a wave-uniform index: a scalar load (s_load_dwordx2), no
  // vector-memory request and no VGPRs for the address of the next page
  typedef const __attribute__((address_space(4))) uint64_t* cptr64_t;
  const cptr64_t ptab = (cptr64_t)(uintptr_t)a.kv.page_ptrs;
  gload(ptab[0], 0);
  // Everything requested so far (the q fragments above all) is retired here, once: otherwise the waits the compiler places in
  // the loop must assume the q loads may still be the newest requests (on the path that issues no prefetch) and turn into
  // vmcnt(0) in the middle of the QK^T phase, i.e. wait for the tile prefetch issued a few instructions earlier.
  uint64_t pg_next = ptab[__builtin_amdgcn_readfirstlane(min(1, ntiles - 1))];
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  unsigned tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_prev = 0;   // wave-uniform (scalar registers)
  const bool tr_on = TRACE && blockIdx.x == gridDim.x / 2 && (wave == 0 || wave == 4);
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (TRACE) {
      if (tr_on) {
        const unsigned now = __builtin_amdgcn_readfirstlane((unsigned)__builtin_readcyclecounter());
        if (k >= 0) tr_acc[k] += now - tr_prev;
        tr_prev = now;
      }
    }
  };
  // State of a tile between its two halves:
  f32x4_t st[QT][4];      // raw S^T, then the rounded + masked scores
  float alpha[QT], m2[QT];
  bool act[QT];           // wave-uniform: the q sub-tile sees anything of the tile (causal)
  bool any = false;
  constexpr int NFV = 2 * DS;

  // first half of a tile: S^T = K . Q^T and the score side of the softmax (rounding chain, mask, running maximum)
  auto scores_part = [&](int tile) __attribute__((always_inline)) {
    const char* ks = smem + (tile & 1) * STAGE_BYTES;
    const int t0 = tile * KV_PAGE_TOKENS;
    any = false;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      act[t] = !a.causal || t0 <= seg_off + q0 + t * 16 + 15;
      any |= act[t];
    }
    if (!any) return;
    // the 4 token sub-tiles are 4 independent accumulators and the fragments are consumed k4-major, so consecutive MFMAs never
    // depend on each other (sub-major order made chains of KS dependent MFMAs, each stalling for the previous one's result);
    // fragment reads run RING fragments ahead of their MFMA (a read used to be followed by s_waitcnt lgkmcnt(0) + its MFMA)
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) st[t][sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int NF = 4 * KS;
      auto kread = [&](int f) {  // fragment f = (k4, sub) = (f / 4, f % 4)
        return *reinterpret_cast<const u32x4_t*>(ks + (((f & 3) * KS + (f >> 2)) * 64 + lane) * 16);
      };
      u32x4_t ring[RING];
#pragma unroll
      for (int f = 0; f < RING; ++f) ring[f] = kread(f);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const bf16x8_t kf = as_frag(ring[f % RING]);
#pragma unroll
        for (int t = 0; t < QT; ++t)
          if (act[t]) {
            if (ABL != 4) st[t][f & 3] = mfma16(kf, qf[t][f >> 2], st[t][f & 3]);
            else asm volatile("" :: "v"(kf));
          }
        if (f + RING < NF) ring[f % RING] = kread(f + RING);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stamp(1);   // QK^T
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (!act[t]) continue;
      const int qpos = seg_off + q0 + t * 16 + c;  // cache position of this lane's q row
      const int lim = a.causal ? min(qpos, seg_tot - 1) : seg_tot - 1;
      // Interior tiles (every token visible to every q row of the sub-tile: all but the diagonal / last tile) skip the
      // per-element predicate -- two compares and a select per score in a VALU-bound softmax.
      const int lim_min = a.causal ? min(seg_off + q0 + t * 16, seg_tot - 1) : seg_tot - 1;
      if (ABL == 2) { alpha[t] = 1.f; m2[t] = 0.f; continue; }
      if (t0 + KV_PAGE_TOKENS - 1 <= lim_min)
        softmax_scores<SMX>(st[t], a.scale, [](int) { return true; }, G, m[t], alpha[t], m2[t], dg);
      else
        softmax_scores<SMX>(st[t], a.scale, [&](int tk) { return t0 + tk <= lim; }, G, m[t], alpha[t], m2[t], dg);
    }
    stamp(2);   // scores
  };
  // second half: p = e^(s - m), row sums, O^T = O^T * alpha + V^T . P^T
  auto probs_part = [&](int tile) __attribute__((always_inline)) {
    if (!any) return;
    const char* vs = smem + (tile & 1) * STAGE_BYTES + KP * 16;
    auto vread = [&](int f) {  // fragment f = (kk, ds) = (f / DS, f % DS)
      return *reinterpret_cast<const u32x4_t*>(vs + (((f % DS) * 2 + f / DS) * 64 + lane) * 16);
    };
    u32x4_t vring[RING];   // the first V^T fragments travel while the exponentials run
#pragma unroll
    for (int f = 0; f < RING; ++f) vring[f] = vread(f);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t pf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (!act[t]) continue;
      if (ABL == 2) {
        u32x4_t u0 = {__float_as_uint(st[t][0][0]), __float_as_uint(st[t][0][1]), __float_as_uint(st[t][1][0]), __float_as_uint(st[t][1][1])};
        u32x4_t u1 = {__float_as_uint(st[t][2][0]), __float_as_uint(st[t][2][1]), __float_as_uint(st[t][3][0]), __float_as_uint(st[t][3][1])};
        pf[t][0] = as_frag(u0); pf[t][1] = as_frag(u1);
        l[t] += 1.f;
      } else
      if (LSUM && a.v_ones_row) softmax_probs<false>(st[t], m2[t], alpha[t], l[t], pf[t], a.scale * 1.4426950408889634f);
      else softmax_probs<true>(st[t], m2[t], alpha[t], l[t], pf[t], SMX == 3 ? a.scale * 1.4426950408889634f : 1.4426950408889634f);
      // once the running max has settled alpha is exactly 1 in every lane of the wave: skip the DS*4 multiplies (x * 1 == x)
      if (__builtin_amdgcn_ballot_w64(alpha[t] != 1.f) != 0) {
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) o[t][ds] *= alpha[t];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp(3);   // probabilities
#pragma unroll
    for (int f = 0; f < NFV; ++f) {  // kk-major: DS independent accumulators in a row
      const bf16x8_t vf = as_frag(vring[f % RING]);
#pragma unroll
      for (int t = 0; t < QT; ++t)
        if (act[t]) {
          if (ABL != 4) o[t][f % DS] = mfma16(vf, pf[t][f / DS], o[t][f % DS]);
          else asm volatile("" :: "v"(vf), "v"(pf[t][f / DS]));
        }
      if (f + RING < NFV) vring[f % RING] = vread(f + RING);
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(4);   // P.V
  };
  auto prefetch = [&](int tile) __attribute__((always_inline)) {
    // unconditional prefetch (the last iteration re-requests its own tile into the free stage and nobody reads it): one code path
    // (the page address itself was requested one iteration earlier: its latency is not in front of the prefetch)
    if (ABL != 3) gload(pg_next, (tile + 1) & 1);
    pg_next = ptab[__builtin_amdgcn_readfirstlane(min(tile + 2, ntiles - 1))];
    stamp(0);   // loop top
  };
  auto stage_and_barrier = [&](int) __attribute__((always_inline)) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of the next tile has landed
    stamp(5);   // wait for the staging DMA
    __syncthreads();
    stamp(6);   // barrier
  };
  stamp(-1);
  // (A "ping-pong" schedule -- the two wave rows of a block half a tile apart, two barriers per tile, so that one row's
  // MFMA + fragment-read phase meets the other's softmax VALU phase -- was built and measured on MI355X: parity-green, 0.785 vs
  // 0.791 ms at S = 8192 causal, 14.87 vs 14.89 ms at 41 k: no gain, the halves do not balance and the second barrier costs what
  // the overlap returns.  Ablations of this kernel at S = 8192 (0.775 ms): no softmax arithmetic 0.48, no staging 0.66, no MFMAs
  // 0.38 -- the parts add up, whatever the phase alignment.)
  for (int tile = 0; tile < ntiles; ++tile) {
    prefetch(tile);
    scores_part(tile);
    probs_part(tile);
    stage_and_barrier(tile);
  }
  if (TRACE) {
    if (tr_on && lane == 0) {
      for (int k = 0; k < 7; ++k) trace[(wave >> 2) * 8 + k] = tr_acc[k];
      trace[(wave >> 2) * 8 + 7] = (unsigned long long)ntiles;
    }
  }
  // Epilogue.  The accumulator layout gives a lane 4 consecutive dims of ONE q row per fragment: stored from there, an instruction
  // writes 8-byte pieces 32 bytes apart on 16 rows.  Row order instead (a.epi_rows; the lesson of the GEMM epilogues, DESIGN.md
  // section 5 round 4): the wave passes its 16 x DV tile through its own slice of the (now free) staging LDS and stores whole rows,
  // 16 bytes per lane, 256-byte (text) / 144-byte (ViT) runs.
  constexpr int EPITCH = DV * 2 + 16;   // bytes per LDS row: 16-byte aligned, rows 4 dwords apart in the banks
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    // LSUM: row 72 = fragment ds 4, lane group G = 2, register 0 of q column c
    const float lt = (LSUM && a.v_ones_row) ? __shfl(o[t][DS - 1][0], 32 + c, 64) : group_sum(l[t]);
    const float inv = 1.0f / lt;
    if (a.epi_rows) {
      char* wb = smem + wave * (16 * EPITCH);
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        uint2 w;
        w.x = pack_bf(o[t][ds][0] * inv, o[t][ds][1] * inv);
        w.y = pack_bf(o[t][ds][2] * inv, o[t][ds][3] * inv);
        *reinterpret_cast<uint2*>(wb + c * EPITCH + (ds * 16 + G * 4) * 2) = w;
      }
      // (same wave wrote and reads: no barrier, the LDS queue is in order)
      const int chunk = lane & 15;              // 16-byte piece of a row
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 4 + (lane >> 4), qr = q0 + t * 16 + row;
        if (qr < seg_rows && chunk * 8 < a.d) {     // a.d = real head dim of the output rows: a multiple of 8 on this path (launcher)
          const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wb + row * EPITCH + chunk * 16);
          *reinterpret_cast<u32x4_t*>((bf16_t*)a.o + ((int64_t)(seg_row0 + qr) * a.nh + head) * a.d + chunk * 8) = v;
        }
      }
      continue;
    }
    const int qr = q0 + t * 16 + c;
    if (qr < seg_rows) {
      bf16_t* op = (bf16_t*)a.o + ((int64_t)(seg_row0 + qr) * a.nh + head) * a.d;  // a.d = real head dim of the output rows
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) {
        if (ds * 16 + G * 4 < a.d) {  // head dims come in multiples of 4, so a 4-wide group is all-in or all-out
          uint2 w;
          w.x = pack_bf(o[t][ds][0] * inv, o[t][ds][1] * inv);
          w.y = pack_bf(o[t][ds][2] * inv, o[t][ds][3] * inv);
          *reinterpret_cast<uint2*>(op + ds * 16 + G * 4) = w;
        }
      }
    }
  }
}

// ---- decode: one wave = one work unit striding over KV pages; K / V^T fragments straight from HBM --------------
// grid (kvh, nsplit), 4 waves per block => 4*nsplit units per kv head; unit u takes pages u, u + nunits, ...
// All g = nh/kvh q heads of the kv head ride along as MFMA columns, so every KV byte is read exactly once.
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecodeArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, c = lane & 15;
  const int kvhd = blockIdx.x;
  const int g = a.nh / a.kvh;
  const int nunits = gridDim.y * 4;
  const int unit = blockIdx.y * 4 + wave;
  const int L = *a.kv_len;
  const int npages = (L + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;

  bf16x8_t qf[4];
  {
    const bf16_t* qp = (const bf16_t*)a.q + (int64_t)(kvhd * g + min(c, g - 1)) * 128;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      u32x4_t v = ld16(qp + k4 * 32 + G * 8);
      if (c >= g) v = u32x4_t{0u, 0u, 0u, 0u};
      qf[k4] = as_frag(v);
    }
  }
  float m = -INFINITY, l = 0.f;
  f32x4_t o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int page = unit; page < npages; page += nunits) {
    const char* base = reinterpret_cast<const char*>(a.kv.page_ptrs[page] + a.kv.layer_off);
    const char* kb = base + (size_t)kvhd * KV_PAGE_TOKENS * 256;
    const char* vb = base + (size_t)a.kvh * KV_PAGE_TOKENS * 256 + (size_t)kvhd * 128 * (KV_PAGE_TOKENS * 2);
    u32x4_t kf[4][4], vf[8][2];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) kf[sub][k4] = ld_nt16(kb + ((sub * 4 + k4) * 64 + lane) * 16);   // fragment-major pages
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) vf[ds][kk] = ld_nt16(vb + ((ds * 2 + kk) * 64 + lane) * 16);

    f32x4_t st[4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      st[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) st[sub] = mfma16(as_frag(kf[sub][k4]), qf[k4], st[sub]);
    }
    float alpha;
    bf16x8_t pf[2];
    const int t0 = page * KV_PAGE_TOKENS;
    softmax_tile(st, a.scale, [&](int t) { return t0 + t < L; }, G, m, l, alpha, pf);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      o[ds] *= alpha;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) o[ds] = mfma16(as_frag(vf[ds][kk]), pf[kk], o[ds]);
    }
  }
  l = group_sum(l);
  if (c < g) {
    const int head = kvhd * g + c;
    float* po = a.part_o + ((int64_t)unit * a.nh + head) * 128;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
      *reinterpret_cast<float4*>(po + ds * 16 + G * 4) = make_float4(o[ds][0], o[ds][1], o[ds][2], o[ds][3]);
    if (G == 0) {
      a.part_ml[((int64_t)unit * a.nh + head) * 2 + 0] = m;
      a.part_ml[((int64_t)unit * a.nh + head) * 2 + 1] = l;
    }
  }
}

// merge the per-unit partials: o = sum_u e^{m_u - M} o_u / sum_u e^{m_u - M} l_u  -> bf16 (the P.V matmul output tensor)
// One block per head; nunits <= 256.  The per-unit weights e^{m_u - M} are computed once (one unit per thread) and
// shared through LDS; the weighted sum over units then runs 8 independent loads deep per thread.
__global__ __launch_bounds__(128) void attn_decode_combine_kernel(AttnDecodeArgs a, int nunits) {
  __shared__ float sw[256];
  __shared__ float red[4];
  const int head = blockIdx.x, d = threadIdx.x, lane = d & 63, wave = d >> 6;
  float mu[2], lu[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = d + i * 128;
    mu[i] = -INFINITY;
    lu[i] = 0.f;
    if (u < nunits) {
      const float2 ml = *reinterpret_cast<const float2*>(a.part_ml + ((int64_t)u * a.nh + head) * 2);
      mu[i] = ml.x;
      lu[i] = ml.y;
    }
  }
  float M = wave_max(fmaxf(mu[0], mu[1]));
  if (lane == 0) red[wave] = M;
  __syncthreads();
  M = fmaxf(red[0], red[1]);
  float lsum = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float w = (mu[i] == -INFINITY) ? 0.f : __expf(mu[i] - M);
    sw[d + i * 128] = w;
    lsum += w * lu[i];
  }
  lsum = wave_sum(lsum);
  if (lane == 0) red[2 + wave] = lsum;
  __syncthreads();
  lsum = red[2] + red[3];
  const float* po = a.part_o + (int64_t)head * 128 + d;
  const int64_t ustride = (int64_t)a.nh * 128;
  float acc = 0.f;
  int u = 0;
  for (; u + 8 <= nunits; u += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = po[(u + j) * ustride];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += sw[u + j] * v[j];
  }
  for (; u < nunits; ++u) acc += sw[u] * po[u * ustride];
  ((bf16_t*)a.o)[head * 128 + d] = f2bf(acc / lsum);
}

__global__ __launch_bounds__(256, 2) void attn_decode_fused_kernel(AttnDecodeFusedArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[ATTN_DECODE_FUSED_LDS];
  attn_decode_fused_body(a, smem, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

}  // namespace

void launch_attn_decode_fused(const AttnDecodeFusedArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(attn_decode_fused_kernel, dim3(a.kvh, a.nsplit), dim3(256), 0, st, a);
}

static int g_attn_smx_override = -1;
void set_attn_variant_override(int smx) { g_attn_smx_override = smx; }
static int g_attn_form_override = -1;
void set_attn_form_override(int form) { g_attn_form_override = form; }

// The kernel form launch_attn_prefill picks for these arguments: 0 = this file's 16-rows-per-wave kernel, 64 / 65 = kernels_attn64.hip
static int attn_prefill_form(const AttnPrefillArgs& a, int smx);
static int attn_prefill_smx(const AttnPrefillArgs& a);
static int attn_prefill_smx(const AttnPrefillArgs& a) {
  static const int smx_env = [] { const char* e = getenv("AHA_ATTN_SMX"); return e ? atoi(e) : 3; }();
  union { float f; uint32_t u; } sb;
  sb.f = a.scale;
  const uint32_t scale_bits = sb.u;
  int smx = g_attn_smx_override >= 0 ? g_attn_smx_override : smx_env;
  if (smx == 1 && (scale_bits & 0xffffu) != 0) smx = 0;
  if (smx == 3 && !(a.scale > 0.f)) smx = 0;   // the f32 chain keeps its running maximum in raw score units: needs a positive scale
  return smx;
}
static int attn_prefill_form(const AttnPrefillArgs& a, int smx) {
  // The kernel form.  AHA_ATTN_FORM / set_attn_form_override: 16 = this file's 16-rows-per-wave kernel, 64 = the one-wave-per-SIMD kernel
  // with 64 rows per wave (kernels_attn64.hip) with a tile's parts in program order, 65 = the same software-pipelined inside the wave;
  // unset = automatic: form 65 (f32 chain only) once its 256-row blocks fill the chip.
  static const int form_env = [] { const char* e = getenv("AHA_ATTN_FORM"); return e ? atoi(e) : -1; }();
  const int form = g_attn_form_override >= 0 ? g_attn_form_override : form_env;
  if (smx != 3 || form == 16) return 0;
  const int64_t blocks64 = (a.rows_hint > 0 ? (int64_t)(a.rows_hint + 255) / 256 : (int64_t)((a.S + 255) / 256 + (a.S2 + 255) / 256)) * a.nh;
  // One 4-wave workgroup per CU: a full launch is rounds of 256 workgroups.  Same-box A/B against the 16-row kernel (scripts/attn64_ab.py,
  // 32 heads x head_dim 128, profiles/r06_attn_prefill.md): full attention 2048 / 3072 / 4096 / 8192 rows = 1 / 1.5 / 2 / 4 rounds:
  // 64 vs 75, 165 vs 160, 234 vs 273, 959 vs 1143 us -- ahead wherever the last round is at least ~80 % full; causal (long blocks first, the
  // rounds blur): 62 vs 55 us at 2048 rows, 95 vs 98 at 3072, 139 vs 154 at 4096, 0.51 vs 0.58 ms at 8192, 11.7 vs 13.5 ms at 40 980.
  const int64_t rounds64 = (blocks64 + 255) / 256;
  const bool fills = a.causal ? blocks64 >= 384 : (blocks64 >= 256 && blocks64 * 5 >= rounds64 * 256 * 4);
  const bool auto64 = fills && (a.d == 128 || a.d == 72);
  if (form == 64) return 64;
  if (form == 65 || (form < 0 && auto64)) return 65;
  return 0;
}
bool attn_prefill_takes_qfuse(const AttnPrefillArgs& a) {
  static const bool on = [] { const char* e = getenv("AHA_ATTN_QFUSE"); return e ? atoi(e) != 0 : true; }();
  return on && a.d == 128 && attn_prefill_form(a, attn_prefill_smx(a)) == 0;
}

void launch_attn_prefill(const AttnPrefillArgs& a_in, hipStream_t st) {
  if (a_in.S <= 0) return;
  AttnPrefillArgs a = a_in;
  static const int nw_env = [] {
    const char* e = getenv("AHA_ATTN_WAVES");
    return e ? atoi(e) : 0;
  }();
  constexpr int qt = 1;  // 2 q sub-tiles per wave were measured slower (300 registers: one wave per SIMD)
  // 8 waves (128 q rows) per staged tile once that still fills the chip's 512 block slots (two 8-wave blocks per CU); below
  // that, 64-row blocks balance a causal launch better (S = 1542 x 32 heads: 44 vs 48 us; equal at 2048, 8 waves ahead from there)
  int nwv = (a.d != 64 && (int64_t)((a.S + 127) / 128 + (a.S2 + 127) / 128) * a.nh >= 512) ? 8 : 4;
  if (nw_env == 4 || (nw_env == 8 && a.d != 64)) nwv = nw_env;
  static const int sched_env = [] {
    const char* e = getenv("AHA_ATTN_SCHED");
    return e ? atoi(e) : 1;
  }();
  const bool xcd_order = sched_env && a.kvh % 8 == 0 && a.nh % a.kvh == 0;
  // AHA_ATTN_SEG2=0: a second segment always as its own launch (A/B)
  static const int seg2_env = [] { const char* e = getenv("AHA_ATTN_SEG2"); return e ? atoi(e) : 1; }();
  if (a.S2 > 0 && !(a.causal && xcd_order && seg2_env)) {
    AttnPrefillArgs b = a;
    b.S2 = 0;
    launch_attn_prefill(b, st);
    b.q = (const char*)a.q + (int64_t)a.S * (a.q_ld ? a.q_ld : (int64_t)a.nh * (a.d == 72 ? 96 : a.d)) * 2;
    b.o = (char*)a.o + (int64_t)a.S * a.nh * a.d * 2;
    if (a.q_rope_tab) b.q_rope_tab = (const char*)a.q_rope_tab + (int64_t)a.S * 128 * 2;   // the table rows travel with the q rows
    b.S = a.S2, b.kv_offset = a.kv_offset2, b.kv_total = a.kv_total2;
    launch_attn_prefill(b, st);
    return;
  }
  const int nqb = (a.S + 16 * qt * nwv - 1) / (16 * qt * nwv) + (a.S2 + 16 * qt * nwv - 1) / (16 * qt * nwv);
  a.nqb = xcd_order ? nqb : 0;
  dim3 grid = a.nqb ? dim3(nqb * a.nh) : dim3(nqb, a.nh), block(nwv * 64);
  // AHA_ATTN_SMX: the score chain.
  //   3 (default since round 5) = the f32 score chain: the scores stay the f32 QK^T accumulators through scale, mask, maximum and
  //       exponential; P is rounded to bf16 once for the P.V MFMA (attn_common.h softmax_scores<3>)
  //   0 = the reference's rounding chain bf16(bf16(q.k) * bf16(scale)) (modules.rs:782-783) in the vector ALU, 1 = the same bits with the
  //       scale multiply on the matrix pipe (needs a bf16-exact scale: every model path has one; an op-level caller with another scale
  //       gets 0) -- the bit-faithful forms, kept for A/B and for the parity tables of rounds 1-4
  const int smx = attn_prefill_smx(a);
  // (a caller that handed over raw q heads -- q_norm_w -- asked attn_prefill_takes_qfuse first; should the form have changed since, the
  // 16-row kernel below is the one that norms and rotates)
  const int form_pick = a.q_norm_w != nullptr ? 0 : attn_prefill_form(a, smx);
  if (form_pick != 0 && launch_attn_prefill64(a_in, st, form_pick == 64 ? 0 : 1)) return;
  // row-order epilogue stores (AHA_ATTN_EPI_ROWS=0: from the accumulator fragments): 16-byte pieces need head dims in multiples of 8
  // and 16-byte aligned output rows
  static const int epi_env = [] { const char* e = getenv("AHA_ATTN_EPI_ROWS"); return e ? atoi(e) : 1; }();
  a.epi_rows = epi_env && a.d % 8 == 0 && ((uintptr_t)a.o & 15) == 0;
#define ATTN_LAUNCH(DQK_, DV_)                                                                                                   \
  do {                                                                                                                           \
    if (nwv == 8) {                                                                                                              \
      if (smx == 3) hipLaunchKernelGGL((attn_prefill_kernel<DQK_, DV_, 1, 8, false, 0, 3>), grid, block, lds, st, a, nullptr); \
      else if (smx == 1) hipLaunchKernelGGL((attn_prefill_kernel<DQK_, DV_, 1, 8, false, 0, 1>), grid, block, lds, st, a, nullptr); \
      else hipLaunchKernelGGL((attn_prefill_kernel<DQK_, DV_, 1, 8>), grid, block, lds, st, a, nullptr);                         \
    } else {                                                                                                                     \
      if (smx == 3) hipLaunchKernelGGL((attn_prefill_kernel<DQK_, DV_, 1, 4, false, 0, 3>), grid, block, lds, st, a, nullptr); \
      else if (smx == 1) hipLaunchKernelGGL((attn_prefill_kernel<DQK_, DV_, 1, 4, false, 0, 1>), grid, block, lds, st, a, nullptr); \
      else hipLaunchKernelGGL((attn_prefill_kernel<DQK_, DV_, 1, 4>), grid, block, lds, st, a, nullptr);                         \
    }                                                                                                                            \
  } while (0)
  if (a.d == 128) {
    const size_t lds = 2 * KV_PAGE_TOKENS * 2 * (128 + 128);
#ifdef AHA_DEBUG_KERNELS   // per-part cycle trace and the ablation instantiations (results wrong by construction): debug builds only
    static const bool ptrace = [] { const char* e = getenv("AHA_ATTN_PTRACE"); return e && atoi(e) != 0; }();
    if (ptrace && nwv == 8) {   // debug: per-part cycle sums of two waves of the middle block
      static unsigned long long* d_tr = nullptr;
      if (!d_tr) (void)hipMalloc((void**)&d_tr, 16 * 8);
      (void)hipMemsetAsync(d_tr, 0, 16 * 8, st);
      hipLaunchKernelGGL((attn_prefill_kernel<128, 128, 1, 8, true>), grid, block, lds, st, a, d_tr);
      unsigned long long h[16];
      (void)hipMemcpyAsync(h, d_tr, sizeof(h), hipMemcpyDeviceToHost, st);
      (void)hipStreamSynchronize(st);
      static const char* names[7] = {"top", "QK^T", "scores", "probs", "P.V", "stage", "barriers"};
      for (int w = 0; w < 2; ++w) {
        fprintf(stderr, "[attn ptrace] wave %d, %llu tiles, cycles per tile:", w * 4, h[w * 8 + 7]);
        for (int k = 0; k < 7; ++k) fprintf(stderr, " %s %.0f", names[k], h[w * 8 + 7] ? (double)h[w * 8 + k] / (double)h[w * 8 + 7] : 0.0);
        fprintf(stderr, "\n");
      }
      return;
    }
    static const int abl = [] { const char* e = getenv("AHA_ATTN_ABL"); return e ? atoi(e) : 0; }();
    if (abl && nwv == 8) {
      if (abl == 2) hipLaunchKernelGGL((attn_prefill_kernel<128, 128, 1, 8, false, 2>), grid, block, lds, st, a, nullptr);
      if (abl == 3) hipLaunchKernelGGL((attn_prefill_kernel<128, 128, 1, 8, false, 3>), grid, block, lds, st, a, nullptr);
      if (abl == 4) hipLaunchKernelGGL((attn_prefill_kernel<128, 128, 1, 8, false, 4>), grid, block, lds, st, a, nullptr);
      return;
    }
#endif
    ATTN_LAUNCH(128, 128);
  } else if (a.d == 64) {  // Qwen3-ASR audio encoder
    const size_t lds = 2 * KV_PAGE_TOKENS * 2 * (64 + 64);
    if (smx == 3) hipLaunchKernelGGL((attn_prefill_kernel<64, 64, 1, 4, false, 0, 3>), grid, block, lds, st, a, nullptr);
    else if (smx == 1) hipLaunchKernelGGL((attn_prefill_kernel<64, 64, 1, 4, false, 0, 1>), grid, block, lds, st, a, nullptr);
    else hipLaunchKernelGGL((attn_prefill_kernel<64, 64, 1, 4>), grid, block, lds, st, a, nullptr);
  } else {  // head_dim 72 (Qwen3-VL ViT): Q/K rows padded to 96, V block to 80
    const size_t lds = 2 * KV_PAGE_TOKENS * 2 * (96 + 80);
    ATTN_LAUNCH(96, 80);
  }
#undef ATTN_LAUNCH
}

void launch_attn_decode(const AttnDecodeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(attn_decode_kernel, dim3(a.kvh, a.nsplit), dim3(256), 0, st, a);
  hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(a.nh), dim3(128), 0, st, a, a.nsplit * 4);
}

}  // namespace aha
