// MFMA fragment helpers and the online-softmax tile update shared by the attention kernels (kernels_attn.hip) and the
// persistent decode-step kernel (decode_mega.hip).  Fragment scheme: see the header comment of kernels_attn.hip.
#pragma once
#include "common.h"
#include "kernels.h"

namespace aha {


__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8_t as_frag(u32x4_t v) {
  union { u32x4_t u; bf16x8_t b; } x;
  x.u = v;
  return x.b;
}
// v_mfma_f32_16x16x16_bf16 with A = diag(v): lane (G, c) holds A[row c][k = 4G .. 4G+3] and B[k = 4G .. 4G+3][col c], and receives
// D[row 4G + reg][col c] -- so D = diag(v) . B hands every lane back its OWN four B values times v.  Exact: each output is ONE
// bf16 x bf16 product (exact in f32) plus zeros.  What it is for: the reference's `bf16(scores) * bf16(scale)` (modules.rs:782-783)
// costs the VALU-bound prefill softmax an unpack (shift / and) and a multiply per score when done in the vector ALU; here the packed
// bf16 pair of the first rounding goes straight into the matrix pipe (which has slack) and comes back as the f32 product.
typedef short __attribute__((ext_vector_type(4))) s16x4_t;
__device__ __forceinline__ s16x4_t diag_frag(float v, int G, int c) {
  const short b = (short)(__float_as_uint(v) >> 16);   // v is bf16-exact (checked on the host)
  const int j = c - 4 * G;
  s16x4_t a;
  a[0] = j == 0 ? b : (short)0;
  a[1] = j == 1 ? b : (short)0;
  a[2] = j == 2 ? b : (short)0;
  a[3] = j == 3 ? b : (short)0;
  return a;
}
__device__ __forceinline__ f32x4_t mfma_diag(s16x4_t dg, uint32_t lo, uint32_t hi) {
  union { uint32_t u[2]; s16x4_t s; } x;
  x.u[0] = lo; x.u[1] = hi;
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dg, x.s, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}
__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float group_max(float v) {  // across the 4 lane groups (same column c)
  // v_permlane32_swap / v_permlane16_swap instead of __shfl_xor (ds_bpermute: an LDS round trip in the middle of the softmax's
  // dependency chain, once per tile); max is exact in any order
  {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// Online-softmax update for one 64-token tile.  st[sub][reg] holds raw S for token sub*16+G*4+reg, column c.
// valid(tok_in_tile) masks both causality and the tail of the last page.  Returns the two P^T fragments.
// Two halves (the prefill kernel starts the V^T fragment reads between them):
//   softmax_scores: the reference's rounding chain on the scores, the mask, the tile maximum, the new running maximum and the
//                   rescale factor alpha of everything accumulated so far; st is left holding the rounded, masked scores;
//   softmax_probs:  p = e^(s - m) (one fma + one v_exp_f32 per score), the row sum, the bf16 P^T fragments.
// SMX (prefill kernel): 0 = the whole rounding chain in the vector ALU; 1 = `* scaling` through the matrix pipe (mfma_diag with
// dg = diag(scale)): 24 of the ~120 vector instructions per 64-token tile become 4 MFMAs -- cfg 3 ViT kernel 152 -> 140 us, text
// 8192 causal 0.685 -> 0.65 ms (profiles/r04_attn_prefill.md; widening the SECOND rounding through diag(1) as well: ViT 144 us)
template <int SMX = 0, typename ValidFn>
__device__ __forceinline__ void softmax_scores(f32x4_t (&st)[4], float scale, ValidFn valid, int G, float& m, float& alpha, float& m2,
                                               s16x4_t dg = s16x4_t{0, 0, 0, 0}) {
  float tmax = -INFINITY;
  constexpr float LOG2E = 1.4426950408889634f;
  if constexpr (SMX == 3) {
    // f32 score chain (round 5): the scores stay the f32 QK^T accumulators -- no bf16(q.k), no bf16(. * scale) -- through mask, maximum
    // and exponential (the reference's eager path rounds twice, modules.rs:782-783: the oracle's `attn_scores_rounded` switch; since
    // round 6 the decode kernels keep f32 scores too, softmax_tile below, so a position sees ONE convention whether it arrives by prefill
    // or by decode).  The running maximum is kept in RAW score units (scale > 0) and the scale rides the exponent's
    // fma: p = exp2(s * (scale log2 e) - m * (scale log2 e)).  ~40 of the ~92 vector instructions per tile go away.
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!valid(sub * 16 + G * 4 + r)) st[sub][r] = -INFINITY;
    // v_max3_f32 by hand: fmaxf on an MFMA output makes clang canonicalise it first (v_max_f32 x, x)
    auto sv = [&](int i) { return st[i >> 2][i & 3]; };
    tmax = max3(sv(0), sv(1), sv(2));
#pragma unroll
    for (int i = 3; i < 15; i += 2) tmax = max3(tmax, sv(i), sv(i + 1));
    tmax = max3(tmax, sv(15), sv(15));
    tmax = group_max(tmax);
    const float m_new = fmaxf(m, tmax);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float c2 = scale * LOG2E;
    alpha = __builtin_amdgcn_exp2f((m - m_use) * c2);         // m = -inf -> 0
    m2 = m_use * c2;
    m = m_new;
    return;
  }
  if constexpr (SMX >= 1) {
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)   // matmul output -> bf16 (v_cvt_pk), x bf16(scaling) exactly in f32
      st[sub] = mfma_diag(dg, pack_bf(st[sub][0], st[sub][1]), pack_bf(st[sub][2], st[sub][3]));
  }
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // matmul output -> bf16, then `* scaling` -> bf16 (modules.rs:782-783)
      float s = SMX == 1 ? rbf(st[sub][r]) : rbf(rbf(st[sub][r]) * scale);
      if (!valid(sub * 16 + G * 4 + r)) s = -INFINITY;
      st[sub][r] = s;
      tmax = fmaxf(tmax, s);
    }
  tmax = group_max(tmax);
  const float m_new = fmaxf(m, tmax);
  const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far: keep everything at zero
  alpha = __expf(m - m_use);                               // m = -inf -> 0
  m2 = m_use * LOG2E;
  m = m_new;
}
// k2s = the factor on the score inside the exponent: log2 e for the rounded chains (the scale is already in the scores), scale * log2 e
// for the f32 chain (SMX 3)
// SUM = false: no row sum here (the ViT instantiation of the f32 chain takes it from the matrix pipe: a ones row in the V^T pad,
// kernels_vit.hip vit_rope_pack_kernel) -- 8 packed adds and the l update less per tile
template <bool SUM = true>
__device__ __forceinline__ void softmax_probs(const f32x4_t (&st)[4], float m2, float alpha, float& l, bf16x8_t (&pf)[2],
                                              float k2s = 1.4426950408889634f) {
  // two scores per instruction where the ISA has a packed f32 form: e^(s - m) = exp2(fma(s, log2 e, -m log2 e)) as v_pk_fma_f32,
  // the row sum as v_pk_add_f32 on two running halves (8 + 8 instead of 16 + 16 issue slots of a VALU-bound loop)
  const f32x2_t k2 = {k2s, k2s}, nm2 = {-m2, -m2};
  f32x2_t psum2 = {0.f, 0.f};
  uint32_t pk[2][4];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    const f32x2_t s01 = {st[sub][0], st[sub][1]}, s23 = {st[sub][2], st[sub][3]};
    const f32x2_t a01 = __builtin_elementwise_fma(s01, k2, nm2), a23 = __builtin_elementwise_fma(s23, k2, nm2);
    const f32x2_t p01 = {__builtin_amdgcn_exp2f(a01[0]), __builtin_amdgcn_exp2f(a01[1])};
    const f32x2_t p23 = {__builtin_amdgcn_exp2f(a23[0]), __builtin_amdgcn_exp2f(a23[1])};
    if (SUM) {
      psum2 += p01;
      psum2 += p23;
    }
    pk[sub >> 1][(sub & 1) * 2 + 0] = pack_bf(p01[0], p01[1]);
    pk[sub >> 1][(sub & 1) * 2 + 1] = pack_bf(p23[0], p23[1]);
  }
  if (SUM) l = l * alpha + (psum2[0] + psum2[1]);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    u32x4_t u = {pk[kk][0], pk[kk][1], pk[kk][2], pk[kk][3]};
    pf[kk] = as_frag(u);
  }
}
// The decode kernels' tile update (one query row per step).  Round 6 (round-5 advisor, medium): the scores stay f32 here too -- s = (q . k) *
// scale in f32, no bf16(q.k), no bf16(. * scale) -- the prefill kernels' default convention (SMX 3, kernels_attn64.hip), where through
// round 5 decode kept the eager path's two roundings (modules.rs:782-783) and the same cache position was scored under two conventions
// depending on how it had arrived.  m stays in SCALED score units (natural-log domain): the split merges (attn_decode_body.h,
// attn_decode_combine_kernel) are unchanged.  AHA_ATTN_DECODE_ROUNDED (compile-time, debug builds) restores the rounded chain.
template <typename ValidFn>
__device__ __forceinline__ void softmax_tile(f32x4_t (&st)[4], float scale, ValidFn valid, int G, float& m, float& l,
                                             float& alpha, bf16x8_t (&pf)[2]) {
  float m2;
#ifdef AHA_ATTN_DECODE_ROUNDED
  softmax_scores(st, scale, valid, G, m, alpha, m2);
#else
  constexpr float LOG2E = 1.4426950408889634f;
  float tmax = -INFINITY;
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = st[sub][r] * scale;
      if (!valid(sub * 16 + G * 4 + r)) s = -INFINITY;
      st[sub][r] = s;
      tmax = fmaxf(tmax, s);
    }
  tmax = group_max(tmax);
  const float m_new = fmaxf(m, tmax);
  const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far: keep everything at zero
  alpha = __expf(m - m_use);                               // m = -inf -> 0
  m2 = m_use * LOG2E;
  m = m_new;
#endif
  softmax_probs(st, m2, alpha, l, pf);
}

}  // namespace aha
