// Shared device code of the 256-row MFMA GEMM kernels (kernels_gemm.hip: gemm256q_kernel / gemm256p_kernel / gemm_glds_kernel;
// kernels_gemm_sk.hip: the persistent, segment-table-driven gemm256s_kernel): tile constants, the XCD-aware tile order, the epilogue
// chains (rounding points = the reference op boundaries) and the main loop of the four-wave 256 x 256 (x 192) tile.  Everything is
// __device__ __forceinline__ in an anonymous namespace: each translation unit gets its own copy and inlines it.
#pragma once
#include <math.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace aha {

namespace {

extern __shared__ __attribute__((aligned(16))) char gemm_smem[];   // the block's dynamic LDS (every GEMM kernel of the unit)

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand per stage

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8_t as_frag(u32x4_t v) {
  union { u32x4_t u; bf16x8_t b; } x;
  x.u = v;
  return x.b;
}
// byte offset of 16-byte slot `slot` (0..7) of row `row` inside a [128][64] bf16 tile
__device__ __forceinline__ int swz(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float gelu_tanh_f(float x) {
  // candle Tensor::gelu / Activation::GeluPytorchTanh: 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), written as
  // x / (1 + e^(-2u)) on the hardware exp2 / rcp (1 ulp each): 8 VALU operations instead of tanhf's ~40.  The ViT fc1 epilogue
  // applies it to 17.6 M elements per block -- with tanhf that was ~25 us of VALU time per launch, as much as the MFMA time.
  // The result is rounded to bf16 by the caller; against the tanhf form it differs in < 0.1 % of the elements, by one bf16 ulp.
  const float k2 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
  const float u = x * fmaf(0.044715f * x, x, 1.0f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(k2 * u));
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
// candle silu: x / (1 + e^-x), written as x * rcp(1 + exp2(-x log2 e)) on the hardware exp2 / rcp (1 ulp each): 6 vector operations where
// expf + an IEEE division are ~26 (range scaling, div_scale / div_fmas / div_fixup) -- the gate+up epilogue applies it to 96 pairs per
// lane and tile, ~6 us of a 256 x 192 tile's time with nothing to overlap it (round 6; like gelu_tanh_f above).  The result is rounded to
// bf16 by the caller; against the exact form it differs in < 0.01 % of the elements, by one bf16 ulp.  -DAHA_SILU_EXACT: the old form (A/B).
#ifdef AHA_SILU_EXACT
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
#else
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
#endif

// XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give each XCD a contiguous run of tiles
// (which share A row panels / W column panels in its private L2).  Bijective for any grid size.
template <int TBM = BM, int TBN = BN>
__device__ __forceinline__ void tile_of_block(const GemmArgs& a, int& m0, int& n0) {
  const int ntm = (a.M + TBM - 1) / TBM, ntn = (a.N + TBN - 1) / TBN, nwg = ntm * ntn;
  int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // Within an XCD's run of tiles, the ~32 tiles in flight on the XCD's 32 CUs should form a compact rectangle: per K step
  // they then pull (rows + columns) operand panels into the XCD's L2 instead of one panel per tile -- an 8 x 4 rectangle
  // fetches 12 panels for 32 tiles (81 % L2 hits), a 32 x 1 strip 33 (50 %: at 8192^3 that is 4.3 GB per GEMM over the
  // fabric, which bounds the kernel).  Grouped order: bands of 8 tiles of the SHORTER dimension, the band's tiles varying
  // fastest.  (PMC FETCH_SIZE of the cfg 3 gate/up GEMM, 7 x 96 tiles, was 1.47 GB per launch in plain row-major order.)
  const int GRP = a.tile_group > 0 ? a.tile_group : (1 << 20);
  if (ntm <= ntn) {
    const int per = GRP * ntn, grp = bid / per, first = grp * GRP, gsz = min(ntm - first, GRP), r = bid - grp * per;
    m0 = (first + r % gsz) * TBM;
    n0 = (r / gsz) * TBN;
  } else {
    const int per = GRP * ntm, grp = bid / per, first = grp * GRP, gsz = min(ntn - first, GRP), r = bid - grp * per;
    n0 = (first + r % gsz) * TBN;
    m0 = (r / gsz) * TBM;
  }
}

// one K tile of MFMA work from an LDS stage: acc[ni][mi] += W-frag(ni) x A-frag(mi)
__device__ __forceinline__ void mma_tile(const char* sa, const char* sw, int wm, int wn, int G, int c, f32x4_t (&acc)[4][4]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8_t af[4], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i] = as_frag(*reinterpret_cast<const u32x4_t*>(sa + swz(wm * 64 + i * 16 + c, ks * 4 + G)));
      wf[i] = as_frag(*reinterpret_cast<const u32x4_t*>(sw + swz(wn * 64 + i * 16 + c, ks * 4 + G)));
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16(wf[ni], af[mi], acc[ni][mi]);
  }
}

// ---- epilogue --------------------------------------------------------------------------------------------------------------
// Every MFMA variant here is issued as W-fragment x A-fragment, so a lane ends up with groups of 4 consecutive output columns
// n..n+3 of one row m: bias / activation / gate*up pairing / residual are lane-local and the store is 8 bytes.
// One group of 4 raw f32 sums -> the reference's rounding chain -> store.
template <int ACT, bool HAS_BIAS, bool HAS_RES>
__device__ __forceinline__ void epi_group(const GemmArgs& a, int m, int n, float x0, float x1, float x2, float x3) {
  if (n >= a.N) return;
  if (ACT == ACT_PARTIAL_F32) {
    *reinterpret_cast<float4*>((float*)a.C + (int64_t)m * a.ldc + n) = make_float4(x0, x1, x2, x3);
    return;
  }
  float v[4] = {rbf(x0), rbf(x1), rbf(x2), rbf(x3)};
  if (HAS_BIAS) {
    const uint2 b2 = *reinterpret_cast<const uint2*>((const bf16_t*)a.bias + n);
    v[0] = rbf(v[0] + lo_bf(b2.x)); v[1] = rbf(v[1] + hi_bf(b2.x));
    v[2] = rbf(v[2] + lo_bf(b2.y)); v[3] = rbf(v[3] + hi_bf(b2.y));
  }
  if (ACT == ACT_GELU_TANH) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rbf(gelu_tanh_f(v[r]));
  } else if (ACT == ACT_GELU_ERF) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rbf(gelu_erf_f(v[r]));
  } else if (ACT == ACT_SILU) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = rbf(silu_f(v[r]));
  }
  if (HAS_RES) {
    const uint2 r2 = *reinterpret_cast<const uint2*>((const bf16_t*)a.residual + (int64_t)m * a.ldc + n);
    v[0] += lo_bf(r2.x); v[1] += hi_bf(r2.x); v[2] += lo_bf(r2.y); v[3] += hi_bf(r2.y);
  }
  uint2 w2;
  w2.x = pack_bf(v[0], v[1]);
  w2.y = pack_bf(v[2], v[3]);
  *reinterpret_cast<uint2*>((bf16_t*)a.C + (int64_t)m * a.ldc + n) = w2;
}
// ACT_SILU_MUL_PAIRS: W rows come in 16-row blocks, gate rows j..j+15 then up rows j..j+15 (the model loader interleaves them);
// n = fused-weight row of the 4 gate values, oc = their output column
__device__ __forceinline__ void epi_pairs(const GemmArgs& a, int m, int n, int oc, const float (&gt)[4], const float (&up)[4]) {
  if (n >= a.N) return;
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gte = rbf(silu_f(rbf(gt[r])));  // gate_proj -> bf16, act_fn -> bf16 (modules.rs:82)
    const float u = rbf(up[r]);                 // up_proj -> bf16 (modules.rs:83)
    v[r] = gte * u;                             // lhs * rhs -> bf16 (modules.rs:84)
  }
  uint2 w2;
  w2.x = pack_bf(v[0], v[1]);
  w2.y = pack_bf(v[2], v[3]);
  *reinterpret_cast<uint2*>((bf16_t*)a.C + (int64_t)m * a.ldc + oc) = w2;
}

// 16x16x32 fragments: lane holds C[m = .. + c][n = .. + G*4 + 0..3].  mb / nb: first row / column of the wave's sub-tile
// (MI x 4 fragments of 16 x 16)
template <int ACT, bool HAS_BIAS, bool HAS_RES, int MI, int NI = 4>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x4_t (&acc)[NI][MI], int mb, int nb, int G, int c) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mb + mi * 16 + c;
    if (m >= a.M) continue;
    if (ACT == ACT_SILU_MUL_PAIRS) {
#pragma unroll
      for (int np = 0; np < NI / 2; ++np) {
        const float gt[4] = {acc[2 * np][mi][0], acc[2 * np][mi][1], acc[2 * np][mi][2], acc[2 * np][mi][3]};
        const float up[4] = {acc[2 * np + 1][mi][0], acc[2 * np + 1][mi][1], acc[2 * np + 1][mi][2], acc[2 * np + 1][mi][3]};
        epi_pairs(a, m, nb + np * 32 + G * 4, nb / 2 + np * 16 + G * 4, gt, up);
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        epi_group<ACT, HAS_BIAS, HAS_RES>(a, m, nb + ni * 16 + G * 4, acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
    }
  }
}

// The same epilogue for the 128^2 kernel in ROW order (round 4): a wave passes each 32-row band of its 64 x 64 sub-tile through
// 4.5 KiB of LDS after the first rounding (Linear output -> bf16) and runs the rest of the chain on 8 consecutive columns per lane:
// 16-byte bias / residual loads and stores, eight whole 128-byte row segments per instruction instead of 8-byte pieces on 16 rows
// (the ViT proj GEMM -- bias + residual, 288 blocks reaching their epilogues together -- spent a third of its time there).
// Not for ACT_SILU_MUL_PAIRS / ACT_PARTIAL_F32 (they keep the fragment-order epilogue above).  wbuf: this wave's 32 x 144 B.
// NI = 16-column fragments of the wave's sub-tile (4: 64 columns, 8 lanes per row, 8 rows per pass; 2: 32 columns, 4 lanes per row, 16 rows)
template <int ACT, bool HAS_BIAS, bool HAS_RES, int NI = 4>
__device__ __forceinline__ void epilogue16_rows(const GemmArgs& a, f32x4_t (&acc)[NI][4], int mb, int nb, int lane, char* wbuf) {
  static_assert(ACT != ACT_SILU_MUL_PAIRS && ACT != ACT_PARTIAL_F32, "fragment-order epilogue for these");
  constexpr int PITCH = NI * 32 + 16, LPR = NI * 2, RPI = 64 / LPR;
  const int G = lane >> 4, c = lane & 15;
  const int rr = lane / LPR, cc = lane % LPR;
  const int n = nb + cc * 8;
#pragma unroll
  for (int band = 0; band < 2; ++band) {
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const f32x4_t& v = acc[ni][band * 2 + mh];
        uint2 w2;
        w2.x = pack_bf(v[0], v[1]);   // Linear output -> bf16
        w2.y = pack_bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(wbuf + (mh * 16 + c) * PITCH + (ni * 16 + G * 4) * 2) = w2;
      }
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int row = it * RPI + rr, m = mb + band * 32 + row;
      const uint4 d4 = *reinterpret_cast<const uint4*>(wbuf + row * PITCH + cc * 16);
      uint32_t d[4] = {d4.x, d4.y, d4.z, d4.w};
      if (m < a.M && n < a.N) {
        if (HAS_BIAS) {
          const uint4 b4 = *reinterpret_cast<const uint4*>((const bf16_t*)a.bias + n);
          const uint32_t b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = pack_bf(lo_bf(d[k]) + lo_bf(b[k]), hi_bf(d[k]) + hi_bf(b[k]));
        }
        if (ACT == ACT_GELU_TANH) {
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = pack_bf(gelu_tanh_f(lo_bf(d[k])), gelu_tanh_f(hi_bf(d[k])));
        } else if (ACT == ACT_GELU_ERF) {
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = pack_bf(gelu_erf_f(lo_bf(d[k])), gelu_erf_f(hi_bf(d[k])));
        } else if (ACT == ACT_SILU) {
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = pack_bf(silu_f(lo_bf(d[k])), silu_f(hi_bf(d[k])));
        }
        if (HAS_RES) {
          const uint4 r4 = *reinterpret_cast<const uint4*>((const bf16_t*)a.residual + (int64_t)m * a.ldc + n);
          const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = pack_bf(lo_bf(d[k]) + lo_bf(r[k]), hi_bf(d[k]) + hi_bf(r[k]));
        }
        *reinterpret_cast<uint4*>((bf16_t*)a.C + (int64_t)m * a.ldc + n) = make_uint4(d[0], d[1], d[2], d[3]);
      }
    }
  }
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 32x32x16 fragments (W-fragment x A-fragment): lane l holds row m = .. + (l & 31) and, in register quad q, the columns
// n = .. + 8 q + 4 (l >> 5) + 0..3.  acc[nf][mf]: NF x MF fragments of 32 x 32.
template <int ACT, bool HAS_BIAS, bool HAS_RES, int NF, int MF, int NFV = NF>
__device__ __forceinline__ void epilogue32(const GemmArgs& a, f32x16_t (&acc)[NF][MF], int mb, int nb, int lane) {
  const int r32 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = mb + mf * 32 + r32;
    if (m >= a.M) continue;
#pragma unroll
    for (int nf = 0; nf < NFV; ++nf) {   // (NFV < NF: the 192-column tile keeps three of a wave's four fragment columns)
      const f32x16_t& v = acc[nf][mf];
      if (ACT == ACT_SILU_MUL_PAIRS) {   // a 32-row W fragment = one gate block (quads 0,1) + its up block (quads 2,3)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float gt[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
          const float up[4] = {v[4 * q + 8], v[4 * q + 9], v[4 * q + 10], v[4 * q + 11]};
          epi_pairs(a, m, nb + nf * 32 + 8 * q + 4 * h, (nb + nf * 32) / 2 + 8 * q + 4 * h, gt, up);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          epi_group<ACT, HAS_BIAS, HAS_RES>(a, m, nb + nf * 32 + 8 * q + 4 * h, v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      }
    }
  }
}

// The same epilogue for the four-wave 256^2 kernel (4 x 4 fragments per wave), with the stores in ROW order.  epilogue32's lanes
// own 4 columns of 32 different rows, so each of its 8-byte stores (and residual loads) touches 32 rows x 16 B: with all CUs in
// their epilogues at once the L2 takes ~8k partial-line requests per block and the epilogue costs ~11 us per 256^2 tile
// (in-kernel timestamps, profiles/r02_gemm_anatomy.md).  Here a wave passes each 32-row band of its sub-tile through 8 KiB of LDS
// (the first rounding -- Linear output -> bf16 -- happens before, so the band is bf16 and the chain's values are unchanged):
// lanes write their 4-column groups (pitch 264 B: conflict-free), then read 16 B of ONE row each, 16 lanes per row, run the rest
// of the chain on 8 consecutive columns (16-byte bias / residual loads) and store 16 B: 4 whole 256-byte rows per instruction.
// ACT_SILU_MUL_PAIRS: the band is the finished 64-column output (8 lanes per row).  wbuf: this wave's 32 x 264 B of LDS, free
// once every wave of the block has left the k loop.
// WC: fused-weight columns of the wave's sub-tile that hold results (128, or 96 for the 192-column tile: the row pass keeps its
// 16 / 8 lanes per row and masks the lanes past the last valid column).
// One 32-row band of a wave's sub-tile: `f` = its four (three) fragments [n fragment of 32], rows mband + 0..31.
template <int ACT, bool HAS_BIAS, bool HAS_RES, int WC = 128>
__device__ __forceinline__ void epilogue32_band(const GemmArgs& a, const f32x16_t& f0, const f32x16_t& f1, const f32x16_t& f2, const f32x16_t& f3,
                                                int mband, int nb, int lane, char* wbuf) {
  constexpr bool PAIRS = ACT == ACT_SILU_MUL_PAIRS;
  constexpr int COLS = PAIRS ? 64 : 128, PITCH = COLS * 2 + 8, LPR = COLS / 8, RPI = 64 / LPR, NIT = 32 / RPI;
  constexpr int NFV = WC / 32, VCOLS = PAIRS ? WC / 2 : WC;   // valid fragment columns / valid output columns of the band
  static_assert(ACT != ACT_PARTIAL_F32, "f32 partial sums keep the fragment-order epilogue");
  const int r32 = lane & 31, h = lane >> 5;
  const int rr = lane / LPR, cc = lane % LPR;
  const int n = (PAIRS ? nb / 2 : nb) + cc * 8;   // this lane's 8 output columns in the row pass
  const int ncols = PAIRS ? a.N / 2 : a.N;
  {
#pragma unroll
    for (int nf0 = 0; nf0 < 4; nf0 += 2) {
      f32x16_t sum[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[j][r] = (nf0 + j == 0 ? f0 : nf0 + j == 1 ? f1 : nf0 + j == 2 ? f2 : f3)[r];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (nf0 + j >= NFV) continue;
        if (PAIRS) {   // a 32-row W fragment = one gate block (quads 0,1) + its up block (quads 2,3) -> 16 output columns
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gte = rbf(silu_f(rbf(sum[j][4 * q + r])));  // gate_proj -> bf16, act_fn -> bf16 (modules.rs:82)
              const float u = rbf(sum[j][4 * q + 8 + r]);             // up_proj -> bf16 (modules.rs:83)
              v[r] = gte * u;                                         // lhs * rhs -> bf16 (modules.rs:84)
            }
            uint2 w2;
            w2.x = pack_bf(v[0], v[1]);
            w2.y = pack_bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(wbuf + r32 * PITCH + ((nf0 + j) * 16 + 8 * q + 4 * h) * 2) = w2;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint2 w2;
            w2.x = pack_bf(sum[j][4 * q], sum[j][4 * q + 1]);   // Linear output -> bf16
            w2.y = pack_bf(sum[j][4 * q + 2], sum[j][4 * q + 3]);
            *reinterpret_cast<uint2*>(wbuf + r32 * PITCH + ((nf0 + j) * 32 + 8 * q + 4 * h) * 2) = w2;
          }
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + rr, m = mband + row;
      const uint2 lo = *reinterpret_cast<const uint2*>(wbuf + row * PITCH + cc * 16);
      const uint2 hi = *reinterpret_cast<const uint2*>(wbuf + row * PITCH + cc * 16 + 8);
      uint32_t d[4] = {lo.x, lo.y, hi.x, hi.y};
      if (m < a.M && n < ncols && cc * 8 < VCOLS) {
        if (!PAIRS) {
          if (HAS_BIAS) {
            const uint4 b4 = *reinterpret_cast<const uint4*>((const bf16_t*)a.bias + n);
            const uint32_t b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = pack_bf(lo_bf(d[k]) + lo_bf(b[k]), hi_bf(d[k]) + hi_bf(b[k]));
          }
          if (ACT == ACT_GELU_TANH) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = pack_bf(gelu_tanh_f(lo_bf(d[k])), gelu_tanh_f(hi_bf(d[k])));
          } else if (ACT == ACT_GELU_ERF) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = pack_bf(gelu_erf_f(lo_bf(d[k])), gelu_erf_f(hi_bf(d[k])));
          } else if (ACT == ACT_SILU) {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = pack_bf(silu_f(lo_bf(d[k])), silu_f(hi_bf(d[k])));
          }
          if (HAS_RES) {
            const uint4 r4 = *reinterpret_cast<const uint4*>((const bf16_t*)a.residual + (int64_t)m * a.ldc + n);
            const uint32_t r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = pack_bf(lo_bf(d[k]) + lo_bf(r[k]), hi_bf(d[k]) + hi_bf(r[k]));
          }
        }
        *reinterpret_cast<uint4*>((bf16_t*)a.C + (int64_t)m * a.ldc + n) = make_uint4(d[0], d[1], d[2], d[3]);
      }
    }
  }
}
// The f32 partial sums of a K slice (ACT_PARTIAL_F32: split-K slabs, tensor-parallel row-split projections) in ROW order as well
// (round 4): the fragment-order epilogue32 writes 16 B per lane on 32 different rows -- 32-byte runs; here a wave passes each 32-row band
// through 16.5 KiB of LDS (pitch 528 B: conflict-free for the 16-byte fragment writes) and stores 512-byte row segments, two whole
// rows per instruction.  wbuf: this wave's 32 x 528 B (32 x 272 B for the 8-wave kernel's 64-column sub-tile: NF = 2).
template <int WC = 128, int NF = 4>
__device__ __forceinline__ void epilogue32_rows_f32(const GemmArgs& a, f32x16_t (&acc)[NF][4], int mb, int nb, int lane, char* wbuf) {
  constexpr int COLS = NF * 32, PITCH = COLS * 4 + 16, NFV = WC / 32, LPR = COLS / 4, RPI = 64 / LPR, NIT = 32 / RPI;   // NF = 2: the 8-wave kernel's 128 x 64 sub-tile
  const int r32 = lane & 31, h = lane >> 5;
  const int rr = lane / LPR, cc = lane % LPR;        // row pass: LPR lanes x 16 B per row, RPI rows per instruction
  const int n = nb + cc * 4;
  float* C = (float*)a.C;
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
#pragma unroll
    for (int nf = 0; nf < NFV; ++nf)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4_t v = {acc[nf][mf][4 * q], acc[nf][mf][4 * q + 1], acc[nf][mf][4 * q + 2], acc[nf][mf][4 * q + 3]};
        *reinterpret_cast<f32x4_t*>(wbuf + r32 * PITCH + (nf * 32 + 8 * q + 4 * h) * 4) = v;
      }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = it * RPI + rr, m = mb + mf * 32 + row;
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(wbuf + row * PITCH + cc * 16);
      if (m < a.M && n < a.N && cc * 4 < WC) *reinterpret_cast<f32x4_t*>(C + (int64_t)m * a.ldc + n) = v;
    }
  }
}
template <int ACT, bool HAS_BIAS, bool HAS_RES, int WC = 128>
__device__ __forceinline__ void epilogue32_rows(const GemmArgs& a, f32x16_t (&acc)[4][4], int mb, int nb, int lane, char* wbuf) {
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
    epilogue32_band<ACT, HAS_BIAS, HAS_RES, WC>(a, acc[0][mf], acc[1][mf], acc[2][mf], acc[3][mf], mb + mf * 32, nb, lane, wbuf);
}

constexpr int BM2 = 256, BN2 = 256;
constexpr int TILE2_BYTES = 256 * BK * 2;  // 32 KiB per operand per stage

// ---- variant 5: the 256 x 256 x 64 tile on FOUR waves (2 x 2, 128 x 128 each), one wave per SIMD ---------------------------------
// Each wave owns 128 x 128 of the output as 4 x 4 fragments of v_mfma_f32_32x32x16_bf16: 256 accumulator registers (the
// accumulation half of the 512-register file) and four register sets of operand fragments (A rows i*64.., W rows j*64.. of the
// wave's slice, 8 fragments = 32 registers each).  A fragment read from LDS feeds FOUR MFMAs (two in the 8-wave kernels): 128 KiB
// of fragment reads per K tile and CU instead of 192, half the barriers, no second wave per SIMD to arbitrate with.
// A K tile is four phases of 16 MFMAs (one 64 x 64 quadrant over the 64-deep K tile); the quadrant order (0,0) (0,1) (1,1) (1,0)
// changes ONE operand set per phase, and that set is read from LDS during the phase before, k-step by k-step, in the shadow of
// the MFMAs:
//     P1: MFMA A0 W0 | read W1(t)   | stage W0(t+2)          P3: MFMA A1 W1 | read A0(t+1) | stage A1(t+2)
//     P2: MFMA A0 W1 | read A1(t)   | stage W1(t+2)          P4: MFMA A1 W0 | read W0(t+1) | stage A0(t+3)
// W0(t+1) goes into the register set W1(t) has left (the W sets swap roles every tile: the loop body is two tiles).
// LDS: two stages of four 16-KiB regions (A half 0 / 1, W half 0 / 1: the 64-row halves of BOTH waves that share the operand).
// Because a region is copied into registers once per tile, it is free again one phase after it was read and is restaged right
// away with the tile two further on: every LDS-DMA (buffer_load_dwordx4 ... lds: scalar base + tile offset, one address
// register per piece, no address arithmetic in the loop) is issued SEVEN phases (~3500 cycles) before its data is read.
// Synchronisation: one s_waitcnt vmcnt(24) lgkmcnt(0) + one s_barrier per phase.  At the barrier of phase p every wave's DMA
// for the region read in p has landed (it is the 7th-newest group of 4: 6 x 4 newer ones may stay in flight) and every wave's
// reads of phase p-1 have returned, so the region they came from is restaged in p.  Tiles past the end of the K range are staged
// from the last valid tile (never read).
// The LDS image of a region is [128 rows][64 k] bf16 with the XOR slot swizzle of the other kernels (applied on the DMA source
// address); image row r of half i = wave slice (r >> 6), tile row (r >> 6) * 128 + i * 64 + (r & 63).
// Requirements: K a multiple of 64 (other shapes stay on gemm256p_kernel).  Rows past M / N are never fetched: their lanes carry
// an out-of-range buffer offset and the DMA writes zeros.
typedef __attribute__((address_space(3))) char* lds_cptr_t;

// ABL (debug, results wrong by construction): 1 = no DMA in the steady state, 2 = no fragment reads in the steady state, 3 = no barriers
// NF3: the 256 x 192 tile (round 3).  Same program, but a wave owns 128 x 96: its W slice is three fragment columns -- W half 0 =
// 64 rows as before, W half 1 = 32 rows -- so the phases that use W1 issue 8 MFMAs instead of 16, W1 is staged with 2 DMA pieces per
// wave instead of 4 (8 pieces = 2 slices x 32 image rows; the region's other 64 image rows are never written or read) and read as
// one fragment set; every counted wait drops by those 2 pieces (16 -> 14, prologue 24 -> 20).  Why: at M = 1542 the gate+up GEMM is
// 576 full 256^2 tiles = 2.25 per CU, i.e. THREE tile times for 2.25 tiles of work; 192-column tiles make it 768 = exactly 3 per CU
// of 3/4-size tiles, and qkv 192 full tiles instead of 144 on 256 CUs.
// The main loop of one (tile, k range): zeroes `acc` ([n fragment of 32][m fragment of 32]), runs K tiles [kt0, kt1) of the tile at
// (m0, n0) and returns with every LDS-DMA of this wave retired (the caller synchronises the block before it reuses `smem`).
// (The dynamic LDS is ONE object at namespace scope, named by every kernel of the translation unit: a `char*` parameter made the
// fragment-read addresses 64 run-time VGPR sums instead of instruction offsets, and a second __shared__ object de-pipelines LDS-DMA
// loops -- cdna guide, "three .s-level traps".)
// ROW5 (round 4; 192-column tile only): the LAST row tile also carries the up to 32 rows past its 256 (M = 1542 = 6 x 256 + 6: six row
// tiles instead of seven -- a tile of six rows costs as much as a full one, profiles/r04_gemm_sk.md section 3).  The wave row wm = 1
// of that tile owns a FIFTH 32-row fragment row: 3 more accumulator fragments (acc5; the 192-column tile leaves 64 accumulator
// registers free), its A rows staged as ONE more LDS-DMA piece per wave and K tile into the half of the W1 region the 192-column tile
// never uses (image rows 32..63), issued with the A1 group (so every counted wait grows by one: 15 / prologue 22), read into four
// registers next to A1 in phase 2 and multiplied in phases 3 (x W1) and 4 (x W0): 12 more MFMAs per K tile for two of the block's
// waves.  Every block of a ROW5 launch issues the extra piece (out of range => zeros, no fetch) so that the counts are uniform; only
// the last row tile's wm = 1 waves run the k loop copy that reads and multiplies it (rows5 > 0).
template <int ACT, bool HAS_BIAS, bool HAS_RES, bool BAR2 = true, int ABL = 0, bool NF3 = false, bool ROW5 = false>
__device__ __forceinline__ void gemm256q_mainloop(const GemmArgs& a, int m0, int n0, int kt0, int kt1, int lane, int wave,
                                                  f32x16_t (&acc)[4][4], float zero = 0.f, int rows5 = 0, f32x16_t* acc5 = nullptr) {   // zero: see gemm256s_kernel
  char* const smem = gemm_smem;
  constexpr int REGION = 128 * 128, STAGE = 4 * REGION;
  constexpr int TN = NF3 ? 192 : 256, WC = TN / 2;   // tile columns, columns per wave
  static_assert(!NF3 || (BAR2 && ABL == 0), "the 192-column tile exists in the shipped schedule only");
  static_assert(!ROW5 || NF3, "the fifth fragment row needs the 192-column tile's free accumulator registers and LDS rows");
  const int wm = wave >> 1, wn = wave & 1;

  // buffer resources: base = first row of this block's panel, offsets below are relative to it (< 2^31: 256 rows)
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)a.A + (int64_t)m0 * a.lda), 0, 0x40000000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)a.W + (int64_t)n0 * a.ldw), 0, 0x40000000, 0x00020000);
  // staging: region = 16 pieces of 8 image rows (1 KiB); wave w issues pieces w*4 + q.  Lane L of a piece lands on image row
  // g*8 + (L >> 3), physical slot L & 7, and therefore fetches logical slot (L & 7) ^ ((row >> 1) & 7).
  int voA[2][4], voW[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ir = (wave * 4 + q) * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ ((ir >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int trow = (ir >> 6) * 128 + h * 64 + (ir & 63);
      voA[h][q] = (m0 + trow < a.M) ? (int)((int64_t)trow * a.lda * 2 + slot * 16) : (int)0x80000000;
      if (!NF3) {
        voW[h][q] = (n0 + trow < a.N) ? (int)((int64_t)trow * a.ldw * 2 + slot * 16) : (int)0x80000000;
      } else if (h == 0) {   // W half 0: rows 0..63 of the wave's 96-row slice
        const int wrow = (ir >> 6) * WC + (ir & 63);
        voW[0][q] = (n0 + wrow < a.N) ? (int)((int64_t)wrow * a.ldw * 2 + slot * 16) : (int)0x80000000;
      } else {               // W half 1: rows 64..95; pieces p = wave * 2 + q (q < 2) cover image rows (p >> 2) * 64 + (p & 3) * 8 ..
        const int p1 = wave * 2 + (q & 1), ir1 = (p1 >> 2) * 64 + (p1 & 3) * 8 + (lane >> 3);
        const int slot1 = (lane & 7) ^ ((ir1 >> 1) & 7);
        const int wrow = (ir1 >> 6) * WC + 64 + (ir1 & 63);
        voW[1][q] = (n0 + wrow < a.N) ? (int)((int64_t)wrow * a.ldw * 2 + slot1 * 16) : (int)0x80000000;
      }
    }
  }
  // ROW5: tile rows 256 .. 287 -> image rows 32 .. 63 of the W1 region; wave w stages image rows 32 + 8 w ..
  int voA5 = (int)0x80000000;
  if (ROW5) {
    const int ir = 32 + wave * 8 + (lane >> 3), trow = 256 + wave * 8 + (lane >> 3);
    const int slot = (lane & 7) ^ ((ir >> 1) & 7);
    if (rows5 > 0 && m0 + trow < a.M) voA5 = (int)((int64_t)trow * a.lda * 2 + slot * 16);
  }
  const lds_cptr_t lbase = (lds_cptr_t)smem;
  // one DMA piece: region R (0..3 = A0 A1 W0 W1) of tile kt, piece q of this wave
  auto dma = [&](int kt, int R, int q) __attribute__((always_inline)) {
    if (NF3 && R == 3 && q >= 2) return;   // W half 1 of the 192-column tile: two pieces per wave
    const int ktc = min(kt, kt1 - 1);  // past the end: the last valid tile again (into the slot the schedule assigns; never read)
    const int piece = (NF3 && R == 3) ? (((wave * 2 + q) >> 2) * 8 + ((wave * 2 + q) & 3)) : wave * 4 + q;   // 1-KiB slot in the region
    const lds_cptr_t dst = lbase + ((kt - kt0) & 1) * STAGE + R * REGION + piece * 1024;
    const int so = ktc * (BK * 2);
    if (R < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)dst, 16, voA[R & 1][q], so, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)dst, 16, voW[R & 1][q], so, 0, 0);
  };
  auto dma5 = [&](int kt) __attribute__((always_inline)) {   // the fifth fragment row's piece of this wave (rides with the A1 group)
    const int ktc = min(kt, kt1 - 1);
    const lds_cptr_t dst = lbase + ((kt - kt0) & 1) * STAGE + 3 * REGION + (4 + wave) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)dst, 16, voA5, ktc * (BK * 2), 0, 0);
  };
  // fragment reads: set of 8 = [32-row fragment f][k-step ks]; lane (r32, hk) reads image row slice*64 + f*32 + r32, slot ks*2 + hk
  const int r32 = lane & 31, hk = lane >> 5;
  // Sixteen per-lane base addresses (stage x {A slice, W slice} x k-step), everything else of a fragment address -- region, 32-row
  // fragment -- is a compile-time constant below 64 KiB that rides in the ds_read offset field.  Written out because the compiler's own
  // factoring of `stage * STAGE + R * REGION + (slice * 64 + f * 32) * 128 + loff` depends on the surrounding code: as the body of a
  // function it kept 64 distinct addresses in registers for each copy of the k loop.
  int lb[2][2][4];   // [stage][0 = this wave's A slice (wm), 1 = its W slice (wn)][k-step]
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int loff = r32 * 128 + (((ks * 2 + hk) ^ ((r32 >> 1) & 7)) << 4);
#pragma unroll
    for (int stage = 0; stage < 2; ++stage) {
      lb[stage][0][ks] = stage * STAGE + (wm & 1) * 8192 + loff;
      lb[stage][1][ks] = stage * STAGE + (wn & 1) * 8192 + loff;
    }
  }
  auto frag = [&](int stage, int R, int slice, int f, int ks) __attribute__((always_inline)) {   // slice: wm for the A regions (R 0, 1), wn for W (2, 3)
    (void)slice;
    return as_frag(*reinterpret_cast<const u32x4_t*>(smem + lb[stage][R >> 1][ks] + (R * REGION + f * 32 * 128)));
  };
  // the fifth fragment row (read by wm = 1 waves only: their A base carries + 8192): image rows 32 .. 63 of region 3
  auto frag5 = [&](int stage, int ks) __attribute__((always_inline)) {
    return as_frag(*reinterpret_cast<const u32x4_t*>(smem + lb[stage][0][ks] + (3 * REGION + 32 * 128 - 8192)));
  };
  bf16x8_t fa5[4];

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = zero;
  bf16x8_t fa[2][2][4], fw[2][2][4];  // [register set][fragment][k-step]; A half i lives in fa[i], W half j of tile t in fw[j ^ parity(t)]

  const int nmf = __builtin_amdgcn_readfirstlane(max(0, min(4, (a.M - (m0 + wm * 128) + 31) / 32)));  // valid 32-row fragments

#define AHA_WAIT(imm) __builtin_amdgcn_s_waitcnt(imm)
#define AHA_BAR()                                \
  do {                                           \
    __builtin_amdgcn_sched_barrier(0);           \
    __builtin_amdgcn_s_barrier();                \
    __builtin_amdgcn_sched_barrier(0);           \
  } while (0)
  // prologue: tile kt0 and kt0+1 requested in steady-state order; A0 / W0 of kt0 into registers; then A0(kt0+2)
  {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0, 0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0, 2, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0, 3, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0, 1, q);
    if (ROW5) dma5(kt0);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0 + 1, 0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0 + 1, 2, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0 + 1, 3, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0 + 1, 1, q);
    if (ROW5) dma5(kt0 + 1);
  }
  if (ROW5) AHA_WAIT(0x4F76);      // vmcnt(22): 30 pieces requested, A0 and W0 of kt0 (8) have landed
  else if (NF3) AHA_WAIT(0x4F74);  // vmcnt(20): 28 pieces requested, A0 and W0 of kt0 (8) have landed
  else AHA_WAIT(0x4F78);           // vmcnt(24): A0, W0 of kt0 have landed
  AHA_BAR();
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[0][f][ks] = frag(0, 0, wm, f, ks);
      fw[0][f][ks] = frag(0, 2, wn, f, ks);
    }
  AHA_WAIT(0xC07F);  // lgkmcnt(0)
  if (!BAR2) {
    AHA_BAR();
#pragma unroll
    for (int q = 0; q < 4; ++q) dma(kt0 + 2, 0, q);
  }

  // one phase: [barrier,] then 4 k-steps of { 4 MFMAs, fragment reads for a later phase, 1 DMA piece }
  auto phase = [&](auto full_tag, auto r5_tag, auto bar_tag, int mi, bf16x8_t (&A)[2][4], int nj, bf16x8_t (&Wf)[2][4], bf16x8_t (&dst)[2][4], int rstage,
                   int rR, int rslice, int dkt, int dR) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr bool R5 = decltype(r5_tag)::value;   // this wave owns a fifth fragment row (wm = 1 of the last row tile of a ROW5 launch)
    if (decltype(bar_tag)::value) {
      if (ROW5) AHA_WAIT(0x007F);       // vmcnt(15) lgkmcnt(0): 4 + 4 + 2 + 5
      else if (NF3) AHA_WAIT(0x007E);   // vmcnt(14) lgkmcnt(0): the four newer groups are 4 + 4 + 2 + 4 pieces
      else if (BAR2) AHA_WAIT(0x4070);  // vmcnt(16) lgkmcnt(0)
      else AHA_WAIT(0x4078);            // vmcnt(24) lgkmcnt(0)
      if (ABL != 3) AHA_BAR();
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      auto mm = [&](int nf, int mf) __attribute__((always_inline)) {
        if (NF3 && nj == 1 && nf == 1) return;   // the 192-column tile has no fourth fragment column
        if (FULL || mi * 2 + mf < nmf) acc[nj * 2 + nf][mi * 2 + mf] = mfma32(Wf[nf][ks], A[mf][ks], acc[nj * 2 + nf][mi * 2 + mf]);
      };
      // k-step: MFMA | DMA piece | MFMA | fragment reads | MFMA MFMA.  The 8 fragment reads go out in the first three k-steps (3, 3, 2):
      // the last ones still have 1.5 k-steps of MFMAs to land in before the next wait.  (Placement A/B on MI355X: the piece in
      // front of the k-step's MFMAs -1 %; each wave's piece behind a different MFMA under scalar branches -17 %.)
      mm(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL != 1) dma(dkt, dR, ks);
      if (ROW5 && dR == 1 && ks == 3) dma5(dkt);   // the fifth row's piece closes the A1 group
      __builtin_amdgcn_sched_barrier(0);
      mm(0, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = ks * 3; r < min((NF3 && rR == 3) ? 4 : 8, ks * 3 + 3); ++r)   // (W half 1 of the 192-column tile: one fragment)
        if (ABL != 2) dst[r >> 2][r & 3] = frag(rstage, rR, rslice, r >> 2, r & 3);
      if (R5 && rR == 1) fa5[ks] = frag5(rstage, ks);   // beside A1(t), phase 2: one read per k-step
      __builtin_amdgcn_sched_barrier(0);
      mm(1, 0);
      mm(1, 1);
      if (R5 && mi == 1) {   // phases 3 (W1: one fragment column) and 4 (W0: two)
        acc5[nj * 2] = mfma32(Wf[0][ks], fa5[ks], acc5[nj * 2]);
        if (nj == 0) acc5[1] = mfma32(Wf[1][ks], fa5[ks], acc5[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto tile = [&](auto full_tag, auto r5_tag, auto par_tag, int kt) __attribute__((always_inline)) {
    constexpr int P = decltype(par_tag)::value;  // parity of (kt - kt0): stage of this tile, and which W register set holds W0
    constexpr std::true_type bar{};
    if (!BAR2) {
      phase(full_tag, r5_tag, bar, 0, fa[0], 0, fw[P], fw[P ^ 1], P, 3, wn, kt + 2, 2);      // A0 W0 | read W1(t)   | stage W0(t+2)
      phase(full_tag, r5_tag, bar, 0, fa[0], 1, fw[P ^ 1], fa[1], P, 1, wm, kt + 2, 3);      // A0 W1 | read A1(t)   | stage W1(t+2)
      phase(full_tag, r5_tag, bar, 1, fa[1], 1, fw[P ^ 1], fa[0], P ^ 1, 0, wm, kt + 2, 1);  // A1 W1 | read A0(t+1) | stage A1(t+2)
      phase(full_tag, r5_tag, bar, 1, fa[1], 0, fw[P], fw[P ^ 1], P ^ 1, 2, wn, kt + 3, 0);  // A1 W0 | read W0(t+1) | stage A0(t+3)
    } else {
      // BAR2: a barrier every SECOND phase.  A region is restaged two phases after it was read (the barrier in between covers
      // both), its data is read six phases after the request; at a barrier the groups of the two coming reads have landed and
      // the four newer ones may still be in flight: vmcnt(16).
      constexpr std::false_type nobar{};
      phase(full_tag, r5_tag, bar, 0, fa[0], 0, fw[P], fw[P ^ 1], P, 3, wn, kt + 2, 0);        // A0 W0 | read W1(t)   | stage A0(t+2)
      phase(full_tag, r5_tag, nobar, 0, fa[0], 1, fw[P ^ 1], fa[1], P, 1, wm, kt + 2, 2);      // A0 W1 | read A1(t)   | stage W0(t+2)
      phase(full_tag, r5_tag, bar, 1, fa[1], 1, fw[P ^ 1], fa[0], P ^ 1, 0, wm, kt + 2, 3);    // A1 W1 | read A0(t+1) | stage W1(t+2)
      phase(full_tag, r5_tag, nobar, 1, fa[1], 0, fw[P], fw[P ^ 1], P ^ 1, 2, wn, kt + 2, 1);  // A1 W0 | read W0(t+1) | stage A1(t+2)
    }
  };
  auto k_loop = [&](auto full_tag, auto r5_tag) __attribute__((always_inline)) {
    int kt = kt0;
    for (; kt + 1 < kt1; kt += 2) {
      tile(full_tag, r5_tag, std::integral_constant<int, 0>{}, kt);
      tile(full_tag, r5_tag, std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < kt1) tile(full_tag, r5_tag, std::integral_constant<int, 0>{}, kt);
  };
  if (ROW5) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc5[i][e] = zero;
  }
  if (ROW5 && rows5 > 0 && wm == 1) k_loop(std::true_type{}, std::true_type{});   // (the tile that carries a fifth row is full)
  else if (nmf == 4) k_loop(std::true_type{}, std::false_type{});
  else k_loop(std::false_type{}, std::false_type{});
  AHA_WAIT(0x0F70);  // vmcnt(0): nothing may still be writing this block's LDS when it retires (or reuses it below)
#undef AHA_WAIT
#undef AHA_BAR
}

}  // namespace

}  // namespace aha
