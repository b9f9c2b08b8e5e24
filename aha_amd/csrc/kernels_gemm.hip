// Prefill / ViT projection GEMM on the matrix cores:
//
//   C[M,N] = A[M,K] . W[N,K]^T  (+ bias[N]) (act) (+ residual[M,N])      bf16 in / bf16 out, f32 accumulate
//
// Both operands are K-contiguous (activations row-major, candle_nn::Linear weights (out,in) row-major), which is the
// natural MFMA feed on CDNA: every fragment is a 16-byte run of one row.  v_mfma_f32_16x16x32_bf16, 128x128x64 block
// tile, 4 waves (2x2, 64x64 each), XOR-swizzled 16-byte LDS slots (conflict-free ds_read_b128).
// Two staging variants:
//   gemm_glds_kernel (default): global_load_lds_dwordx4 straight into a 32-KiB LDS tile (no VGPR round trip, no
//       ds_write pass); the swizzle is applied on the per-lane SOURCE address because the DMA destination is
//       lane-linear; 32 KiB LDS and ~110 VGPRs let 4 blocks share a CU, which is what overlaps load and MFMA.
//   gemm_kernel (AHA_GEMM_GLDS=0): register-staged double buffer, kept for A/B.
// The MFMA is issued as W-fragment x A-fragment so that each lane ends up with 4 consecutive output columns of one row:
// bias / activation / gate*up pairing / residual are then lane-local and the store is 8 bytes.
// Rounding points follow the reference op boundaries (Linear matmul -> bf16, + bias -> bf16, act -> bf16, + residual -> bf16).
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace aha {

namespace {

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
  // candle Tensor::gelu / Activation::GeluPytorchTanh: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float k = 0.7978845608028654f;
  return 0.5f * x * (1.0f + tanhf(k * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give each XCD a contiguous run of tiles
// (which share A row panels / W column panels in its private L2).  Bijective for any grid size.
template <int TBM = BM, int TBN = BN>
__device__ __forceinline__ void tile_of_block(const GemmArgs& a, int& m0, int& n0) {
  const int ntm = (a.M + TBM - 1) / TBM, ntn = (a.N + TBN - 1) / TBN, nwg = ntm * ntn;
  int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // Within an XCD's run the SHORTER tile dimension varies fastest, so the ~32 tiles in flight on the XCD form a compact
  // rectangle (few distinct A / W panels in its L2) instead of one long row: PMC FETCH_SIZE of the gate/up GEMM
  // (7 x 96 tiles) was 1.47 GB per launch with row-major order -- every W panel re-fetched for each of the 7 row tiles.
  if (ntm <= ntn) {
    m0 = (bid % ntm) * TBM;
    n0 = (bid / ntm) * TBN;
  } else {
    m0 = (bid / ntn) * TBM;
    n0 = (bid % ntn) * TBN;
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

// ---- epilogue: lane holds C[m = .. + c][n = .. + G*4 + 0..3] -------------------------------------------------------
// mb / nb: first row / column of the wave's sub-tile (MI x 4 fragments of 16 x 16)
template <int ACT, bool HAS_BIAS, bool HAS_RES, int MI>
__device__ __forceinline__ void epilogue(const GemmArgs& a, f32x4_t (&acc)[4][MI], int mb, int nb, int G, int c) {
  bf16_t* C = (bf16_t*)a.C;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mb + mi * 16 + c;
    if (m >= a.M) continue;
    if (ACT == ACT_PARTIAL_F32) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = nb + ni * 16 + G * 4;
        if (n >= a.N) continue;
        *reinterpret_cast<float4*>((float*)a.C + (int64_t)m * a.ldc + n) =
            make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
      }
    } else if (ACT == ACT_SILU_MUL_PAIRS) {
      // W rows come in 16-row blocks: gate rows j..j+15 then up rows j..j+15 (the model loader interleaves them)
#pragma unroll
      for (int np = 0; np < 2; ++np) {
        const int n = nb + np * 32 + G * 4;  // fused-weight row of the gate values
        if (n >= a.N) continue;
        const int oc = nb / 2 + np * 16 + G * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gte = rbf(silu_f(rbf(acc[2 * np][mi][r])));  // gate_proj -> bf16, act_fn -> bf16 (modules.rs:82)
          const float up = rbf(acc[2 * np + 1][mi][r]);            // up_proj -> bf16 (modules.rs:83)
          v[r] = gte * up;                                          // lhs * rhs -> bf16 (modules.rs:84)
        }
        uint2 w2;
        w2.x = pack_bf(v[0], v[1]);
        w2.y = pack_bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(C + (int64_t)m * a.ldc + oc) = w2;
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = nb + ni * 16 + G * 4;
        if (n >= a.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(acc[ni][mi][r]);
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
        *reinterpret_cast<uint2*>(C + (int64_t)m * a.ldc + n) = w2;
      }
    }
  }
}

// ---- variant 1: direct-to-LDS staging -------------------------------------------------------------------------------
// Per K tile each wave issues 4 + 4 global_load_lds_dwordx4 (1 KiB each: 8 rows x 128 B of the LDS image).  Lane i of
// round j writes LDS row (j*4+wave)*8 + i/8, slot position i%8; it therefore READS logical slot (i%8) ^ f(row) from
// global memory (source-side swizzle), and the fragment reads apply the same XOR.  K tails / nothing-to-load lanes
// point at a 16-byte zero block.
template <int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(256) void gemm_glds_kernel(GemmArgs a, const void* zeros) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [A tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, c = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0;
  tile_of_block(a, m0, n0);
  const bf16_t* A = (const bf16_t*)a.A;
  const bf16_t* W = (const bf16_t*)a.W;
  const int nk = (a.K + BK - 1) / BK;

  const bf16_t* ga[4];
  const bf16_t* gw[4];
  int kofs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const int s = (lane & 7) ^ ((row >> 1) & 7);  // logical k-slot this lane fetches
    kofs[j] = s * 8;
    ga[j] = A + (int64_t)min(m0 + row, a.M - 1) * a.lda + s * 8;
    gw[j] = W + (int64_t)min(n0 + row, a.N - 1) * a.ldw + s * 8;
  }
  char* sa = smem;
  char* sw = smem + TILE_BYTES;

  f32x4_t acc[4][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = kt * BK + kofs[j] < a.K;
      const void* pa = ok ? (const void*)(ga[j] + kt * BK) : zeros;
      const void* pw = ok ? (const void*)(gw[j] + kt * BK) : zeros;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)pa, (lds_ptr_t)(sa + (j * 4 + wave) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)pw, (lds_ptr_t)(sw + (j * 4 + wave) * 1024), 16, 0, 0);
    }
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier
    mma_tile(sa, sw, wm, wn, G, c, acc);
    __syncthreads();
  }
  epilogue<ACT, HAS_BIAS, HAS_RES, 4>(a, acc, m0 + wm * 64, n0 + wn * 64, G, c);
}

// ---- variant 2: register-staged double buffer ----------------------------------------------------------------------
template <int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, c = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0;
  tile_of_block(a, m0, n0);
  const bf16_t* A = (const bf16_t*)a.A;
  const bf16_t* W = (const bf16_t*)a.W;
  const int nk = (a.K + BK - 1) / BK;

  // staging assignment: 1024 16-byte pieces per operand tile, 4 per thread; piece p -> row p>>3, slot p&7
  const bf16_t* ga[4];
  const bf16_t* gw[4];
  int lds_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = tid + i * 256, row = p >> 3, slot = p & 7;
    ga[i] = A + (int64_t)min(m0 + row, a.M - 1) * a.lda + slot * 8;
    gw[i] = W + (int64_t)min(n0 + row, a.N - 1) * a.ldw + slot * 8;
    lds_off[i] = swz(row, slot);
  }
  const int kslot = (tid & 7) * 8;  // k offset of this thread's pieces inside a K tile

  u32x4_t ra[4], rw[4];
  auto gload = [&](int kt) {
    const bool ok = kt * BK + kslot < a.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = ok ? ld16(ga[i] + kt * BK) : u32x4_t{0u, 0u, 0u, 0u};
      rw[i] = ok ? ld16(gw[i] + kt * BK) : u32x4_t{0u, 0u, 0u, 0u};
    }
  };
  auto lstore = [&](int stage) {
    char* sa = smem + stage * 2 * TILE_BYTES;
    char* sw = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4_t*>(sa + lds_off[i]) = ra[i];
      *reinterpret_cast<u32x4_t*>(sw + lds_off[i]) = rw[i];
    }
  };

  f32x4_t acc[4][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* sa = smem + cur * 2 * TILE_BYTES;
    mma_tile(sa, sa + TILE_BYTES, wm, wn, G, c, acc);
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }
  epilogue<ACT, HAS_BIAS, HAS_RES, 4>(a, acc, m0 + wm * 64, n0 + wn * 64, G, c);
}

// ---- variant 3: 256 x 256 x 64 tile, 8 waves (2 x 4, 128 x 64 each), two LDS stages, one barrier per K tile -------------
// For M >= ~1k.  Per K tile a wave issues 4 + 4 global_load_lds_dwordx4 for tile kt+1 into the other stage, then runs
// 64 MFMAs (8 A fragments x 4 W fragments x 2 k-steps) on stage kt; __syncthreads() drains the DMA (vmcnt(0)) and is the
// only barrier of the iteration.  128 KiB LDS, ~210 VGPRs => 1 block (8 waves) per CU; each LDS byte read feeds 2.7
// MFMAs (2 in the 128^2 kernel).  gridDim.y > 1 = split-K: slice z accumulates k tiles [z*kps, (z+1)*kps) and writes an
// f32 slab a.C + z*M*ldc (ACT_PARTIAL_F32); gemm_splitk_reduce_kernel sums the slabs and applies the epilogue chain.
constexpr int BM2 = 256, BN2 = 256;
constexpr int TILE2_BYTES = 256 * BK * 2;  // 32 KiB per operand per stage

template <int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs a, const void* zeros, int kt_per_slice) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, c = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;
  int m0, n0;
  tile_of_block<BM2, BN2>(a, m0, n0);
  const bf16_t* A = (const bf16_t*)a.A;
  const bf16_t* W = (const bf16_t*)a.W;
  const int nk_all = (a.K + BK - 1) / BK;
  const int kt0 = blockIdx.y * kt_per_slice, kt1 = min(nk_all, kt0 + kt_per_slice);
  if (ACT == ACT_PARTIAL_F32) a.C = (float*)a.C + (int64_t)blockIdx.y * a.M * a.ldc;

  // staging: a stage's operand tile is 32 row groups of 8 rows (1 KiB each); wave w fills groups j*8 + w, j = 0..3
  const bf16_t* ga[4];
  const bf16_t* gw[4];
  int kofs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (j * 8 + wave) * 8 + (lane >> 3);
    const int s = (lane & 7) ^ ((row >> 1) & 7);  // logical k-slot this lane fetches (source-side swizzle)
    kofs[j] = s * 8;
    ga[j] = A + (int64_t)min(m0 + row, a.M - 1) * a.lda + s * 8;
    gw[j] = W + (int64_t)min(n0 + row, a.N - 1) * a.ldw + s * 8;
  }
  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * 2 * TILE2_BYTES;
    char* sw = sa + TILE2_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = kt * BK + kofs[j] < a.K;
      const void* pa = ok ? (const void*)(ga[j] + kt * BK) : zeros;
      const void* pw = ok ? (const void*)(gw[j] + kt * BK) : zeros;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)pa, (lds_ptr_t)(sa + (j * 8 + wave) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)pw, (lds_ptr_t)(sw + (j * 8 + wave) * 1024), 16, 0, 0);
    }
  };

  f32x4_t acc[4][8];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) issue(kt0, 0);
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < kt1) issue(kt + 1, cur ^ 1);
    const char* sa = smem + cur * 2 * TILE2_BYTES;
    const char* sw = sa + TILE2_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[8], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = as_frag(*reinterpret_cast<const u32x4_t*>(sw + swz(wn * 64 + i * 16 + c, ks * 4 + G)));
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = as_frag(*reinterpret_cast<const u32x4_t*>(sa + swz(wm * 128 + i * 16 + c, ks * 4 + G)));
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni][mi] = mfma16(wf[ni], af[mi], acc[ni][mi]);
    }
    __syncthreads();  // stage cur is free again, and tile kt+1 has landed (the fence in front of the barrier is vmcnt(0))
  }
  epilogue<ACT, HAS_BIAS, HAS_RES, 8>(a, acc, m0 + wm * 128, n0 + wn * 64, G, c);
}

// Sums the split-K slabs and runs the same rounding chain as the in-kernel epilogue: Linear output -> bf16, + bias -> bf16,
// activation -> bf16, + residual -> bf16.  One thread per 4 consecutive columns.  (Not used with ACT_SILU_MUL_PAIRS.)
template <int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs a, const float* slabs, int nsl) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nq = a.N >> 2;
  if (q >= (int64_t)a.M * nq) return;
  const int m = (int)(q / nq), n = (int)(q % nq) * 4;
  const int64_t slab = (int64_t)a.M * a.N;
  float4 s = *reinterpret_cast<const float4*>(slabs + (int64_t)m * a.N + n);
  for (int z = 1; z < nsl; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(slabs + z * slab + (int64_t)m * a.N + n);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  float v[4] = {rbf(s.x), rbf(s.y), rbf(s.z), rbf(s.w)};
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
  bf16_t* C = (bf16_t*)a.C;
  if (HAS_RES) {
    const uint2 r2 = *reinterpret_cast<const uint2*>((const bf16_t*)a.residual + (int64_t)m * a.ldc + n);
    v[0] += lo_bf(r2.x); v[1] += hi_bf(r2.x); v[2] += lo_bf(r2.y); v[3] += hi_bf(r2.y);
  }
  uint2 w2;
  w2.x = pack_bf(v[0], v[1]);
  w2.y = pack_bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(C + (int64_t)m * a.ldc + n) = w2;
}

const void* zero_block() {
  static void* z = nullptr;
  if (!z) {
    hipMalloc(&z, 256);
    hipMemset(z, 0, 256);
  }
  return z;
}

template <int ACT, bool B, bool R>
void launch_one(const GemmArgs& a, dim3 grid, bool glds, hipStream_t st) {
  if (glds) hipLaunchKernelGGL((gemm_glds_kernel<ACT, B, R>), grid, dim3(256), 2 * TILE_BYTES, st, a, zero_block());
  else hipLaunchKernelGGL((gemm_kernel<ACT, B, R>), grid, dim3(256), 4 * TILE_BYTES, st, a);
}

template <int ACT>
void launch_act(const GemmArgs& a, dim3 grid, bool glds, hipStream_t st) {
  if (a.bias && a.residual) launch_one<ACT, true, true>(a, grid, glds, st);
  else if (a.bias) launch_one<ACT, true, false>(a, grid, glds, st);
  else if (a.residual) launch_one<ACT, false, true>(a, grid, glds, st);
  else launch_one<ACT, false, false>(a, grid, glds, st);
}

template <int ACT, bool B, bool R>
void launch256_one(const GemmArgs& a, int splitk, hipStream_t st) {
  const int ntm = (a.M + BM2 - 1) / BM2, ntn = (a.N + BN2 - 1) / BN2;
  const int nk = (a.K + BK - 1) / BK;
  const size_t lds = 4 * TILE2_BYTES;
  static bool attr_done = false;  // > 64 KiB of dynamic LDS needs the opt-in once per kernel instance
  if (splitk <= 1) {
    static bool once = false;
    if (!once) {
      hipFuncSetAttribute((const void*)gemm256_kernel<ACT, B, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      once = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<ACT, B, R>), dim3(ntm * ntn), dim3(512), lds, st, a, zero_block(), nk);
    return;
  }
  (void)attr_done;
  static bool once_p = false;
  if (!once_p) {
    hipFuncSetAttribute((const void*)gemm256_kernel<ACT_PARTIAL_F32, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    once_p = true;
  }
  GemmArgs p = a;  // pass 1: f32 slabs [splitk][M][N] in the caller's workspace
  p.C = a.workspace;
  p.ldc = a.N;
  p.bias = nullptr;
  p.residual = nullptr;
  p.act = ACT_PARTIAL_F32;
  const int kps = (nk + splitk - 1) / splitk;
  hipLaunchKernelGGL((gemm256_kernel<ACT_PARTIAL_F32, false, false>), dim3(ntm * ntn, splitk), dim3(512), lds, st, p, zero_block(), kps);
  const int64_t quads = (int64_t)a.M * (a.N >> 2);
  hipLaunchKernelGGL((gemm_splitk_reduce_kernel<ACT, B, R>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a,
                     (const float*)a.workspace, (nk + kps - 1) / kps);
}

template <int ACT>
void launch256_act(const GemmArgs& a, int splitk, hipStream_t st) {
  if (a.bias && a.residual) launch256_one<ACT, true, true>(a, splitk, st);
  else if (a.bias) launch256_one<ACT, true, false>(a, splitk, st);
  else if (a.residual) launch256_one<ACT, false, true>(a, splitk, st);
  else launch256_one<ACT, false, false>(a, splitk, st);
}

struct GemmPlan { int tile, splitk; };

// Tile choice by a two-parameter cost model fitted to scripts/bench_gemm.py on MI355X (256 CUs): a CU retires one
// (128^2 tile, 64-deep k step) in ~0.68 us when it has two or more 128^2 blocks resident, and one (256^2 tile, k step) in
// ~2.0 us (= 0.5 us per 128^2-equivalent: 1.35x better, measured 823 -> 1074 TFLOP/s at 8192^3); the last, partly filled
// round of work units costs a full round.  256^2 tiles therefore win when the unit count fills the CUs (long prompts,
// wide N) and lose on wave quantisation (e.g. 272 units = 2 rounds).  Few tiles + long K: split K into f32 slabs
// (caller's workspace) and pay a reduce pass that streams (splitk + 1) x M x N x 4 bytes.
int g_force_tile = 0, g_force_splitk = 0;  // aha_hip_debug_gemm_plan (tests): 0 = automatic

GemmPlan plan_gemm(const GemmArgs& a) {
  if (g_force_tile == 128) return GemmPlan{128, 1};
  if (g_force_tile == 256 && a.M >= 1) {
    int sk = g_force_splitk > 1 ? g_force_splitk : 1;
    if (sk > 1 && (a.act == ACT_SILU_MUL_PAIRS || a.act == ACT_PARTIAL_F32 || !a.workspace || (a.N & 3) ||
                   (size_t)sk * a.M * a.N * 4 > a.workspace_bytes)) sk = 1;
    return GemmPlan{256, sk};
  }
  static const char* e_tile = getenv("AHA_GEMM_TILE");
  static const char* e_sk = getenv("AHA_GEMM_SPLITK");
  const double nk = (a.K + BK - 1) / BK;
  const double t128 = (double)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const double t256 = (double)((a.M + 255) / 256) * ((a.N + 255) / 256);
  const double cost128 = ceil(t128 / 256.0) * nk * 0.68;
  const bool can_split = a.act != ACT_SILU_MUL_PAIRS && a.act != ACT_PARTIAL_F32 && a.workspace != nullptr && (a.N & 3) == 0;
  GemmPlan best{128, 1};
  double best_cost = cost128;
  for (int sk = 1; sk <= 4; sk *= 2) {
    if (sk > 1 && (!can_split || (size_t)sk * a.M * a.N * 4 > a.workspace_bytes || nk / sk < 8)) break;
    double c = ceil(t256 * sk / 256.0) * ceil(nk / sk) * 2.0;
    if (sk > 1) c += (double)(sk + 1) * a.M * a.N * 4.0 / 4.0e6 + 3.0;
    if (e_sk && atoi(e_sk) != sk) continue;
    if (c < best_cost || (e_tile && atoi(e_tile) == 256 && best.tile != 256)) {
      best = GemmPlan{256, sk};
      best_cost = c;
    }
  }
  if (a.M < 256) best = GemmPlan{128, 1};
  if (e_tile && atoi(e_tile) == 128) best = GemmPlan{128, 1};
  return best;
}

}  // namespace

void set_gemm_plan_override(int tile, int splitk) {
  g_force_tile = tile;
  g_force_splitk = splitk;
}

static thread_local void* tl_ws = nullptr;
static thread_local size_t tl_ws_bytes = 0;
void set_gemm_workspace(void* ws, size_t bytes) {
  tl_ws = ws;
  tl_ws_bytes = bytes;
}

void launch_gemm(const GemmArgs& a_in, hipStream_t st) {
  if (a_in.M <= 0 || a_in.N <= 0) return;
  GemmArgs a = a_in;
  if (a.workspace == nullptr) {
    a.workspace = tl_ws;
    a.workspace_bytes = tl_ws_bytes;
  }
  const GemmPlan plan = plan_gemm(a);
  if (plan.tile == 256) {
    switch (a.act) {
      case ACT_NONE: launch256_act<ACT_NONE>(a, plan.splitk, st); break;
      case ACT_GELU_TANH: launch256_act<ACT_GELU_TANH>(a, plan.splitk, st); break;
      case ACT_GELU_ERF: launch256_act<ACT_GELU_ERF>(a, plan.splitk, st); break;
      case ACT_SILU: launch256_act<ACT_SILU>(a, plan.splitk, st); break;
      case ACT_SILU_MUL_PAIRS: launch256_one<ACT_SILU_MUL_PAIRS, false, false>(a, 1, st); break;
      case ACT_PARTIAL_F32: launch256_one<ACT_PARTIAL_F32, false, false>(a, 1, st); break;
    }
    return;
  }
  static const bool glds = [] {
    const char* e = getenv("AHA_GEMM_GLDS");
    return e ? atoi(e) != 0 : true;
  }();
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  dim3 grid(ntm * ntn);
  switch (a.act) {
    case ACT_NONE: launch_act<ACT_NONE>(a, grid, glds, st); break;
    case ACT_GELU_TANH: launch_act<ACT_GELU_TANH>(a, grid, glds, st); break;
    case ACT_GELU_ERF: launch_act<ACT_GELU_ERF>(a, grid, glds, st); break;
    case ACT_SILU: launch_act<ACT_SILU>(a, grid, glds, st); break;
    case ACT_SILU_MUL_PAIRS: launch_one<ACT_SILU_MUL_PAIRS, false, false>(a, grid, glds, st); break;
    case ACT_PARTIAL_F32: launch_one<ACT_PARTIAL_F32, false, false>(a, grid, glds, st); break;
  }
}

}  // namespace aha
