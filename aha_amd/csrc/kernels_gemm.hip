// Prefill / ViT projection GEMM on the matrix cores:
//
//   C[M,N] = A[M,K] . W[N,K]^T  (+ bias[N]) (act) (+ residual[M,N])      bf16 in / bf16 out, f32 accumulate
//
// Both operands are K-contiguous (activations row-major, candle_nn::Linear weights (out,in) row-major), which is the
// natural MFMA feed on CDNA: every fragment is a 16-byte run of one row.  v_mfma_f32_16x16x32_bf16, 128x128x64 block
// tile, 4 waves (2x2, 64x64 each), LDS double buffer, XOR-swizzled 16-byte slots (conflict-free ds_read_b128).
// The MFMA is issued as W-fragment x A-fragment so that each lane ends up with 4 consecutive output columns of one row:
// bias / activation / gate*up pairing / residual are then lane-local and the store is 8 bytes.
// Rounding points follow the reference op boundaries (Linear matmul -> bf16, + bias -> bf16, act -> bf16, + residual -> bf16).
#include "common.h"
#include "kernels.h"

namespace aha {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand per stage

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

template <int ACT, bool HAS_BIAS, bool HAS_RES>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, c = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give each XCD a contiguous run of
  // tiles (which share A row panels / W column panels in its private L2).  Bijective for any grid size.
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN, nwg = ntm * ntn;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // within an XCD's run walk N fastest in groups of 8 column tiles so the A panel stays hot
  const int tn = bid % ntn, tm = bid / ntn;
  const int m0 = tm * BM, n0 = tn * BN;

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
    const int k = kt * BK + kslot;
    const bool ok = k < a.K;
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
    const char* sw = sa + TILE_BYTES;
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
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = .. + c][n = .. + G*4 + 0..3] ----------------------------------------------
  bf16_t* C = (bf16_t*)a.C;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + c;
    if (m >= a.M) continue;
    if (ACT == ACT_SILU_MUL_PAIRS) {
      // W rows come in 16-row blocks: gate rows j..j+15 then up rows j..j+15 (the model loader interleaves them)
#pragma unroll
      for (int np = 0; np < 2; ++np) {
        const int n = n0 + wn * 64 + np * 32 + G * 4;  // fused-weight row of the gate values
        if (n >= a.N) continue;
        const int oc = (n0 + wn * 64) / 2 + np * 16 + G * 4;
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
        const int n = n0 + wn * 64 + ni * 16 + G * 4;
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

template <int ACT>
void launch_act(const GemmArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = 4 * TILE_BYTES;
  if (a.bias && a.residual) hipLaunchKernelGGL((gemm_kernel<ACT, true, true>), grid, dim3(256), lds, st, a);
  else if (a.bias) hipLaunchKernelGGL((gemm_kernel<ACT, true, false>), grid, dim3(256), lds, st, a);
  else if (a.residual) hipLaunchKernelGGL((gemm_kernel<ACT, false, true>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((gemm_kernel<ACT, false, false>), grid, dim3(256), lds, st, a);
}

}  // namespace

void launch_gemm(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return;
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  dim3 grid(ntm * ntn);
  switch (a.act) {
    case ACT_NONE: launch_act<ACT_NONE>(a, grid, st); break;
    case ACT_GELU_TANH: launch_act<ACT_GELU_TANH>(a, grid, st); break;
    case ACT_GELU_ERF: launch_act<ACT_GELU_ERF>(a, grid, st); break;
    case ACT_SILU: launch_act<ACT_SILU>(a, grid, st); break;
    case ACT_SILU_MUL_PAIRS: hipLaunchKernelGGL((gemm_kernel<ACT_SILU_MUL_PAIRS, false, false>), grid, dim3(256), 4 * TILE_BYTES, st, a); break;
  }
}

}  // namespace aha
