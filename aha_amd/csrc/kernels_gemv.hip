// Batch-1 weight-streaming matvec: the kernel that bounds decode throughput.
//
//   y[n] = sum_k h[k] * W[n,k]     W (N,K) bf16 row-major (HF / candle_nn::Linear layout), h (K) bf16, f32 accumulate
//
// HBM-bound: every weight byte is read exactly once per token, so the design goal is nothing but keeping enough
// 16-byte non-temporal loads in flight per CU.  No LDS staging of W (read once, not shared between waves); the
// activation vector lives in LDS as f32 in a lane-linear image so each ds_read_b128 is conflict-free.
//   * one wave owns R consecutive rows; per 512-column chunk it issues R x U independent 1-KiB loads
//   * persistent grid-stride over row tiles; the (optional) fused RMSNorm prologue runs once per block
//   * epilogues fuse the reference's next op: residual add (qwen3/model.rs:81,86), silu(gate)*up
//     (modules.rs:81-87), f32 logits + argmax partials (generate.rs:75-84)
// Rounding points follow the reference's op boundaries: Linear output -> bf16, then each further op -> bf16.
#include <stdlib.h>

#include "gemv_body.h"

namespace aha {

namespace {

template <int R, int U, int EPI, bool FAST>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // K f32 + 8 floats reduction scratch
  gemv_body<R, U, EPI, false, FAST>(a, xs, (int)blockIdx.x, (int)gridDim.x, [] {});
}

struct GemvPlan { int R, U, grid; };

// Pick rows-per-wave so that there are at least ~2 tiles per CU (256 CUs), and keep R*U*NW <= 16 loads in flight.
GemvPlan plan_gemv(int N, int K, GemvEpi epi) {
  const int nchunks = (K + 511) / 512;
  const int nw = epi == GEMV_SILU_MUL ? 2 : 1;
  int R = 4;
  while (R > 1 && (N + 4 * R - 1) / (4 * R) < 1024) R >>= 1;
  if (nw == 2 && R > 2) R = 2;
  int U = 16 / (R * nw);
  if (U > nchunks) U = nchunks;
  if (U > 8) U = 8;
  if (U < 1) U = 1;
  // U must be one of the instantiated values {1,2,4,8}
  int Up = 1;
  while (Up * 2 <= U) Up *= 2;
  // persistent blocks: 3 per CU on 256 CUs (<= 160 VGPRs: 12 waves per CU).  A/B on the Qwen3-VL-8B decode step, same box:
  // 512 -> 267.7 tok/s, 640 -> 270.9, 768 -> 276.2, 1024 -> 269.6.
  int gmax = 768;
  // tuning knobs for scripts/bench_gemv.py (not used by the product path unless set)
  static const char* e_grid = getenv("AHA_GEMV_GRID");
  static const char* e_r = getenv("AHA_GEMV_R");
  static const char* e_u = getenv("AHA_GEMV_U");
  if (e_grid) gmax = atoi(e_grid);
  static const char* e_grid1 = getenv("AHA_GEMV_GRID_R1");  // separate cap for the 1-row-per-wave kernels (118 VGPRs: 4 blocks per CU fit)
  if (e_grid1 && R == 1) gmax = atoi(e_grid1);
  if (e_r) {
    R = atoi(e_r);
    if (nw == 2 && R > 2) R = 2;
  }
  if (e_u) Up = atoi(e_u);
  if (e_r || e_u) {
    if (Up > nchunks) Up = nchunks;
    int q = 1;
    while (q * 2 <= Up) q *= 2;
    Up = q;
  }
  const int ntiles = (N + 4 * R - 1) / (4 * R);
  int grid = ntiles < gmax ? ntiles : gmax;
  // Whole rounds: with 1024 tiles (N = 4096: o_proj, down_proj) 768 blocks leave a third of them a second tile and the rest idle
  // behind it; the largest grid in [gmax / 2, gmax] that divides the tile count (512 there: two tiles each) measured +1.9 % on the
  // 8B decode step (316.9 -> 323.0 tok/s, same box; 1024 and 1536 blocks: 313.9 / 304.6).  Multiples of 8 only (one share per XCD).
  static const bool even_on = [] { const char* e = getenv("AHA_GEMV_EVEN_GRID"); return e ? atoi(e) != 0 : true; }();
  if (even_on && !e_grid && !(e_grid1 && R == 1) && ntiles > gmax) {
    for (int g = gmax; g >= gmax / 2; g -= 8)
      if (ntiles % g == 0) {
        grid = g;
        break;
      }
  }
  return {R, Up, grid};
}

}  // namespace

int gemv_num_tiles(int N, int K) { return plan_gemv(N, K, GEMV_LOGITS).grid; }

template <int EPI>
static void launch_gemv_epi(const GemvArgs& a, const GemvPlan& p, hipStream_t st) {
  const size_t lds = (size_t)((a.K + 511) / 512) * 512 * 4 + 64;
  dim3 grid(p.grid), block(GEMV_THREADS);
  // FAST form (gemv_body.h): every chunk group full and in range, weights non-temporal
  static const bool fast_ok = [] { const char* e = getenv("AHA_GEMV_FAST"); return e ? atoi(e) != 0 : true; }();
  const bool fast = fast_ok && a.K % (512 * p.U) == 0 && !a.cached && a.N >= 1;
#define GV(RR, UU)                                                                            \
  do {                                                                                        \
    if (fast) hipLaunchKernelGGL((gemv_kernel<RR, UU, EPI, true>), grid, block, lds, st, a);  \
    else hipLaunchKernelGGL((gemv_kernel<RR, UU, EPI, false>), grid, block, lds, st, a);      \
  } while (0)
  if (p.R == 4) {
    if (p.U >= 4) GV(4, 4); else if (p.U == 2) GV(4, 2); else GV(4, 1);
  } else if (p.R == 2) {
    if (p.U >= 8) GV(2, 8); else if (p.U == 4) GV(2, 4); else if (p.U == 2) GV(2, 2); else GV(2, 1);
  } else {
    if (p.U >= 8) GV(1, 8); else if (p.U == 4) GV(1, 4); else if (p.U == 2) GV(1, 2); else GV(1, 1);
  }
#undef GV
}

void launch_gemv(const GemvArgs& a_in, GemvEpi epi, hipStream_t st) {
  GemvArgs a = a_in;
  static const char* e_cached = getenv("AHA_GEMV_CACHED");
  if (e_cached) a.cached = atoi(e_cached);
  const GemvPlan p = plan_gemv(a.N, a.K, epi);
  switch (epi) {
    case GEMV_STORE: launch_gemv_epi<GEMV_STORE>(a, p, st); break;
    case GEMV_RESIDUAL: launch_gemv_epi<GEMV_RESIDUAL>(a, p, st); break;
    case GEMV_SILU_MUL: launch_gemv_epi<GEMV_SILU_MUL>(a, p, st); break;
    case GEMV_LOGITS: launch_gemv_epi<GEMV_LOGITS>(a, p, st); break;
    case GEMV_PARTIAL_F32: launch_gemv_epi<GEMV_PARTIAL_F32>(a, p, st); break;
  }
}

}  // namespace aha
