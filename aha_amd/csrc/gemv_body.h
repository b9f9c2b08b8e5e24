// Body of the batch-1 weight-streaming matvec (kernels_gemv.hip has the design notes).
#pragma once
#include "common.h"
#include "kernels.h"

namespace aha {


constexpr int GEMV_THREADS = 256;
constexpr int GEMV_WAVES = 4;

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// LDS image of h: chunk c (512 elements) is stored as [half(2)][lane(64)][4] f32 so that a wave reading
// "its" 8 elements of the chunk does two fully contiguous 1-KiB ds_read_b128.
__device__ __forceinline__ int xs_index(int k) {
  const int c = k >> 9, r = k & 511, lane = r >> 3, e = r & 7;
  return (c << 9) + ((e >> 2) << 8) + (lane << 2) + (e & 3);
}

// xs: LDS, (ceil(K/512)*512 + 16) floats.  bid/nblk: this block's index in, and the size of, the persistent grid.
// after_issue() runs once the block's first weight tile has been requested and before anything that depends on the
// activation vector -- the stand-alone kernel passes a no-op, the persistent kernel waits on its grid barrier there.
template <int R, int U, int EPI, bool COH, bool FAST, class AfterIssue>
__device__ __forceinline__ void gemv_body(const GemvArgs& a, float* xs, const int bid, const int nblk, AfterIssue&& after_issue) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, N = a.N;
  const int nchunks = (K + 511) >> 9;  // K % 8 == 0; the tail of the last chunk is zero-filled
  float* red = xs + (nchunks << 9);

  // ---- flattened work list: this block's tiles x the chunk groups of a tile ---------------------------------------
  constexpr int ROWS_PER_TILE = GEMV_WAVES * R;
  constexpr int NW = (EPI == GEMV_SILU_MUL) ? 2 : 1;
  const int n_out = N;  // for SILU_MUL a "row" runs over the I outputs; the wave streams gate row j and up row j together
  const int ntiles = (n_out + ROWS_PER_TILE - 1) / ROWS_PER_TILE;
  const int gpt = (nchunks + U - 1) / U;                                  // chunk groups per tile
  const int my_tiles = (ntiles - bid + nblk - 1) / nblk;
  const int ngroups = my_tiles * gpt;
  const bf16_t* Wb = (const bf16_t*)a.W;
  const bf16_t* W2b = (const bf16_t*)a.W2;

  // FAST (K a multiple of 512 * U, non-temporal weights): no per-load predicates at all.  A load under a run-time select
  // (`ok ? load : 0`) makes hipcc branch around it and wait vmcnt(0) at the join, and a buffer refilled under `if (more)`
  // becomes a phi resolved with copies + vmcnt(0): the two-buffer pipeline then degenerates to one buffer per wave.  In
  // the FAST form the steady-state loop refills each buffer unconditionally right after it has been consumed, the waits
  // are counted (the other buffer's R*U*NW loads stay in flight), and the tile's residual values are requested in FRONT of
  // the tile's last weight group (in-order return) instead of inside the epilogue, where waiting for them drained the queue.
  // issue the R*U*NW 16-byte loads of work item gi (no dependence on x: they go out BEFORE the prologue)
  auto issue = [&](int gi, u32x4_t (&buf)[U][NW][R], bf16_t (&resv)[R]) {
    const int tile = bid + (gi / gpt) * nblk;
    const int c0 = (gi % gpt) * U;
    const int row0 = tile * ROWS_PER_TILE + wave * R;
    if (FAST && EPI == GEMV_RESIDUAL) {
#pragma unroll
      for (int r = 0; r < R; ++r) resv[r] = ((const bf16_t*)a.residual)[min(row0 + r, n_out - 1)];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = min(row0 + r, n_out - 1);  // clamp: out-of-range rows are computed but never stored
      const bf16_t* p0;
      const bf16_t* p1 = nullptr;
      if (NW == 2) {
        if (W2b != nullptr) {  // separate gate / up matrices (op-level entry point)
          p0 = Wb + (size_t)row * K;
          p1 = W2b + (size_t)row * K;
        } else {  // the model's fused matrix: 16-row blocks alternating gate / up
          const size_t fr = (size_t)(row >> 4) * 32 + (row & 15);
          p0 = Wb + fr * K;
          p1 = Wb + (fr + 16) * K;
        }
      } else {
        p0 = Wb + (size_t)row * K;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = ((c0 + u) << 9) + lane * 8;
        if (FAST) {
          buf[u][0][r] = ld_nt16(p0 + k);
          if (NW == 2) buf[u][NW - 1][r] = ld_nt16(p1 + k);
          continue;
        }
        const bool ok = (c0 + u < nchunks) && (k < K);
        buf[u][0][r] = ok ? (a.cached ? ld16(p0 + k) : ld_nt16(p0 + k)) : u32x4_t{0u, 0u, 0u, 0u};
        if (NW == 2) buf[u][NW - 1][r] = ok ? (a.cached ? ld16(p1 + k) : ld_nt16(p1 + k)) : u32x4_t{0u, 0u, 0u, 0u};
      }
    }
  };

  const int tslot = bid == 0 ? 0 : bid == 255 ? 1 : bid == nblk - 1 ? 2 : -1;
  auto stamp = [&](int k) {
    if (a.trace != nullptr && tslot >= 0 && tid == 0) a.trace[tslot * 6 + k] = wall_clock64();
  };
  stamp(0);
  // Memory returns are IN ORDER within a wave: a load issued behind the first weight tile only returns after that
  // tile has arrived (several microseconds under a chip-wide burst).  The activation vector and the norm weights the
  // prologue needs are therefore requested FIRST (stand-alone kernels, K <= 8 * 2048), the weight tile right behind them.
  constexpr int XPRE = 8;
  const bool pre = !COH && (nchunks << 6) <= XPRE * GEMV_THREADS;
  u32x4_t xpre[XPRE], npre[XPRE];
  if (pre) {
#pragma unroll
    for (int j = 0; j < XPRE; ++j) {
      const int v = tid + j * GEMV_THREADS;
      xpre[j] = u32x4_t{0u, 0u, 0u, 0u};
      npre[j] = u32x4_t{0u, 0u, 0u, 0u};
      if (v * 8 < K) {
        xpre[j] = ld16((const bf16_t*)a.x + v * 8);
        if (a.norm_w != nullptr) npre[j] = ld16((const bf16_t*)a.norm_w + v * 8);
      }
    }
  }
  u32x4_t bufA[U][NW][R], bufB[U][NW][R];
  bf16_t resA[R], resB[R];   // FAST + GEMV_RESIDUAL: the residual values of the buffer's tile, requested with it
  if (ngroups > 0) issue(0, bufA, resA);
  stamp(1);
  after_issue();  // grid barrier of the persistent decode kernel: the first weight tile is already in flight

  // ---- prologue: h = x, or h = bf16(RMSNorm(x) * norm_w) (qwen3/model.rs:79,83,186) ------------------------
  {
    const bf16_t* x = (const bf16_t*)a.x;
    const bf16_t* nw = (const bf16_t*)a.norm_w;
    float ss = 0.f;
    auto stage = [&](int v, u32x4_t xv) {  // bf16 x8 -> the f32 LDS image, running sum of squares
      float f[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[2 * j] = lo_bf(xv[j]); f[2 * j + 1] = hi_bf(xv[j]); }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
      const int base = xs_index(v * 8);
      *reinterpret_cast<float4*>(xs + base) = make_float4(f[0], f[1], f[2], f[3]);
      *reinterpret_cast<float4*>(xs + base + 256) = make_float4(f[4], f[5], f[6], f[7]);
    };
    if (pre) {
#pragma unroll
      for (int it = 0; it < XPRE; ++it) {
        const int v = tid + it * GEMV_THREADS;
        if (v < (nchunks << 6)) stage(v, xpre[it]);
      }
    } else {
      for (int v = tid; v < (nchunks << 6); v += GEMV_THREADS) {
        u32x4_t xv = {0u, 0u, 0u, 0u};
        if (v * 8 < K) xv = act_ld16<COH>(x + v * 8);
        stage(v, xv);
      }
    }
    if (nw != nullptr) {
      ss = wave_sum(ss);
      if (lane == 0) red[wave] = ss;
      __syncthreads();
      const float tot = red[0] + red[1] + red[2] + red[3];
      const float rinv = 1.0f / sqrtf(tot / (float)K + a.eps);
      auto norm = [&](int v, u32x4_t wv) {
        const int base = xs_index(v * 8);
        float4 lo = *reinterpret_cast<float4*>(xs + base), hi = *reinterpret_cast<float4*>(xs + base + 256);
        lo.x = rbf(lo.x * rinv * lo_bf(wv[0])); lo.y = rbf(lo.y * rinv * hi_bf(wv[0]));
        lo.z = rbf(lo.z * rinv * lo_bf(wv[1])); lo.w = rbf(lo.w * rinv * hi_bf(wv[1]));
        hi.x = rbf(hi.x * rinv * lo_bf(wv[2])); hi.y = rbf(hi.y * rinv * hi_bf(wv[2]));
        hi.z = rbf(hi.z * rinv * lo_bf(wv[3])); hi.w = rbf(hi.w * rinv * hi_bf(wv[3]));
        *reinterpret_cast<float4*>(xs + base) = lo;
        *reinterpret_cast<float4*>(xs + base + 256) = hi;
        if (a.h_out != nullptr && bid == 0) {
          u32x4_t o;
          o[0] = pack_bf(lo.x, lo.y); o[1] = pack_bf(lo.z, lo.w); o[2] = pack_bf(hi.x, hi.y); o[3] = pack_bf(hi.z, hi.w);
          *reinterpret_cast<u32x4_t*>((bf16_t*)a.h_out + v * 8) = o;
        }
      };
      if (pre) {
#pragma unroll
        for (int it = 0; it < XPRE; ++it) {
          const int v = tid + it * GEMV_THREADS;
          if (v < (K >> 3)) norm(v, npre[it]);
        }
      } else {
        for (int v = tid; v < (K >> 3); v += GEMV_THREADS) norm(v, ld16(nw + v * 8));
      }
    }
    __syncthreads();
  }

  // ---- main: consume item g while item g+1 is in flight (two named register buffers, static indexing) ------------
  float tile_best = -INFINITY;
  uint32_t tile_best_i = 0xffffffffu;
  float acc[NW][R];
#pragma unroll
  for (int m = 0; m < NW; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = 0.f;

  auto consume = [&](int gi, u32x4_t (&buf)[U][NW][R], bf16_t (&resv)[R]) {
    const int c0 = (gi % gpt) * U;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (FAST || c0 + u < nchunks) {
        const float4 xlo = *reinterpret_cast<const float4*>(xs + ((c0 + u) << 9) + (lane << 2));
        const float4 xhi = *reinterpret_cast<const float4*>(xs + ((c0 + u) << 9) + 256 + (lane << 2));
#pragma unroll
        for (int m = 0; m < NW; ++m)
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const u32x4_t w = buf[u][m][r];
            float s = acc[m][r];
            // ACTIVATION FIRST, weight second -- on purpose.  clang packs two rows' fmas into v_pk_fma_f32 and broadcasts the shared
            // activation element with op_sel; as the SECOND source that is `op_sel:[0,1,0]` for the odd elements, the form that was
            // measured to read the wrong element in lanes 16-31 / 48-63 when another stream's kernel shares the SIMD; as the FIRST
            // source (`op_sel:[1,0,0]` / `op_sel_hi:[0,1,1]`) it measured clean (profiles/r03_simd_coresidency.md; a * b == b * a, so
            // the bits do not change; tests/test_isa_cpu.py keeps the affected forms out of every kernel).
            s = fmaf(xlo.x, lo_bf(w[0]), s); s = fmaf(xlo.y, hi_bf(w[0]), s);
            s = fmaf(xlo.z, lo_bf(w[1]), s); s = fmaf(xlo.w, hi_bf(w[1]), s);
            s = fmaf(xhi.x, lo_bf(w[2]), s); s = fmaf(xhi.y, hi_bf(w[2]), s);
            s = fmaf(xhi.z, lo_bf(w[3]), s); s = fmaf(xhi.w, hi_bf(w[3]), s);
            acc[m][r] = s;
          }
      }
    }
    if (gi % gpt != gpt - 1) return;
    // last chunk group of the tile: reduce across the wave and run the epilogue
    const int tile = bid + (gi / gpt) * nblk;
    const int row0 = tile * ROWS_PER_TILE + wave * R;
#pragma unroll
    for (int m = 0; m < NW; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[m][r] = wave_sum(acc[m][r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
        if (row < n_out) {
          const float lin = rbf(acc[0][r]);  // candle_nn::Linear output tensor (bf16)
          if (EPI == GEMV_PARTIAL_F32) {
            a.y_f32[row] = acc[0][r];
          } else if (EPI == GEMV_STORE) {
            act_st_bf<COH>((bf16_t*)a.y + row, f2bf(lin));
          } else if (EPI == GEMV_RESIDUAL) {
            act_st_bf<COH>((bf16_t*)a.y + row, f2bf(bf2f(FAST ? resv[r] : act_ld_bf<COH>((const bf16_t*)a.residual + row)) + lin));
          } else if (EPI == GEMV_SILU_MUL) {
            const float g = rbf(silu_f(lin));            // gate_proj -> act_fn   (modules.rs:82)
            const float up = rbf(acc[NW - 1][r]);        // up_proj               (modules.rs:83)
            act_st_bf<COH>((bf16_t*)a.y + row, f2bf(g * up));  // lhs * rhs             (modules.rs:84)
          } else {  // GEMV_LOGITS: logits tensor is bf16 in the reference, read back as f32 (generate.rs:75)
            a.y_f32[row] = lin;
            if (lin > tile_best || (lin == tile_best && (uint32_t)row < tile_best_i)) { tile_best = lin; tile_best_i = row; }
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < NW; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
  };

  stamp(2);
  if (FAST) {
    int g = 0;
    for (; g + 2 < ngroups; g += 2) {   // steady state: both refills unconditional
      __builtin_amdgcn_sched_barrier(0);
      issue(g + 1, bufB, resB);
      __builtin_amdgcn_sched_barrier(0);
      consume(g, bufA, resA);
      __builtin_amdgcn_sched_barrier(0);
      issue(g + 2, bufA, resA);
      __builtin_amdgcn_sched_barrier(0);
      consume(g + 1, bufB, resB);
    }
    if (g + 1 < ngroups) {
      __builtin_amdgcn_sched_barrier(0);
      issue(g + 1, bufB, resB);
      __builtin_amdgcn_sched_barrier(0);
      consume(g, bufA, resA);
      consume(g + 1, bufB, resB);
    } else if (g < ngroups) {
      consume(g, bufA, resA);
    }
  } else {
    for (int g = 0; g < ngroups; g += 2) {
      if (g + 1 < ngroups) issue(g + 1, bufB, resB);
      consume(g, bufA, resA);
      if (g == 0) stamp(3);
      if (g + 2 < ngroups) issue(g + 2, bufA, resA);
      if (g + 1 < ngroups) consume(g + 1, bufB, resB);
    }
  }
  stamp(4);
  if (EPI == GEMV_LOGITS) {
    // per-block argmax partial: 4 wave leaders -> slot bid
    __syncthreads();
    if (lane == 0) { red[wave] = tile_best; reinterpret_cast<uint32_t*>(red)[4 + wave] = tile_best_i; }
    __syncthreads();
    if (tid == 0) {
      float bv = red[0];
      uint32_t bi = reinterpret_cast<uint32_t*>(red)[4];
      for (int w = 1; w < 4; ++w) {
        const float v = red[w];
        const uint32_t i = reinterpret_cast<uint32_t*>(red)[4 + w];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
      }
      a.blk_max[bid] = bv;
      a.blk_idx[bid] = bi;
    }
  }
}

}  // namespace aha
