// Prefill / ViT projection GEMM on the matrix cores:
//
//   C[M,N] = A[M,K] . W[N,K]^T  (+ bias[N]) (act) (+ residual[M,N])      bf16 in / bf16 out, f32 accumulate
//
// Both operands are K-contiguous (activations row-major, candle_nn::Linear weights (out,in) row-major), which is the
// natural MFMA feed on CDNA: every fragment is a 16-byte run of one row.  v_mfma_f32_16x16x32_bf16, 128x128x64 block
// tile, 4 waves (2x2, 64x64 each), XOR-swizzled 16-byte LDS slots (conflict-free ds_read_b128).
// Staging:
//   gemm_glds_kernel: global_load_lds_dwordx4 straight into a 32-KiB LDS tile (no VGPR round trip, no
//       ds_write pass); the swizzle is applied on the per-lane SOURCE address because the DMA destination is
//       lane-linear; 32 KiB LDS and ~110 VGPRs let 4 blocks share a CU, which is what overlaps load and MFMA.
// The MFMA is issued as W-fragment x A-fragment so that each lane ends up with 4 consecutive output columns of one row:
// bias / activation / gate*up pairing / residual are then lane-local and the store is 8 bytes.
// Rounding points follow the reference op boundaries (Linear matmul -> bf16, + bias -> bf16, act -> bf16, + residual -> bf16).
#include <math.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "gemm256_body.h"
#include "kernels.h"

namespace aha {

// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel: "once" flags are one bit per device id (a process may drive several
// GPUs through the device= argument of the Python API).
struct DevOnce {
  std::atomic<unsigned long long> mask[4] = {};   // device ids 0..255, one bit each (not folded: device 64 is not device 0)
  std::mutex mu;
  // `if (auto g = once.first()) { hipFuncSetAttribute(...); }`: the guard holds the mutex for the whole if statement and publishes the
  // device's bit in its destructor, AFTER the attribute calls -- a second thread on the same device either sees the bit (attributes in
  // place) or waits for the mutex (round-5 advisor: the bit used to be set before the caller ran hipFuncSetAttribute, so a concurrent
  // launch of a > 64 KiB dynamic-LDS kernel could slip in between and fail).
  struct Guard {
    DevOnce* o;
    int dev;
    bool need;
    std::unique_lock<std::mutex> lk;
    explicit operator bool() const { return need; }
    ~Guard() {
      if (need && dev >= 0 && dev < 256) o->mask[dev >> 6].fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
  };
  Guard first() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
    auto seen = [&] { return dev >= 0 && dev < 256 && (mask[dev >> 6].load(std::memory_order_acquire) >> (dev & 63) & 1); };
    if (seen()) return Guard{this, dev, false, {}};
    std::unique_lock<std::mutex> lk(mu);
    return Guard{this, dev, !seen(), std::move(lk)};   // (ids >= 256: set the attribute every time, never a stale "done")
  }
};

namespace {

// ---- variant 1: direct-to-LDS staging -------------------------------------------------------------------------------
// Per K tile each wave issues 4 + 4 global_load_lds_dwordx4 (1 KiB each: 8 rows x 128 B of the LDS image).  Lane i of
// round j writes LDS row (j*4+wave)*8 + i/8, slot position i%8; it therefore READS logical slot (i%8) ^ f(row) from
// global memory (source-side swizzle), and the fragment reads apply the same XOR.  K tails / nothing-to-load lanes
// point at a 16-byte zero block.
template <int ACT, bool HAS_BIAS, bool HAS_RES, bool ROWS = false>   // ROWS: the row-order epilogue (gemm256_body.h epilogue16_rows)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void gemm_glds_kernel(GemmArgs a, const void* zeros) {   // three blocks per CU: <= 168 VGPRs
  char* const smem = gemm_smem;  // [A tile | W tile]
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
  // (every wave is past the k loop's last __syncthreads: the tile buffers are free for the row-order epilogue's bands)
  if constexpr (ROWS) epilogue16_rows<ACT, HAS_BIAS, HAS_RES>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, smem + wave * (32 * 144));
  else epilogue<ACT, HAS_BIAS, HAS_RES, 4>(a, acc, m0 + wm * 64, n0 + wn * 64, G, c);
}

// ---- variant 1b: the same 128 x 128 tile with a RING of LDS stages (round 6) ---------------------------------------------------------
// gemm_glds_kernel stages a K tile, drains, multiplies, and relies on the three or four blocks a CU holds to hide each other's memory
// round trip.  A launch of <= 256 blocks has ONE block per CU: every k step then costs the whole trip of its own tile -- 1.4 us per k step
// on cold weights, where the multiply is 0.21 us (BASELINE cfg 4, Qwen3-ASR: 187 GEMMs of 390 / 406 rows per prefill ran this way, 21-27 us
// each for 2.5-5 GFLOP).  Here the block prefetches for itself: NST stages of [A tile | W tile], tile kt + NST - 1 requested while tile kt
// is multiplied, its pieces issued between the MFMA groups, ONE counted s_waitcnt vmcnt and ONE raw barrier per k step (a __syncthreads
// would drain the DMA queue).  Tiles past the end are requested from the zero block so that the counts stay uniform.  Same tile, same k
// order per accumulator, same epilogue chain: bit-identical to gemm_glds_kernel.
// The block is WM x WN waves, each 64 rows x (128 / WN) columns:
//   WM 2, WN 4 = the 128 x 128 tile on EIGHT waves of 64 x 32 (two per SIMD: while one wave of a SIMD is held in the issue of its LDS-DMA
//                pieces -- ~60 cycles each, the MFMA pipe idle behind a lone wave -- the other one multiplies; four waves of 64 x 64 cost
//                0.49 us per k step, cfg 4 prefill 4.42 against 4.12 ms same box); four stages of 32 KiB;
//   WM 4, WN 2 = a 256 x 128 tile on eight waves of 64 x 64, three stages of 48 KiB: for launches whose 128^2 tiling overflows one block
//                per CU where 256 x 128 tiles do not (ViT proj: 288 -> 144 blocks; cfg 2 qkv: 512 -> 256).
// A stage is [A tile: WM x 8 KiB | W tile: 16 KiB] in 8-row pieces of 1 KiB; wave w stages pieces j * NW + w.
template <int ACT, bool HAS_BIAS, bool HAS_RES, int NST, bool ROWS = true, int WM = 2, int WN = 2>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(2))) void gemm_glds_ring_kernel(GemmArgs a, const void* zeros, int kt_per_slice) {
  constexpr int NW = WM * WN, A_BYTES = WM * 64 * BK * 2, STAGE = A_BYTES + TILE_BYTES;
  constexpr int NI = 8 / WN;              // 16-column fragments of a wave's 64 x (128 / WN) sub-tile
  constexpr int NPA = WM * 8 / NW, NPW = 16 / NW;   // 8-row pieces of A / W per wave and k step
  constexpr int NPJ = NPA > NPW ? NPA : NPW;        // piece slots j = 0 .. NPJ - 1 (slot j: A piece j if j < NPA, W piece j if j < NPW)
  constexpr int LPT = NPA + NPW;                    // LDS-DMA loads per wave and K tile
  constexpr int FH = (NPA < NPJ / 2 ? NPA : NPJ / 2) + (NPW < NPJ / 2 ? NPW : NPJ / 2);   // ... of them in the first half of a k step
  static_assert(NPJ == 2 || NPJ == 4, "piece slots per half: 1 or 2");
  char* const smem = gemm_smem;  // NST x [A tile | W tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, c = lane & 15;
  const int wm = wave / WN, wn = wave % WN;
  int m0, n0;
  tile_of_block<WM * 64, BN>(a, m0, n0);
  const bf16_t* A = (const bf16_t*)a.A;
  const bf16_t* W = (const bf16_t*)a.W;
  // gridDim.y > 1 = split-K (ACT_PARTIAL_F32 only): slice z owns k tiles [z * kt_per_slice, ...) and writes the f32 slab a.C + z*M*ldc
  int kt0 = 0, nk = (a.K + BK - 1) / BK;
  if constexpr (ACT == ACT_PARTIAL_F32) {
    kt0 = (int)blockIdx.y * kt_per_slice;
    nk = min(nk, kt0 + kt_per_slice);
    a.C = (float*)a.C + (int64_t)blockIdx.y * a.M * a.ldc;
  }
  const bf16_t* ga[NPA];
  const bf16_t* gw[NPW];
  int kofs[NPJ];   // (the k slot a lane fetches depends on its row inside the 8-row piece and the piece's parity: the same for A and W)
#pragma unroll
  for (int j = 0; j < NPJ; ++j) {
    const int row = (j * NW + wave) * 8 + (lane >> 3);
    const int sl = (lane & 7) ^ ((row >> 1) & 7);  // logical k-slot this lane fetches
    kofs[j] = sl * 8;
    if (j < NPA) ga[j] = A + (int64_t)min(m0 + row, a.M - 1) * a.lda + sl * 8;
    if (j < NPW) gw[j] = W + (int64_t)min(n0 + row, a.N - 1) * a.ldw + sl * 8;
  }
  f32x4_t acc[NI][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#define AHA_RING_BAR()                           \
  do {                                           \
    __builtin_amdgcn_sched_barrier(0);           \
    __builtin_amdgcn_s_barrier();                \
    __builtin_amdgcn_sched_barrier(0);           \
  } while (0)
  // piece j of tile kt into stage st_: 8 rows of A and (j < NPW) 8 rows of W.  The pieces of a tile ride between the MFMAs of the tile being
  // multiplied (a piece costs ~60 cycles of issue among MFMAs, 100-185 in a burst next to fragment reads)
  auto stage_piece = [&](int kt, int st_, int j) __attribute__((always_inline)) {
    char* sa = smem + st_ * STAGE;
    char* sw = sa + A_BYTES;
    const bool ok = kt < nk && kt * BK + kofs[j] < a.K;
    if (j < NPA) {
      const void* pa = ok ? (const void*)(ga[j < NPA ? j : 0] + kt * BK) : zeros;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)pa, (lds_ptr_t)(sa + (j * NW + wave) * 1024), 16, 0, 0);
    }
    if (j < NPW) {
      const void* pw = ok ? (const void*)(gw[j < NPW ? j : 0] + kt * BK) : zeros;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)pw, (lds_ptr_t)(sw + (j * NW + wave) * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
#pragma unroll
    for (int j = 0; j < NPJ; ++j) stage_piece(kt0 + t, t, j);
  // One wave per SIMD (WM = 2): nothing else hides a fragment read's LDS round trip, so the reads run one half tile (32 of the 64 k) AHEAD of
  // the MFMAs, in two register sets -- F1 = (kt, k 32-63) is requested among the MFMAs on F0 = (kt, k 0-31), F0 = (kt + 1, k 0-31) in
  // front of the MFMAs on F1.  ONE barrier per k step, between the two halves: in front of it a wave waits for its own pieces of tile
  // kt + 1 (counted vmcnt: the NST - 3 newer tiles and the first half of the tile being requested stay in flight) and for its LDS reads
  // (lgkmcnt(0): they were issued 16 MFMAs ago), so past the barrier tile kt + 1 is whole and nobody reads tile kt - 1 any more -- which is
  // the stage the pieces of tile kt + NST - 1 go to.  (Reading first and multiplying after cost 0.52 us per k step for 0.21 us of MFMAs.)
  auto read_frags = [&](int st_, int ks, bf16x8_t (&af)[4], bf16x8_t (&wf)[NI]) __attribute__((always_inline)) {
    const char* sa = smem + st_ * STAGE;
    const char* sw = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i] = as_frag(*reinterpret_cast<const u32x4_t*>(sa + swz(wm * 64 + i * 16 + c, ks * 4 + G)));
      if (i < NI) wf[i] = as_frag(*reinterpret_cast<const u32x4_t*>(sw + swz(wn * (NI * 16) + i * 16 + c, ks * 4 + G)));
    }
  };
  // piece slot jh (of this half's NPJ / 2) goes behind MFMA group g when g == (jh + 1) * NI / (NPJ / 2) - 1
  auto pieces_behind = [&](int g, int kt_new, int st_new_, int jbase) __attribute__((always_inline)) {
#pragma unroll
    for (int jh = 0; jh < NPJ / 2; ++jh)
      if (g == (jh + 1) * NI / (NPJ / 2) - 1) {
        __builtin_amdgcn_sched_barrier(0);
        stage_piece(kt_new, st_new_, jbase + jh);
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  bf16x8_t a0[4], w0[NI], a1[4], w1[NI];
  {
    constexpr int KEEP0 = LPT * (NST - 2);
    __builtin_amdgcn_s_waitcnt(0x0F70 | (KEEP0 & 15) | ((KEEP0 >> 4) << 14));
    AHA_RING_BAR();
    read_frags(0, 0, a0, w0);
  }
  int st = 0, st_new = NST - 1;
  for (int kt = kt0; kt < nk; ++kt) {
    const int st_next = (st + 1 == NST) ? 0 : st + 1;
    // (F1 is requested after the first four MFMAs on F0, not in front of them: the compiler's wait for F0 is an lgkmcnt(0), and with F1
    // already in the queue it would wait for both)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16(w0[ni], a0[mi], acc[ni][mi]);
      if (ni == 0) {
        __builtin_amdgcn_sched_barrier(0);
        read_frags(st, 1, a1, w1);
        __builtin_amdgcn_sched_barrier(0);
      }
      pieces_behind(ni, kt + NST - 1, st_new, 0);
    }
    // the first half requested FH of the tile's LPT pieces; vmcnt is a 6-bit field: [3:0] and [15:14]; lgkmcnt(0) = bits [11:8]
    constexpr int KEEP = LPT * (NST - 3) + FH;
    __builtin_amdgcn_s_waitcnt(0x0070 | (KEEP & 15) | ((KEEP >> 4) << 14));
    AHA_RING_BAR();
    read_frags(st_next, 0, a0, w0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16(w1[ni], a1[mi], acc[ni][mi]);
      pieces_behind(ni, kt + NST - 1, st_new, NPJ / 2);
    }
    st_new = st;
    st = st_next;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the requests past the end still write their (zero) pieces: nothing in flight when the epilogue reuses the LDS
  AHA_RING_BAR();
#undef AHA_RING_BAR
  if constexpr (ROWS) epilogue16_rows<ACT, HAS_BIAS, HAS_RES, NI>(a, acc, m0 + wm * 64, n0 + wn * (NI * 16), lane, smem + wave * (32 * 144));
  else epilogue<ACT, HAS_BIAS, HAS_RES, 4, NI>(a, acc, m0 + wm * 64, n0 + wn * (NI * 16), G, c);
}

// ---- 256 x 256 x 64 tiles ----------------------------------------------------------------------------------------------------
// (The first 256^2 kernel -- 8 waves, one __syncthreads per K tile with the LDS-DMA drained in front of it, 16x16x32 MFMA: 1.41 us
// per K step alone on the chip, 1.03-1.06 PF at 8192^3 -- was the A/B reference of the two below through round 2 and is retired;
// its measurements are in profiles/r01_gemm_tile_ab.md and profiles/r02_gemm_anatomy.md.)
// gridDim.y > 1 = split-K: slice z accumulates k tiles [z*kps, (z+1)*kps) and writes an f32 slab a.C + z*M*ldc (ACT_PARTIAL_F32);
// gemm_splitk_reduce_kernel sums the slabs and applies the epilogue chain.

// ---- the 256 x 256 x 64 tile on eight waves as a staggered, counted-wait pipeline (gemm256p_kernel) ------------------------------
// 2 x 4 waves, 128 x 64 each; two LDS stages of [A tile | W tile], rows of 128 B with the XOR slot swizzle of the 128^2 kernel:
//   * a K tile is staged as four 16-KiB HALF-tiles (A rows 0-127 / 128-255, W rows 0-127 / 128-255), one half-tile per
//     segment (2 LDS-DMA instructions per thread), and nothing in the loop ever waits for all of them: the only wait is a
//     counted s_waitcnt vmcnt(2) once per K tile, which leaves the newest half-tile in flight (raw s_barrier -- a
//     __syncthreads() would drain the DMA queue);
//   * the wave's 128 x 64 output is computed as four 64 x 32 quadrants of 8 v_mfma_f32_32x32x16_bf16 each, fed from registers that were read
//     from LDS one segment earlier (A half 8 x ds_read_b128, W half 4 x);
//   * the two wave rows (wm = 0 / 1: one wave of each per SIMD) run the same program ONE barrier interval apart, so in every
//     interval one wave of a SIMD is in a 16-MFMA segment (s_setprio 1) while its partner reads fragments and issues DMA.
// Per K tile and wave (L = load segment, C = compute segment; G0 = waves wm 0 at intervals 8t+1.., G1 one interval later):
//   L1: read A0, W0      C1: quadrant (0,0) + DMA A-lo(t+1)
//   L2: read W1          C2: quadrant (0,1) + DMA A-hi(t+1)
//   L3: read A1          C3: quadrant (1,1) + DMA W-lo(t+2)
//   L4: vmcnt(2)         C4: quadrant (1,0) + DMA W-hi(t+2)
// (the DMA pieces are issued between the MFMAs of the compute segment: ~60 issue cycles each in the MFMAs' shadow, against
// 100-185 next to ds_reads)
// Hazards (intervals; a read issued in interval i is complete once its wave is past the lgkmcnt at the start of its next
// segment, i.e. before the barrier that ends interval i+1):
//   WAR  W(t) last read L2 (8t+3 / 8t+4, complete by 8t+5) -> W(t+2) DMA in C3 / C4 (8t+6.. ); A-lo(t) last read by G0 in L3
//        (8t+5), A-hi(t) by G1 (8t+6) -> A(t+2) DMA in C1 / C2 of tile t+1 (8t+10.. / 8t+12..).
//   RAW  tile t+1 is first read at 8t+9 (G0) / 8t+10 (G1); its W halves were issued a whole tile earlier, its A halves in C1 /
//        C2 of tile t; every wave retires them with the vmcnt(2) of its L4 (8t+7 / 8t+8), one barrier before the first read.
// Half-tiles past the last K tile are issued all the same (from the zero block, into a buffer nobody reads any more) so the
// counts stay uniform.
// TRACE (AHA_GEMM_TRACE=1, debug): block 0, waves 0 and 4 stamp s_memtime at every segment boundary of K tiles 8 and 9 into `trace`.
template <int ACT, bool HAS_BIAS, bool HAS_RES, int MODE = 0>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(GemmArgs a, const void* zeros, int kt_per_slice, unsigned long long* trace = nullptr) {
  char* const smem = gemm_smem;  // [2 stages][A tile 32 KiB | W tile 32 KiB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int m0, n0;
  tile_of_block<BM2, BN2>(a, m0, n0);
  const bf16_t* A = (const bf16_t*)a.A;
  const bf16_t* W = (const bf16_t*)a.W;
  const int nk_all = (a.K + BK - 1) / BK;
  const int kt0 = blockIdx.y * kt_per_slice, kt1 = min(nk_all, kt0 + kt_per_slice);
  if (ACT == ACT_PARTIAL_F32) a.C = (float*)a.C + (int64_t)blockIdx.y * a.M * a.ldc;

  // staging sources: half h (rows h*128..), piece j = 0,1: LDS row group (h*16 + j*8 + wave) of 8 rows (1 KiB)
  const int kofs = (((lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7))) * 8;   // logical k offset this lane fetches (source-side swizzle)
  const bf16_t* ga[2][2];
  const bf16_t* gw[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = h * 128 + (j * 8 + wave) * 8 + (lane >> 3);
      ga[h][j] = A + (int64_t)min(m0 + row, a.M - 1) * a.lda + kofs;
      gw[h][j] = W + (int64_t)min(n0 + row, a.N - 1) * a.ldw + kofs;
    }
  auto issue_half = [&](int kt, int stage, int which /* 0 = A, 1 = W */, int h) {
    if (MODE == 2 && kt > kt0 + 1) return;   // ablation: no DMA in the steady state
    char* dst = smem + stage * 2 * TILE2_BYTES + which * TILE2_BYTES + h * (TILE2_BYTES / 2);
    const bool ok = kt < kt1 && kt * BK + kofs < a.K;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const void* src = ok ? (const void*)((which ? gw[h][j] : ga[h][j]) + (int64_t)kt * BK) : zeros;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(dst + (j * 8 + wave) * 1024), 16, 0, 0);
    }
  };

  f32x16_t acc[2][4];  // [n fragment of 32][m fragment of 32]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

// 8 MFMAs (32x32x16) of one 64 x 32 quadrant (W fragments WF, accumulator columns NI0.., rows MI0..) with the segment's half-tile DMA
// issued in the shadow of the first MFMAs (an LDS-DMA piece costs ~60 issue cycles among MFMAs, 100-185 among ds_reads)
#define MMA_QUAD(WF, NF, MF0, DMA)                                                                                    \
  do {                                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                \
      _Pragma("unroll") for (int mf = 0; mf < 2; ++mf) {                                                            \
        if (!FULL && MF0 + mf >= nmf) continue;                                                                      \
        if (MODE != 3) acc[NF][MF0 + mf] = mfma32(WF[ks], af[ks][mf], acc[NF][MF0 + mf]);                           \
        else asm volatile("" :: "v"(WF[ks]), "v"(af[ks][mf]));                                                     \
      }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    DMA;                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    _Pragma("unroll") for (int ks = 2; ks < 4; ++ks)                                                                \
      _Pragma("unroll") for (int mf = 0; mf < 2; ++mf) {                                                            \
        if (!FULL && MF0 + mf >= nmf) continue;                                                                      \
        if (MODE != 3) acc[NF][MF0 + mf] = mfma32(WF[ks], af[ks][mf], acc[NF][MF0 + mf]);                           \
        else asm volatile("" :: "v"(WF[ks]), "v"(af[ks][mf]));                                                     \
      }                                                                                                              \
  } while (0)

#define AHA_BAR()                                \
  do {                                           \
    __builtin_amdgcn_sched_barrier(0);           \
    __builtin_amdgcn_s_barrier();                \
    __builtin_amdgcn_sched_barrier(0);           \
  } while (0)

  if (MODE == 1 && blockIdx.x == 0 && tid == 0) { trace[40] = __builtin_readcyclecounter(); trace[41] = wall_clock64(); }
  // prologue: tile kt0 complete + the W halves of tile kt0+1 in flight
  issue_half(kt0, 0, 1, 0);
  issue_half(kt0, 0, 1, 1);
  issue_half(kt0, 0, 0, 0);
  issue_half(kt0, 0, 0, 1);
  issue_half(kt0 + 1, 1, 1, 0);
  issue_half(kt0 + 1, 1, 1, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  AHA_BAR();
  if (wm == 1) AHA_BAR();   // the second wave row runs one interval behind

  bf16x8_t af[4][2], wf0[4], wf1[4];   // [k step of 16][32-row fragment]
  const int r32 = lane & 31, hk = lane >> 5;
  // Ragged last row tile (M = 1542: the seventh tile holds 6 rows): a wave skips the MFMAs and fragment reads of its 32-row
  // fragments that lie entirely past M (wave-uniform), so such a tile costs its DMA + barrier skeleton, not a full tile.
  const int nmf = __builtin_amdgcn_readfirstlane(max(0, min(4, (a.M - (m0 + wm * 128) + 31) / 32)));
  int tr_i = 0;
  auto stamp = [&](int kt) {
    if (MODE == 1) {
      if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && (kt == kt0 + 8 || kt == kt0 + 9) && tr_i < 20)
        trace[(wave >> 2) * 20 + tr_i++] = __builtin_readcyclecounter();
    }
  };
  auto k_loop = [&](auto full_tag) {
  constexpr bool FULL = decltype(full_tag)::value;   // all four 32-row fragments of this wave hold valid rows: no predicates
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    stamp(kt);
    const char* sa = smem + cur * 2 * TILE2_BYTES;
    const char* sw = sa + TILE2_BYTES;
    // ---- L1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE != 4 || kt == kt0) wf0[ks] = as_frag(*reinterpret_cast<const u32x4_t*>(sw + swz(wn * 64 + r32, ks * 2 + hk)));
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if ((MODE != 4 || kt == kt0) && (FULL || i < nmf)) af[ks][i] = as_frag(*reinterpret_cast<const u32x4_t*>(sa + swz(wm * 128 + i * 32 + r32, ks * 2 + hk)));
    }
    stamp(kt);
    AHA_BAR();
    stamp(kt);
    // ---- C1: quadrant (0,0)
    __builtin_amdgcn_s_setprio(1);
    MMA_QUAD(wf0, 0, 0, issue_half(kt + 1, cur ^ 1, 0, 0));
    __builtin_amdgcn_s_setprio(0);
    stamp(kt);
    AHA_BAR();
    stamp(kt);
    // ---- L2
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      if (MODE != 4 || kt == kt0) wf1[ks] = as_frag(*reinterpret_cast<const u32x4_t*>(sw + swz(wn * 64 + 32 + r32, ks * 2 + hk)));
    stamp(kt);
    AHA_BAR();
    stamp(kt);
    // ---- C2: quadrant (0,1)
    __builtin_amdgcn_s_setprio(1);
    MMA_QUAD(wf1, 1, 0, issue_half(kt + 1, cur ^ 1, 0, 1));
    __builtin_amdgcn_s_setprio(0);
    stamp(kt);
    AHA_BAR();
    stamp(kt);
    // ---- L3
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if ((MODE != 4 || kt == kt0) && (FULL || 2 + i < nmf)) af[ks][i] = as_frag(*reinterpret_cast<const u32x4_t*>(sa + swz(wm * 128 + 64 + i * 32 + r32, ks * 2 + hk)));
    stamp(kt);
    AHA_BAR();
    stamp(kt);
    // ---- C3: quadrant (1,1)
    __builtin_amdgcn_s_setprio(1);
    MMA_QUAD(wf1, 1, 2, issue_half(kt + 2, cur, 1, 0));
    __builtin_amdgcn_s_setprio(0);
    stamp(kt);
    AHA_BAR();
    stamp(kt);
    // ---- L4
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // A(t+1) has landed (its W halves long before); W-lo(t+2) stays in flight
    stamp(kt);
    AHA_BAR();
    stamp(kt);
    // ---- C4: quadrant (1,0)
    __builtin_amdgcn_s_setprio(1);
    MMA_QUAD(wf0, 0, 2, issue_half(kt + 2, cur, 1, 1));
    __builtin_amdgcn_s_setprio(0);
    AHA_BAR();
  }
  };
  if (nmf == 4) k_loop(std::true_type{});
  else k_loop(std::false_type{});
  if (wm == 0) AHA_BAR();   // arrivals of the two wave rows balance
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (dummy) half-tiles
#undef AHA_BAR
#undef MMA_QUAD
  if (MODE == 1 && blockIdx.x == 0 && tid == 0) { trace[42] = __builtin_readcyclecounter(); trace[43] = wall_clock64(); }
  if (ACT == ACT_PARTIAL_F32 && a.partial_rows) {   // f32 slabs in row order (gemm256_body.h epilogue32_rows_f32): 256-byte row segments
    __syncthreads();   // every wave has left the k loop: the stages are free
    epilogue32_rows_f32<64, 2>(a, acc, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * (32 * 272));
  } else {
    epilogue32<ACT, HAS_BIAS, HAS_RES, 2, 4>(a, acc, m0 + wm * 128, n0 + wn * 64, lane);
  }
}

// ---- variant 5: the 256 x 256 x 64 tile on FOUR waves (gemm256_body.h gemm256q_mainloop: schedule, LDS plan, counted waits) ---------
// ROW5 (gemm256_body.h): rows5 = M mod 256 (1..32) rows ride as a fifth fragment row of the last row tile; the grid has M / 256 row tiles.
// GRP (launch_gemm_grouped): row groups -- GemmArgs::groups segments of M rows, the tile grid runs over groups x ceil(M / 256) row tiles.
template <int ACT, bool HAS_BIAS, bool HAS_RES, bool BAR2 = true, int ABL = 0, bool NF3 = false, bool ROW5 = false, bool GRP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm256q_kernel(GemmArgs a, int kt_per_slice, int rows5) {
  char* const smem = gemm_smem;  // [stage][A0 | A1 | W0 | W1] x 16 KiB
  constexpr int TN = NF3 ? 192 : 256, WC = TN / 2;   // tile columns, columns per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int m0, n0;
  if (ROW5) {
    GemmArgs at = a;
    at.M = a.M - rows5;   // the tiling covers the multiple of 256; the last row tile owns the rest
    tile_of_block<BM2, TN>(at, m0, n0);
  } else if (GRP) {
    // row groups (GemmArgs::groups): a tile never straddles two segments
    const int seg = (a.M + BM2 - 1) / BM2 * BM2;
    GemmArgs at = a;
    at.M = a.groups * seg;
    tile_of_block<BM2, TN>(at, m0, n0);
    const int g = m0 / seg;
    m0 -= g * seg;
    const int crow = a.c_row0 + g * a.c_gstride;
    a.A = (const bf16_t*)a.A + (int64_t)g * a.a_gstride * a.lda;
    a.C = (bf16_t*)a.C + (int64_t)crow * a.ldc;
    a.M = max(0, min(a.M, a.m_total - crow));
    if (m0 >= a.M) return;   // (block-uniform: a segment past the end of the sequence)
  } else {
    tile_of_block<BM2, TN>(a, m0, n0);
  }
  const int rows5_here = ROW5 && m0 + 256 + rows5 == a.M ? rows5 : 0;
  // Claim the SIMD's whole register file (512 VGPRs: v255 and a255 count as used).  The kernel runs one wave per SIMD by design, so
  // the claim costs nothing -- but the 192-column variant needs only 424 registers, which left room for a wave of ANOTHER kernel
  // (one running on a concurrent stream) on the same SIMD, and on gfx950 such co-residents were measured to compute
  // `v_pk_add_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]` with the wrong half of vB in lanes 16-31 / 48-63: the RMSNorm of a
  // neighbouring stream lost a term of its sum of squares in ~20% of launches (profiles/r03_simd_coresidency.md).  With the claim
  // nothing else is placed on this kernel's CUs.  tests/test_isa_cpu.py holds the 512, tests/test_ops_gpu.py::
  // test_small_kernels_beside_a_gemm_on_another_stream runs the two-stream case.
  asm volatile("" ::: "v255", "a255");
  const int nk_all = a.K / BK;
  const int kt0 = blockIdx.y * kt_per_slice, kt1 = min(nk_all, kt0 + kt_per_slice);
  if (ACT == ACT_PARTIAL_F32) a.C = (float*)a.C + (int64_t)blockIdx.y * a.M * a.ldc;
  f32x16_t acc[4][4];  // [n fragment of 32][m fragment of 32]
  f32x16_t acc5[3];    // ROW5: the fifth fragment row of the wm = 1 waves of the last row tile
  gemm256q_mainloop<ACT, HAS_BIAS, HAS_RES, BAR2, ABL, NF3, ROW5>(a, m0, n0, kt0, kt1, lane, wave, acc, 0.f, rows5_here, acc5);
  if (ACT == ACT_PARTIAL_F32) {
    __syncthreads();   // every wave has left the k loop (and waited for its own DMAs): the stages are free
    if (a.partial_rows) epilogue32_rows_f32<WC>(a, acc, m0 + wm * 128, n0 + wn * WC, lane, smem + wave * (32 * 528));
    else epilogue32<ACT, HAS_BIAS, HAS_RES, 4, 4, WC / 32>(a, acc, m0 + wm * 128, n0 + wn * WC, lane);
  } else {
    __syncthreads();   // every wave has left the k loop (and waited for its own DMAs): the stages are free
    epilogue32_rows<ACT == ACT_PARTIAL_F32 ? ACT_NONE : ACT, HAS_BIAS, HAS_RES, WC>(a, acc, m0 + wm * 128, n0 + wn * WC, lane, smem + wave * 8448);
    if (ROW5 && rows5_here > 0 && wm == 1)
      epilogue32_band<ACT == ACT_PARTIAL_F32 ? ACT_NONE : ACT, HAS_BIAS, HAS_RES, WC>(a, acc5[0], acc5[1], acc5[2], acc5[2], m0 + 256, n0 + wn * WC, lane,
                                                                                      smem + wave * 8448);
  }
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

// The reduce pass with the RMSNorm of the finished rows folded in (prefill: o_proj / down_proj + residual, then the next
// RMSNorm, /root/reference/src/models/qwen3/model.rs:79-90): one wave per row, lane L owns the 8-column vectors i * 64 + L exactly
// as rmsnorm_rows_kernel does, so the sum of squares is accumulated and reduced in the same order and the normalised rows are
// bit-identical to gemm_splitk_reduce_kernel followed by rmsnorm_rows_kernel -- one launch and one read of the row less.
// N = VPL * 512 (no predicated loads); ACT_NONE, no bias.  All slab loads of a row are issued before the first store.
template <bool HAS_RES, int VPL>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_norm_kernel(GemmArgs a, const float* __restrict__ slabs, int nsl) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const int64_t slab = (int64_t)a.M * a.N;
  const float* sp = slabs + row * a.N;
  float4 s[VPL][2];
  u32x4_t wv[VPL];   // the norm weights: requested first, consumed last (behind the stores they would otherwise be serialised, one
#pragma unroll       // load -> wait -> store round trip per vector: the compiler cannot move a load across a store it may alias)
  for (int i = 0; i < VPL; ++i) wv[i] = ld16((const bf16_t*)a.norm_w + (i * 64 + lane) * 8);
  u32x4_t rv[VPL];   // likewise the residual row (it is the output row: read here, overwritten at the end by the same lane)
  if (HAS_RES) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) rv[i] = ld16((const bf16_t*)a.residual + row * a.ldc + (i * 64 + lane) * 8);
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int n = (i * 64 + lane) * 8;
    s[i][0] = *reinterpret_cast<const float4*>(sp + n);
    s[i][1] = *reinterpret_cast<const float4*>(sp + n + 4);
  }
  for (int z = 1; z < nsl; ++z) {   // a whole slab row in flight before the first add (slice order: the reduce kernel's sums)
    float4 t[VPL][2];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int n = (i * 64 + lane) * 8;
      t[i][0] = *reinterpret_cast<const float4*>(sp + z * slab + n);
      t[i][1] = *reinterpret_cast<const float4*>(sp + z * slab + n + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      s[i][0].x += t[i][0].x; s[i][0].y += t[i][0].y; s[i][0].z += t[i][0].z; s[i][0].w += t[i][0].w;
      s[i][1].x += t[i][1].x; s[i][1].y += t[i][1].y; s[i][1].z += t[i][1].z; s[i][1].w += t[i][1].w;
    }
  }
  u32x4_t v[VPL];
  float ss = 0.f;
  bf16_t* C = (bf16_t*)a.C + row * a.ldc;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    float x[8] = {rbf(s[i][0].x), rbf(s[i][0].y), rbf(s[i][0].z), rbf(s[i][0].w), rbf(s[i][1].x), rbf(s[i][1].y), rbf(s[i][1].z), rbf(s[i][1].w)};
    if (HAS_RES) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[2 * j] += lo_bf(rv[i][j]); x[2 * j + 1] += hi_bf(rv[i][j]); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = pack_bf(x[2 * j], x[2 * j + 1]);
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    *reinterpret_cast<u32x4_t*>(C + (i * 64 + lane) * 8) = v[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p = lo_bf(v[i][j]), q = hi_bf(v[i][j]);
      ss += p * p + q * q;
    }
  }
  ss = wave_sum(ss);
  const float rinv = 1.0f / sqrtf(ss / (float)a.N + a.norm_eps);
  bf16_t* Y = (bf16_t*)a.norm_out + row * a.N;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int n = (i * 64 + lane) * 8;
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf(lo_bf(v[i][j]) * rinv * lo_bf(wv[i][j]), hi_bf(v[i][j]) * rinv * hi_bf(wv[i][j]));
    *reinterpret_cast<u32x4_t*>(Y + n) = o;
  }
}
template <bool HAS_RES>
bool launch_reduce_norm(const GemmArgs& a, const float* slabs, int nsl, hipStream_t st) {
  const dim3 grid((unsigned)((a.M + 3) / 4)), block(256);
  switch (a.N / 512) {
    case 2: hipLaunchKernelGGL((gemm_splitk_reduce_norm_kernel<HAS_RES, 2>), grid, block, 0, st, a, slabs, nsl); return true;
    case 4: hipLaunchKernelGGL((gemm_splitk_reduce_norm_kernel<HAS_RES, 4>), grid, block, 0, st, a, slabs, nsl); return true;
    case 5: hipLaunchKernelGGL((gemm_splitk_reduce_norm_kernel<HAS_RES, 5>), grid, block, 0, st, a, slabs, nsl); return true;
    case 8: hipLaunchKernelGGL((gemm_splitk_reduce_norm_kernel<HAS_RES, 8>), grid, block, 0, st, a, slabs, nsl); return true;
    case 10: hipLaunchKernelGGL((gemm_splitk_reduce_norm_kernel<HAS_RES, 10>), grid, block, 0, st, a, slabs, nsl); return true;
    default: return false;
  }
}

// The reduce pass with the LAYERNORM of the finished rows folded in (the ViT block: fc2 + bias + residual, then norm1 of the next block
// -- /root/reference/src/models/qwen3vl/model.rs:346-370): one wave per row, lane L owns the 8-column vectors i * 64 + L as
// layernorm_rows_kernel does (kernels_vit.hip), the element chain is gemm_splitk_reduce_kernel's (slab sum in slice order -> bf16, + bias ->
// bf16, + residual -> bf16), the statistics run over the stored bf16 values in layernorm_rows_kernel's order: the normalised rows are
// bit-identical to the two launches they replace (11.4 + 6.4 -> 14.3 us per block at cfg 3; round-5 verdict, weak #5).  N a multiple of 8
// and <= VPL * 512 (predicated on the vector index: the ViT's 1152 = 2.25 x 512).
template <bool HAS_BIAS, bool HAS_RES, int VPL>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_layernorm_kernel(GemmArgs a, const float* __restrict__ slabs, int nsl) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const int64_t slab = (int64_t)a.M * a.N;
  const float* sp = slabs + row * a.N;
  const int nvec = a.N >> 3;
  float4 s[VPL][2];
  u32x4_t wv[VPL], bnv[VPL], bv[VPL], rv[VPL];   // norm weight / norm bias / Linear bias / residual row: requested first, consumed last
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
      wv[i] = ld16((const bf16_t*)a.norm_w + vi * 8);
      bnv[i] = ld16((const bf16_t*)a.norm_b + vi * 8);
      if (HAS_BIAS) bv[i] = ld16((const bf16_t*)a.bias + vi * 8);
      if (HAS_RES) rv[i] = ld16((const bf16_t*)a.residual + row * a.ldc + vi * 8);
      s[i][0] = *reinterpret_cast<const float4*>(sp + vi * 8);
      s[i][1] = *reinterpret_cast<const float4*>(sp + vi * 8 + 4);
    }
  }
  for (int z = 1; z < nsl; ++z) {
    float4 t[VPL][2];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = i * 64 + lane;
      if (vi < nvec) {
        t[i][0] = *reinterpret_cast<const float4*>(sp + z * slab + vi * 8);
        t[i][1] = *reinterpret_cast<const float4*>(sp + z * slab + vi * 8 + 4);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = i * 64 + lane;
      if (vi < nvec) {
        s[i][0].x += t[i][0].x; s[i][0].y += t[i][0].y; s[i][0].z += t[i][0].z; s[i][0].w += t[i][0].w;
        s[i][1].x += t[i][1].x; s[i][1].y += t[i][1].y; s[i][1].z += t[i][1].z; s[i][1].w += t[i][1].w;
      }
    }
  }
  float f[VPL][8];
  float sum = 0.f;
  bf16_t* C = (bf16_t*)a.C + row * a.ldc;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
      float x[8] = {rbf(s[i][0].x), rbf(s[i][0].y), rbf(s[i][0].z), rbf(s[i][0].w), rbf(s[i][1].x), rbf(s[i][1].y), rbf(s[i][1].z), rbf(s[i][1].w)};
      if (HAS_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[2 * j] = rbf(x[2 * j] + lo_bf(bv[i][j])); x[2 * j + 1] = rbf(x[2 * j + 1] + hi_bf(bv[i][j])); }
      }
      if (HAS_RES) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[2 * j] += lo_bf(rv[i][j]); x[2 * j + 1] += hi_bf(rv[i][j]); }
      }
      u32x4_t v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = pack_bf(x[2 * j], x[2 * j + 1]);
      *reinterpret_cast<u32x4_t*>(C + vi * 8) = v;
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[i][2 * j] = lo_bf(v[j]); f[i][2 * j + 1] = hi_bf(v[j]); }
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += f[i][j];
    }
  }
  const float mean = wave_sum(sum) / (float)a.N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; q += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)a.N + a.norm_eps);
  bf16_t* Y = (bf16_t*)a.norm_out + row * a.N;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf((f[i][2 * j] - mean) * rstd * lo_bf(wv[i][j]) + lo_bf(bnv[i][j]), (f[i][2 * j + 1] - mean) * rstd * hi_bf(wv[i][j]) + hi_bf(bnv[i][j]));
      *reinterpret_cast<u32x4_t*>(Y + vi * 8) = o;
    }
  }
}
template <bool HAS_BIAS, bool HAS_RES>
bool launch_reduce_layernorm(const GemmArgs& a, const float* slabs, int nsl, hipStream_t st) {
  if ((a.N & 7) != 0 || a.N > 3 * 512 || (a.ldc & 7) != 0) return false;
  const dim3 grid((unsigned)((a.M + 3) / 4)), block(256);
  hipLaunchKernelGGL((gemm_splitk_reduce_layernorm_kernel<HAS_BIAS, HAS_RES, 3>), grid, block, 0, st, a, slabs, nsl);
  return true;
}

// 256 zero bytes on the CURRENT device (the K tail of the LDS-DMA kernels reads them): one block per device -- a process may drive several
// GPUs (round-4 advisor's finding on the stream-K arena, the same construct)
const void* zero_block() {
  static std::mutex mu;
  static std::map<int, void*> by_dev;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
  std::lock_guard<std::mutex> lk(mu);
  auto it = by_dev.find(dev);
  if (it != by_dev.end()) return it->second;
  void* z = nullptr;
  if (hipMalloc(&z, 256) != hipSuccess || hipMemset(z, 0, 256) != hipSuccess) (void)hipGetLastError();
  by_dev[dev] = z;
  return z;
}

// one launch of the ring kernel (WM x WN waves, NST stages); > 64 KiB of dynamic LDS needs the opt-in once per instantiation and device
template <int ACT, bool B, bool R, bool ROWS, int WM, int WN, int NST>
void launch_ring(const GemmArgs& a, dim3 grid, hipStream_t st, int kt_per_slice) {
  constexpr int LDS = NST * (WM * 64 * BK * 2 + TILE_BYTES);
  static DevOnce once;
  if (auto once_guard = once.first()) {
    hipFuncSetAttribute((const void*)gemm_glds_ring_kernel<ACT, B, R, NST, ROWS, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  }
  hipLaunchKernelGGL((gemm_glds_ring_kernel<ACT, B, R, NST, ROWS, WM, WN>), grid, dim3(WM * WN * 64), LDS, st, a, zero_block(), kt_per_slice);
}
template <int ACT, bool B, bool R>
void launch_one(const GemmArgs& a, dim3 grid, hipStream_t st, bool tall = false) {
  // <= one block per CU: the block hides its own memory round trips behind a four-stage ring (gemm_glds_ring_kernel; 128 KiB of LDS).
  // AHA_GEMM_RING=0: the single-stage kernel everywhere (A/B; bit-identical outputs)
  static const bool ring_on = [] { const char* e = getenv("AHA_GEMM_RING"); return e ? atoi(e) != 0 : true; }();
  const int64_t blocks = (int64_t)grid.x * grid.y * grid.z;
  const bool ring = ring_on && blocks <= gemm_streamk_cus() && (a.K + BK - 1) / BK >= 4;
  const int nk = (a.K + BK - 1) / BK;
  // row-order stores need whole 16-byte column groups: N a multiple of 8, rows of C / residual 16-byte aligned; else fragment order
  bool rows = false;
  if constexpr (ACT != ACT_SILU_MUL_PAIRS && ACT != ACT_PARTIAL_F32)
    rows = a.partial_rows && (a.N & 7) == 0 && (a.ldc & 7) == 0 && ((uintptr_t)a.C & 15) == 0 && (!R || ((uintptr_t)a.residual & 15) == 0) &&
           (!B || ((uintptr_t)a.bias & 15) == 0);
  if (tall) {   // 256 x 128 tiles on eight waves, three stages of 48 KiB (plan tile 2128; `grid` counts those tiles)
    if constexpr (ACT != ACT_SILU_MUL_PAIRS && ACT != ACT_PARTIAL_F32) {
      if (rows) return launch_ring<ACT, B, R, true, 4, 2, 3>(a, grid, st, nk);
    }
    return launch_ring<ACT, B, R, false, 4, 2, 3>(a, grid, st, nk);
  }
  if (ring) {
    if constexpr (ACT != ACT_SILU_MUL_PAIRS && ACT != ACT_PARTIAL_F32) {
      if (rows) return launch_ring<ACT, B, R, true, 2, 4, 4>(a, grid, st, nk);
    }
    return launch_ring<ACT, B, R, false, 2, 4, 4>(a, grid, st, nk);
  }
  if constexpr (ACT != ACT_SILU_MUL_PAIRS && ACT != ACT_PARTIAL_F32) {
    if (rows) {
      hipLaunchKernelGGL((gemm_glds_kernel<ACT, B, R, true>), grid, dim3(256), 2 * TILE_BYTES, st, a, zero_block());
      return;
    }
  }
  hipLaunchKernelGGL((gemm_glds_kernel<ACT, B, R>), grid, dim3(256), 2 * TILE_BYTES, st, a, zero_block());
}

template <int ACT>
void launch_act(const GemmArgs& a, dim3 grid, hipStream_t st, bool tall = false) {
  if (a.bias && a.residual) launch_one<ACT, true, true>(a, grid, st, tall);
  else if (a.bias) launch_one<ACT, true, false>(a, grid, st, tall);
  else if (a.residual) launch_one<ACT, false, true>(a, grid, st, tall);
  else launch_one<ACT, false, false>(a, grid, st, tall);
}

// pass 2 of a split-K plan: the f32 slabs in a.workspace summed, then the epilogue chain -- with the norm riding on the call folded in where a
// reduce pass owns whole rows (RMSNorm: launch_reduce_norm; LayerNorm: launch_reduce_layernorm)
template <int ACT, bool B, bool R>
void launch_splitk_reduce(const GemmArgs& a, int nsl, hipStream_t st, bool* norm_fused) {
  const int64_t quads = (int64_t)a.M * (a.N >> 2);
  if (norm_fused != nullptr) *norm_fused = false;
  if (a.norm_w != nullptr && a.norm_b == nullptr && norm_fused != nullptr && ACT == ACT_NONE && !B && a.N % 512 == 0 &&
      launch_reduce_norm<R>(a, (const float*)a.workspace, nsl, st)) {
    *norm_fused = true;
    return;
  }
  if (a.norm_w != nullptr && a.norm_b != nullptr && norm_fused != nullptr && ACT == ACT_NONE &&
      launch_reduce_layernorm<B, R>(a, (const float*)a.workspace, nsl, st)) {
    *norm_fused = true;
    return;
  }
  hipLaunchKernelGGL((gemm_splitk_reduce_kernel<ACT, B, R>), dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a,
                     (const float*)a.workspace, nsl);
}

// Split-K on 128^2 tiles (round 6): tiles x slices blocks of gemm_glds_ring_kernel<ACT_PARTIAL_F32> -- at most one per CU -- write f32 slabs,
// the reduce pass above finishes.  For the few-row projections with a long K (BASELINE cfg 4: o_proj / down_proj / fc2 at 390-406 rows, 32
// tiles of 128^2): the 256^2 split plans put 32-48 blocks on the chip, 18.6 us + the reduce pass.
template <int ACT, bool B, bool R>
void launch_ring_splitk(const GemmArgs& a, int splitk, hipStream_t st, bool* norm_fused) {
  GemmArgs p = a;  // pass 1: f32 slabs [slices][M][N] in the caller's workspace
  p.C = a.workspace;
  p.ldc = a.N;
  p.bias = nullptr;
  p.residual = nullptr;
  p.act = ACT_PARTIAL_F32;
  const int nk = (a.K + BK - 1) / BK, kps = (nk + splitk - 1) / splitk, nsl = (nk + kps - 1) / kps;
  const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  launch_ring<ACT_PARTIAL_F32, false, false, false, 2, 4, 4>(p, dim3(ntm * ntn, nsl), st, kps);
  launch_splitk_reduce<ACT, B, R>(a, nsl, st, norm_fused);
}

template <int ACT>
void launch_ring_splitk_act(const GemmArgs& a, int splitk, hipStream_t st, bool* norm_fused) {
  if (a.bias && a.residual) launch_ring_splitk<ACT, true, true>(a, splitk, st, norm_fused);
  else if (a.bias) launch_ring_splitk<ACT, true, false>(a, splitk, st, norm_fused);
  else if (a.residual) launch_ring_splitk<ACT, false, true>(a, splitk, st, norm_fused);
  else launch_ring_splitk<ACT, false, false>(a, splitk, st, norm_fused);
}

// the (ACT, bias, residual) combinations that have a 192-column instantiation (gemm256q_kernel<.., NF3 = true>): gate+up and the
// plain projection -- the shapes whose tile count quantises badly at 256 columns (plan_gemm)
template <int ACT, bool B, bool R>
constexpr bool has_n192() { return !B && !R && (ACT == ACT_SILU_MUL_PAIRS || ACT == ACT_NONE); }

template <int ACT, bool B, bool R>
void launch256_one(const GemmArgs& a, int splitk, hipStream_t st, bool* norm_fused = nullptr, bool n192 = false) {
  const int ntm = (a.M + BM2 - 1) / BM2, ntn = (a.N + BN2 - 1) / BN2;
  const int nk = (a.K + BK - 1) / BK;
  const size_t lds = 4 * TILE2_BYTES;
  // (> 64 KiB of dynamic LDS needs the opt-in once per kernel instance)
  if constexpr (has_n192<ACT, B, R>()) {
    if (n192 && splitk <= 1 && a.K % BK == 0 && 256.0 * 2.0 * (double)std::max(a.lda, a.ldw) < 1.0e9) {
      static DevOnce once192;
      if (auto once_guard = once192.first()) {
        hipFuncSetAttribute((const void*)gemm256q_kernel<ACT, B, R, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)gemm256q_kernel<ACT, B, R, true, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }
      // M = 256 q + r with 1 <= r <= 32: the r rows ride as a fifth fragment row of the last row tile (one row of tiles less) -- where
      // that saves a ROUND.  A tile with a fifth row costs 1.28 x a plain one (its wm = 1 waves issue 60 MFMAs per K tile instead of 48),
      // so below one round it is slower than the ragged tiles it replaces, which run on otherwise idle CUs (qkv at M = 1542: 78.3 us
      // against 74.5 us; gate+up, 768 tiles instead of 896: 273.0 against 286.7 us -- profiles/r04_gemm_row5.md).
      static const bool row5_on = [] { const char* e = getenv("AHA_GEMM_ROW5"); return e ? atoi(e) != 0 : true; }();
      const int r5 = a.M % 256;
      const int ntn192 = (a.N + 191) / 192, cus = gemm_streamk_cus();
      if (row5_on && a.M > 256 && r5 >= 1 && r5 <= 32 && ((a.M / 256) * ntn192 + cus - 1) / cus < (ntm * ntn192 + cus - 1) / cus) {
        hipLaunchKernelGGL((gemm256q_kernel<ACT, B, R, true, 0, true, true>), dim3((a.M / 256) * ((a.N + 191) / 192)), dim3(256), lds, st, a, nk, r5);
        return;
      }
      hipLaunchKernelGGL((gemm256q_kernel<ACT, B, R, true, 0, true>), dim3(ntm * ((a.N + 191) / 192)), dim3(256), lds, st, a, nk, 0);
      return;
    }
  }
  if (splitk <= 1) {
    static const bool quad = [] { const char* e = getenv("AHA_GEMM_QUAD"); return e ? atoi(e) != 0 : true; }();
    // four waves x 128 x 128 (gemm256q_kernel).  Not for short K loops (< 32 K tiles) whose epilogue reads a residual on top of a
    // bias / GELU (ViT proj, K = 1152: 37.5 us against 35.6 us on the 128^2 kernel at 4 blocks per CU, which overlap each other's
    // prologue and epilogue).  With the row-order epilogue the other ViT projections moved here: qkv 44.8 -> 39.1 us, fc1
    // (bias + GELU) 49.1 -> 47.1 us against the 8-wave kernel (rocprofv3 averages inside bench.py, same box).
    // (its staging addresses are 32-bit offsets from the block's first row: 256 rows of an operand must span < 1 GiB)
    const bool q_addr_ok = 256.0 * 2.0 * (double)std::max(a.lda, a.ldw) < 1.0e9;
    if (quad && q_addr_ok && a.K % BK == 0 && (nk >= 32 || !R || (!B && ACT != ACT_GELU_TANH && ACT != ACT_GELU_ERF))) {
      static DevOnce onceq;
      if (auto once_guard = onceq.first()) {
        hipFuncSetAttribute((const void*)gemm256q_kernel<ACT, B, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }
#ifdef AHA_DEBUG_KERNELS   // measurement scaffolding (A/B of the barrier schedule; ablations whose results are wrong by construction): debug builds only
      static const bool bar2 = [] { const char* e = getenv("AHA_GEMM_BAR2"); return e ? atoi(e) != 0 : true; }();
      if (!bar2 && ACT == ACT_NONE && !B && !R) {   // A/B: one barrier per phase
        static DevOnce once1;
        if (auto once_guard = once1.first()) {
          hipFuncSetAttribute((const void*)gemm256q_kernel<ACT_NONE, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        hipLaunchKernelGGL((gemm256q_kernel<ACT_NONE, false, false, false>), dim3(ntm * ntn), dim3(256), lds, st, a, nk, 0);
        return;
      }
      static const int abl = [] { const char* e = getenv("AHA_GEMM_ABL"); return e ? atoi(e) : 0; }();
      if (abl >= 1 && abl <= 3 && ACT == ACT_NONE && !B && !R) {   // ablations (debug; results are wrong by construction)
        static DevOnce once2;
        if (auto once_guard = once2.first()) {
          hipFuncSetAttribute((const void*)gemm256q_kernel<ACT_NONE, false, false, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipFuncSetAttribute((const void*)gemm256q_kernel<ACT_NONE, false, false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipFuncSetAttribute((const void*)gemm256q_kernel<ACT_NONE, false, false, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (abl == 1) hipLaunchKernelGGL((gemm256q_kernel<ACT_NONE, false, false, true, 1>), dim3(ntm * ntn), dim3(256), lds, st, a, nk, 0);
        if (abl == 2) hipLaunchKernelGGL((gemm256q_kernel<ACT_NONE, false, false, true, 2>), dim3(ntm * ntn), dim3(256), lds, st, a, nk, 0);
        if (abl == 3) hipLaunchKernelGGL((gemm256q_kernel<ACT_NONE, false, false, true, 3>), dim3(ntm * ntn), dim3(256), lds, st, a, nk, 0);
        return;
      }
#endif
      hipLaunchKernelGGL((gemm256q_kernel<ACT, B, R>), dim3(ntm * ntn), dim3(256), lds, st, a, nk, 0);
      return;
    }
    {
      static DevOnce once2;
      if (auto once_guard = once2.first()) {
        hipFuncSetAttribute((const void*)gemm256p_kernel<ACT, B, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      }
#ifdef AHA_DEBUG_KERNELS   // ablations (results wrong by construction) and the segment timeline: debug builds only
      static const int mode = [] { const char* e = getenv("AHA_GEMM_MODE"); return e ? atoi(e) : 0; }();
      if (mode >= 2 && mode <= 4 && ACT == ACT_NONE && !B && !R) {   // ablations (debug; results are wrong by construction)
        static DevOnce once3;
        if (auto once_guard = once3.first()) {
          hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_NONE, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_NONE, false, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_NONE, false, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        if (mode == 2) hipLaunchKernelGGL((gemm256p_kernel<ACT_NONE, false, false, 2>), dim3(ntm * ntn), dim3(512), lds, st, a, zero_block(), nk, nullptr);
        if (mode == 3) hipLaunchKernelGGL((gemm256p_kernel<ACT_NONE, false, false, 3>), dim3(ntm * ntn), dim3(512), lds, st, a, zero_block(), nk, nullptr);
        if (mode == 4) hipLaunchKernelGGL((gemm256p_kernel<ACT_NONE, false, false, 4>), dim3(ntm * ntn), dim3(512), lds, st, a, zero_block(), nk, nullptr);
        return;
      }
      static const bool tr = [] { const char* e = getenv("AHA_GEMM_TRACE"); return e && atoi(e) != 0; }();
      if (tr && ACT == ACT_NONE && !B && !R) {
        static unsigned long long* d_tr = nullptr;
        if (!d_tr) {
          hipMalloc((void**)&d_tr, 48 * 8);
          hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_NONE, false, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
        hipMemsetAsync(d_tr, 0, 48 * 8, st);
        hipLaunchKernelGGL((gemm256p_kernel<ACT_NONE, false, false, 1>), dim3(ntm * ntn), dim3(512), lds, st, a, zero_block(), nk, d_tr);
        unsigned long long h[48];
        hipMemcpyAsync(h, d_tr, sizeof(h), hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        for (int w = 0; w < 2; ++w) {
          fprintf(stderr, "[gemm trace] wave row %d:", w);
          for (int i = 1; i < 20 && h[w * 20 + i]; ++i) fprintf(stderr, " %llu", h[w * 20 + i] - h[w * 20 + i - 1]);
          fprintf(stderr, "\n");
        }
        fprintf(stderr, "[gemm trace] main loop: %llu shader cycles in %.2f us => %.3f GHz\n", h[42] - h[40], (double)(h[43] - h[41]) * 0.01,
                (double)(h[42] - h[40]) / ((double)(h[43] - h[41]) * 10.0));
        return;
      }
#endif
      hipLaunchKernelGGL((gemm256p_kernel<ACT, B, R>), dim3(ntm * ntn), dim3(512), lds, st, a, zero_block(), nk);
    }
    return;
  }
  static DevOnce once_p;
  if (auto once_guard = once_p.first()) {
    hipFuncSetAttribute((const void*)gemm256p_kernel<ACT_PARTIAL_F32, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  GemmArgs p = a;  // pass 1: f32 slabs [splitk][M][N] in the caller's workspace
  p.C = a.workspace;
  p.ldc = a.N;
  p.bias = nullptr;
  p.residual = nullptr;
  p.act = ACT_PARTIAL_F32;
  const int kps = (nk + splitk - 1) / splitk;
  static const bool quad_sk = [] { const char* e = getenv("AHA_GEMM_QUAD"); return e ? atoi(e) != 0 : true; }();
  if (quad_sk && a.K % BK == 0 && 256.0 * 2.0 * (double)std::max(a.lda, a.ldw) < 1.0e9) {
    static DevOnce onceq;
    if (auto once_guard = onceq.first()) {
      hipFuncSetAttribute((const void*)gemm256q_kernel<ACT_PARTIAL_F32, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute((const void*)gemm256q_kernel<ACT_PARTIAL_F32, false, false, true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (n192) hipLaunchKernelGGL((gemm256q_kernel<ACT_PARTIAL_F32, false, false, true, 0, true>), dim3(ntm * ((a.N + 191) / 192), splitk), dim3(256), lds, st, p, kps, 0);
    else hipLaunchKernelGGL((gemm256q_kernel<ACT_PARTIAL_F32, false, false>), dim3(ntm * ntn, splitk), dim3(256), lds, st, p, kps, 0);
  } else {
    hipLaunchKernelGGL((gemm256p_kernel<ACT_PARTIAL_F32, false, false>), dim3(ntm * ntn, splitk), dim3(512), lds, st, p, zero_block(), kps, nullptr);
  }
  launch_splitk_reduce<ACT, B, R>(a, (nk + kps - 1) / kps, st, norm_fused);
}

template <int ACT>
void launch256_act(const GemmArgs& a, int splitk, hipStream_t st, bool* norm_fused, bool n192) {
  // (n192 with bias / residual only matters under split-K: the slabs are plain f32 sums whatever the epilogue of the reduce pass)
  if (a.bias && a.residual) launch256_one<ACT, true, true>(a, splitk, st, norm_fused, n192);
  else if (a.bias) launch256_one<ACT, true, false>(a, splitk, st, norm_fused, n192);
  else if (a.residual) launch256_one<ACT, false, true>(a, splitk, st, norm_fused, n192);
  else launch256_one<ACT, false, false>(a, splitk, st, norm_fused, n192);
}

struct GemmPlan { int tile, splitk; double cost = 0; bool streamk = false; };   // streamk: the persistent kernel of kernels_gemm_sk.hip on `tile`-column tiles

// Tile choice by a two-parameter cost model fitted to scripts/bench_gemm.py on MI355X (256 CUs): a CU retires one
// (128^2 tile, 64-deep k step) in ~0.68 us when it has two or more 128^2 blocks resident, and one (256^2 tile, k step) in
// ~2.0 us (= 0.5 us per 128^2-equivalent: 1.35x better, measured 823 -> 1074 TFLOP/s at 8192^3); the last, partly filled
// round of work units costs a full round.  256^2 tiles therefore win when the unit count fills the CUs (long prompts,
// wide N) and lose on wave quantisation (e.g. 272 units = 2 rounds).  Few tiles + long K: split K into f32 slabs
// (caller's workspace) and pay a reduce pass that streams (splitk + 1) x M x N x 4 bytes.
int g_force_tile = 0, g_force_splitk = 0;  // aha_hip_debug_gemm_plan (tests): 0 = automatic

GemmPlan plan_gemm(const GemmArgs& a) {
  if (g_force_tile == 2128) return GemmPlan{(a.K + BK - 1) / BK >= 2 ? 2128 : 128, 1};
  if (g_force_tile == 128) {
    int sk = g_force_splitk > 1 ? g_force_splitk : 1;
    const int nk_ = (a.K + BK - 1) / BK;
    if (sk > 1 && (a.act == ACT_SILU_MUL_PAIRS || a.act == ACT_PARTIAL_F32 || !a.workspace || (a.N & 3) || nk_ / sk < 2 ||
                   (size_t)sk * a.M * a.N * 4 > a.workspace_bytes)) sk = 1;
    return GemmPlan{128, sk};
  }
  if (g_force_tile == 1256 || g_force_tile == 1192) {   // tests: the persistent kernel wherever it has an instantiation and a workspace
    const int tn = g_force_tile - 1000;
    if (a.M >= 1 && a.K % BK == 0 && streamk_has_kernel(a.act, a.bias != nullptr, a.residual != nullptr, tn == 192) &&
        streamk_estimate(a, tn, nullptr, nullptr) >= 0.0)
      return GemmPlan{tn, 1, 0.0, true};
    return GemmPlan{128, 1};
  }
  if ((g_force_tile == 256 || g_force_tile == 192) && a.M >= 1) {
    int sk = g_force_splitk > 1 ? g_force_splitk : 1;
    if (sk > 1 && (a.act == ACT_SILU_MUL_PAIRS || a.act == ACT_PARTIAL_F32 || !a.workspace || (a.N & 3) ||
                   (size_t)sk * a.M * a.N * 4 > a.workspace_bytes)) sk = 1;
    // 192-column tiles: unsplit for the plain / gate+up epilogues, f32 slabs (any epilogue in the reduce pass) under split-K
    if (g_force_tile == 192 && a.K % BK == 0 && (sk > 1 || (!a.bias && !a.residual && (a.act == ACT_NONE || a.act == ACT_SILU_MUL_PAIRS))))
      return GemmPlan{192, sk};
    return GemmPlan{256, sk};
  }
  static const char* e_tile = getenv("AHA_GEMM_TILE");
  static const char* e_sk = getenv("AHA_GEMM_SPLITK");
  const double nk = (a.K + BK - 1) / BK;
  const double t128 = (double)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const double t256 = (double)((a.M + 255) / 256) * ((a.N + 255) / 256);
  // 128^2 kernel: 0.68 us per tile and k step is the throughput of a CU holding four blocks; a launch that leaves CUs with a single
  // block (or none) runs at that block's own latency.  Round 6: that latency was priced at 1.4 us per k step from cfg 3's deep-K shapes;
  // on the short-K shapes of BASELINE cfg 4 (Qwen3-ASR: M = 390 / 406 rows, K = 896 / 1024 = 14-16 k steps) a lone block measures
  // ~0.75 us per k step + ~4.5 us of launch, prologue and epilogue (scripts/tune_gemm.py, profiles/r06_tune_gemm_small.txt: qkv 14.4 us on
  // 128^2 tiles against 18.4-21.1 on the 256-row kernels the old constant chose, fc1 14.7 against 18.8, text qkv 16.3 against 20.4) --
  // and a GELU-erf epilogue on the 28 CUs that hold 256^2 tiles costs more than the GEMM (52 us in the model).  AHA_GEMM_LAT128 = the
  // per-k-step microseconds (A/B).
  static const double lat128 = [] { const char* e = getenv("AHA_GEMM_LAT128"); return e ? atof(e) : 0.75; }();
  // (the ring kernel -- <= one block per CU, >= 4 k steps, launch_one -- runs a lone block's k step in ~0.5 us: AHA_GEMM_LAT_RING)
  static const double lat_ring = [] { const char* e = getenv("AHA_GEMM_LAT_RING"); return e ? atof(e) : 0.5; }();
  const bool ring128 = t128 <= (double)gemm_streamk_cus() && nk >= 4;
  // (a norm riding on the call is folded into a split plan's reduce pass; behind an unsplit ring launch it is a ~6-us launch of its own --
  // cfg 2 o_proj, 2048 x 1024 x 2048: 23.1 us unsplit against 27.6 us as 256^2 x 4 slices, but 4.99 against 4.57 ms per prefill)
  const double cost128 = std::max(ceil(t128 / 256.0) * nk * 0.68, t128 < 512.0 ? nk * (ring128 ? lat_ring : lat128) + 4.5 : 0.0) +
                         (ring128 && a.norm_w ? 6.0 : 0.0);
  const bool can_split = a.act != ACT_SILU_MUL_PAIRS && a.act != ACT_PARTIAL_F32 && a.workspace != nullptr && (a.N & 3) == 0;
  GemmPlan best{128, 1};
  double best_cost = cost128;
  // 256 x 128 tiles on the eight-wave ring kernel (plan tile 2128): where the 128^2 tiling overflows one block per CU and this one does not
  static const bool tall_on = [] { const char* e = getenv("AHA_GEMM_TALL"); return e ? atoi(e) != 0 : true; }();
  static const double lat_tall = [] { const char* e = getenv("AHA_GEMM_LAT_TALL"); return e ? atof(e) : 0.85; }();
  const double t2128 = (double)((a.M + 255) / 256) * ((a.N + 127) / 128);
  if (tall_on && a.M >= 256 && nk >= 8 && nk <= 24 && t128 > (double)gemm_streamk_cus() && t2128 <= (double)gemm_streamk_cus() && !e_tile && !e_sk &&
      a.act != ACT_PARTIAL_F32) {
    const double c = nk * lat_tall + 5.0;   // (a riding norm is a launch of its own here as behind the 128^2 tiling this competes with)
    if (c < best_cost) {
      best = GemmPlan{2128, 1};
      best_cost = c;
    }
  }
  // 128^2 tiles x K slices on the ring kernel, one block per CU at most (launch_ring_splitk).  The reduce pass of these few-row shapes is
  // small and launch-sized: ~2 TB/s over its (slices + 1) slabs, not the 4 TB/s of the 256^2 plans' larger ones (scripts/tune_gemm.py at
  // 406 x 1024: 4 -> 8 slices + 1.9 us with four k steps less)
  static const bool ring_sk_on = [] { const char* e = getenv("AHA_GEMM_RING_SPLITK"); return e ? atoi(e) != 0 : true; }();
  if (ring_sk_on && can_split && nk >= 24 && !e_tile && !(e_sk && atoi(e_sk) == 1)) {   // (K >= 1536: below, the slices save ~2 us by the model -- inside its error)
    const double red_rate = (double)a.M * a.N < 1.0e6 ? 2.0e6 : 4.0e6;   // (bytes per us: a reduce pass of < 1 M outputs does not fill the chip)
    for (int sk : {2, 3, 4, 6, 8}) {
      if (e_sk && atoi(e_sk) != sk) continue;
      if ((size_t)sk * a.M * a.N * 4 > a.workspace_bytes || nk / sk < 4 || t128 * sk > (double)gemm_streamk_cus()) continue;
      const double c = ceil(nk / sk) * lat_ring + 4.5 + (double)(sk + 1) * a.M * a.N * 4.0 / red_rate + 3.0;
      if (c < best_cost) {
        best = GemmPlan{128, sk};
        best_cost = c;
      }
    }
  }
  // 256^2 units: tiles x K slices.  A CU retires one 64-deep k step of a 256^2 tile in ~1.5 us with the whole chip busy on the
  // four-wave kernel (1.75 us on the eight-wave one; both clock-bound there); the last, partly filled round of units costs a full round, so the
  // split factor is chosen to make tiles * sk land just under a multiple of the CU count -- any factor, not only powers of
  // two (M = 1542: qkv 168 tiles x 3 = 504 units).  The reduce pass streams (sk + 1) x M x N x 4 bytes.
  static const int sks[] = {1, 2, 3, 4, 5, 6, 8};
  for (int sk : sks) {
    if (a.M < 256) break;   // (a 256-row tile of fewer rows: the 128^2 plans above)
    if (sk > 1 && (!can_split || (size_t)sk * a.M * a.N * 4 > a.workspace_bytes || nk / sk < 8)) continue;
    const bool q4 = a.K % BK == 0 && (sk > 1 || nk >= 32 || !a.residual || (!a.bias && a.act != ACT_GELU_TANH && a.act != ACT_GELU_ERF));   // gemm256q (launch256_one)
    double c = ceil(t256 * sk / 256.0) * ceil(nk / sk) * (q4 ? 1.5 : 1.75);
    if (sk > 1) c += (double)(sk + 1) * a.M * a.N * 4.0 / 4.0e6 + 3.0;
    if (e_sk && atoi(e_sk) != sk) continue;
    if (c < best_cost || (e_tile && atoi(e_tile) == 256 && best.tile != 256)) {
      best = GemmPlan{256, sk};
      best_cost = c;
    }
  }
  // 256 x 192 tiles (gemm256q_kernel NF3; unsplit, plain / gate+up epilogues).  Measured on MI355X (scripts/bench_gemm_ragged.py,
  // profiles/r03_gemm_n192.md): a k step of the 3/4-size tile costs 0.73-0.78 of the 256^2 one, but whenever both tilings fill the
  // chip the GEMM runs at the same 1.15-1.28 PF either way (gate+up at M = 1280 / 1536 / 1542 / 1792: 228 / 269 / 262 / 281 us on
  // 256^2 against 235 / 241 / 275 / 293 us) -- whole-chip throughput at this clock, not tile rounds, sets the time, and the ragged
  // row tiles of the finer tiling stream one more pass of W.  Where it does pay is a launch whose 256^2 tiles leave CUs idle: the
  // cfg 3 qkv GEMM is 168 tiles of 256^2 on 256 CUs and 224 of 256 x 192 (78-80 -> 64-73 us; 81.0 -> 75.3 us inside the prefill).
  static const bool n192_on = [] { const char* e = getenv("AHA_GEMM_N192"); return e ? atoi(e) != 0 : true; }();
  if (n192_on && !a.bias && !a.residual && (a.act == ACT_NONE || a.act == ACT_SILU_MUL_PAIRS) && a.K % BK == 0 && nk >= 8 && a.M >= 256 &&
      !(e_sk && atoi(e_sk) != 1) && !(e_tile && atoi(e_tile) != 192)) {
    const double t192 = (double)((a.M + 255) / 256) * ((a.N + 191) / 192);
    if ((best.tile == 256 && best.splitk == 1 && t256 < 256.0 && t192 <= 256.0 && t192 > t256) || (e_tile && atoi(e_tile) == 192)) {
      best = GemmPlan{192, 1};
      best_cost = nk * 1.5 * 0.78;
    }
    // ROW5 (round 4): M = 256 q + r with r <= 32 runs as q row tiles on the 192-column kernel, the r rows as a fifth fragment row of the
    // last one.  cfg 3 gate+up: 6 x 128 = 768 tiles = exactly three rounds of 3/4-size tiles where 256^2 tiles need three rounds of
    // full-size ones (672 tiles, 96 of them six rows high and as slow as a full one).
    // By rounds (round 4; measured at M = 1280 / 1536 / 1542 / 1792, profiles/r03_gemm_n192.md, r04_gemm_row5.md): a round of 192-column
    // tiles costs ~0.80 of a round of 256^2 ones, so the finer tiling wins where it needs fewer than 1.25 x the rounds (M = 1536: 3 vs 3;
    // not M = 1280: 3 vs 2, nor M = 1792: 4 vs 3).
    static const bool row5_on = [] { const char* e = getenv("AHA_GEMM_ROW5"); return e ? atoi(e) != 0 : true; }();
    const int r5 = a.M % 256;
    const bool row5 = row5_on && a.M > 256 && r5 >= 1 && r5 <= 32;
    if (!e_tile) {
      const double t5 = (double)(row5 ? a.M / 256 : (a.M + 255) / 256) * ((a.N + 191) / 192);
      // (a tile that carries a fifth row costs 1.28 x: the round that holds them is that much longer)
      const double c5 = (ceil(t5 / 256.0) + (row5 ? 0.28 : 0.0)) * nk * 1.5 * 0.80;
      if (c5 < best_cost) {
        best = GemmPlan{192, 1};
        best_cost = c5;
      }
    }
  }
  // 192-column tiles under split-K (round 4: gemm256q_kernel<ACT_PARTIAL_F32, .., NF3>; any epilogue -- it runs in the reduce pass):
  // where tiles x slices make ONE round that fills at least half the chip.  ViT fc2 (4096 x 1152 x 4352, bias + residual): 96 tiles x 2
  // = 192 blocks of 34 K tiles, 50.8 us, against 80 x 3 = 240 blocks of 23 on 256^2 tiles, 54.9 us (one f32 slab less through the
  // reduce pass; 0.99 us per K tile at that fill) -- scripts/bench_gemm_fc2.py.  Also 0.6B down_proj at 2 k / 4 k tokens: 29.9 / 41.6 us
  // against 34.0 / 47.7 us.  The cfg 3 text projections never qualify (o / down: 154 tiles x 2 = two rounds, measured 82.7 / 196.7 us
  // against 61.8 / 138.3 us).
  if (n192_on && can_split && a.K % BK == 0 && a.M >= 256 && !e_tile && !e_sk && 256.0 * 2.0 * (double)std::max(a.lda, a.ldw) < 1.0e9) {
    const double t192 = (double)((a.M + 255) / 256) * ((a.N + 191) / 192);
    for (int sk : {2, 3, 4}) {
      if ((size_t)sk * a.M * a.N * 4 > a.workspace_bytes || nk / sk < 16 || t192 * sk > 256.0 || t192 * sk < 128.0) continue;   // (short K loops: ViT proj, K = 1152, 28.5 us against 26.3 us on the 128^2 kernel)
      const double c = ceil(nk / sk) * 1.5 * 0.72 + (double)(sk + 1) * a.M * a.N * 4.0 / 4.0e6 + 3.0;
      if (c < best_cost) {
        best = GemmPlan{192, sk};
        best_cost = c;
      }
    }
  }
  // The persistent kernel (kernels_gemm_sk.hip): whole tiles round by round, the last round cut along K and finished in the launch.
  // Its plan gives the k steps of a full 256^2 tile the slowest worker runs (segment overheads and the chip-wide cost of publishing and
  // re-reading the chunks included); priced with the same 1.5 us per k step as the rounds above (x 0.78 for the 192-column tile).  A
  // split-K plan above ends in a reduce pass that also does the RMSNorm riding on the call (GemmArgs::norm_w); every other plan pays
  // ~9 us for the separate launch_rmsnorm_rows.
  // What MI355X said (profiles/r04_gemm_sk.md): with whole tiles the persistent kernel runs exactly as fast as one tile per block
  // (gate+up 271.6 vs 271.7 us, 8192^3 730 vs 734 us), and every cut LOSES at the cfg 3 shapes -- the chunks are 256 KiB per piece
  // through the fabric, all pieces at once (o_proj in two pieces 80 + 8.5 us of RMSNorm against 73 us for the f32-slab plan whose
  // reduce pass holds the norm; gate+up 335 against 272 us) -- so there is no shape yet where it is picked for speed.
  // AHA_GEMM_STREAMK: 0 = never, 1 (default) = only when CUs are reserved for a concurrent stream (tensor-parallel prefill: a grid of
  // one-tile blocks cannot leave CUs free, a worker count can), 2 = wherever the model says it wins, 3 = wherever it can run.
  static const int sk_env = [] { const char* e = getenv("AHA_GEMM_STREAMK"); return e ? atoi(e) : 1; }();
  const bool reserving = gemm_streamk_workers() < gemm_streamk_cus();
  const int sk_mode = sk_env == 1 ? (reserving ? 2 : 0) : (sk_env >= 2 ? sk_env - 1 : 0);   // 0 off, 1 by the model, 2 wherever it can run
  const double norm_pen = a.norm_w ? 9.0 : 0.0;
  if (sk_mode > 0 && a.M >= 256 && a.K % BK == 0 && nk >= 8 && a.workspace && a.sk_counters && !e_tile && !e_sk &&
      256.0 * 2.0 * (double)std::max(a.lda, a.ldw) < 1.0e9) {
    double ref = best_cost + (best.splitk > 1 ? 0.0 : norm_pen);
    if (sk_mode >= 2) ref = 1.0e30;
    for (int tn : {256, 192}) {
      if (!streamk_has_kernel(a.act, a.bias != nullptr, a.residual != nullptr, tn == 192)) continue;
      const double steps = streamk_estimate(a, tn, nullptr, nullptr);
      if (steps < 0.0) continue;
      const double c = steps * 1.5 * (tn == 192 ? 0.78 : 1.0) + 2.0;
      if (c + norm_pen < 0.97 * ref) {
        best = GemmPlan{tn, 1, c, true};
        best_cost = c;
        ref = (c + norm_pen) / 0.97;
      }
    }
  }
  static const char* e_dbg = getenv("AHA_GEMM_PLAN_DEBUG");
  if (e_dbg && atoi(e_dbg)) fprintf(stderr, "[gemm plan] M=%d N=%d K=%d act=%d -> tile %d splitk %d%s (cost %.1f us, 128^2 %.1f us)\n", a.M, a.N, a.K, a.act, best.tile, best.splitk, best.streamk ? " persistent" : "", best_cost, cost128);
  best.cost = best_cost;
  if (a.M < 256 && best.tile != 128) best = GemmPlan{128, 1, cost128};
  if (e_tile && atoi(e_tile) == 128) best = GemmPlan{128, 1, cost128};
  return best;
}

}  // namespace

// The launch(es) launch_gemm would issue for a shape, without issuing them (host only): out[0..2] = {tile, splitk, 1 if the columns are
// split into a multiple of 256 + a tail launch}.  `ws_bytes` > 0 stands for a caller workspace of that size.
void debug_plan_gemm(int M, int N, int K, int act, bool has_bias, bool has_res, size_t ws_bytes, int* out, bool has_norm) {
  GemmArgs a{};
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = act == ACT_SILU_MUL_PAIRS ? N / 2 : N; a.act = act;
  static int dummy;
  a.bias = has_bias ? &dummy : nullptr;
  a.residual = has_res ? &dummy : nullptr;
  a.norm_w = has_norm ? &dummy : nullptr;
  a.workspace = ws_bytes ? &dummy : nullptr;
  a.workspace_bytes = ws_bytes;
  a.sk_counters = ws_bytes ? &dummy : nullptr;
  const GemmPlan plan = plan_gemm(a);
  out[0] = plan.tile + (plan.streamk ? 1000 : 0); out[1] = plan.splitk; out[2] = 0;   // (1256 / 1192: the persistent kernel)
  if (g_force_tile == 0 && a.act != ACT_SILU_MUL_PAIRS && a.act != ACT_PARTIAL_F32 && a.N > 512 && a.N % 256 != 0 && a.M >= 256) {   // (launch_gemm)
    GemmArgs am = a, at = a;
    am.N = a.N / 256 * 256;
    at.N = a.N - am.N;
    const GemmPlan pm = plan_gemm(am), pt = plan_gemm(at);
    if (pm.cost + pt.cost + 3.0 < plan.cost) { out[0] = pm.tile; out[1] = pm.splitk; out[2] = 1; }
  }
}

void set_gemm_plan_override(int tile, int splitk) {
  g_force_tile = tile;
  g_force_splitk = splitk;
  set_streamk_forced_cut((tile == 1256 || tile == 1192) ? splitk : 0);   // persistent kernel: splitk = style * 10 + cuts of the last round
}

static thread_local void* tl_ws = nullptr;
static thread_local size_t tl_ws_bytes = 0;
static thread_local void* tl_sk_counters = nullptr;
void get_gemm_workspace(void** ws, size_t* bytes, void** sk_counters) {
  *ws = tl_ws;
  *bytes = tl_ws_bytes;
  *sk_counters = tl_sk_counters;
}
void set_gemm_workspace(void* ws, size_t bytes, void* sk_counters) {
  tl_ws = ws;
  tl_ws_bytes = bytes;
  tl_sk_counters = sk_counters;
}

static void launch_planned(const GemmArgs& a, const GemmPlan& plan, hipStream_t st, bool* norm_fused = nullptr);
// the norm riding on a GEMM call as its own launch (no reduce pass to fold it into): RMSNorm, or LayerNorm when a bias vector rides along
static void launch_riding_norm(const GemmArgs& a, hipStream_t st) {
  if (a.norm_b) launch_layernorm_rows(a.C, a.norm_w, a.norm_b, a.norm_out, a.M, a.N, a.norm_eps, st);   // (contiguous rows: ldc == N on this path)
  else launch_rmsnorm_rows(a.C, a.norm_w, a.norm_out, a.M, a.N, a.ldc, a.N, a.norm_eps, st);
}

void launch_gemm(const GemmArgs& a_in, hipStream_t st) {
  if (a_in.M <= 0 || a_in.N <= 0) return;
  GemmArgs a = a_in;
  if (a.workspace == nullptr) {
    a.workspace = tl_ws;
    a.workspace_bytes = tl_ws_bytes;
    a.sk_counters = tl_sk_counters;
  }
  static const int e_grp = [] { const char* e = getenv("AHA_GEMM_GROUP"); return e ? atoi(e) : 8; }();
  a.tile_group = e_grp;
  static const int e_prow = [] { const char* e = getenv("AHA_GEMM_PARTIAL_ROWS"); return e ? atoi(e) : 1; }();
  a.partial_rows = e_prow;
  const GemmPlan plan = plan_gemm(a);
  // Ragged N (ViT fc1: N = 4304 = 16 x 256 + 208): the 17th column of 256^2 tiles makes 272 units = two rounds on 256 CUs for
  // 1.06 rounds of work.  Where the model says it pays, the columns up to the last multiple of 256 run as one GEMM and the
  // remaining columns as a second one (sub-views of W / C / bias / residual: same rounding, same results).
  static const bool nsplit_on = [] { const char* e = getenv("AHA_GEMM_NSPLIT"); return e ? atoi(e) != 0 : true; }();
  if (nsplit_on && g_force_tile == 0 && a.act != ACT_SILU_MUL_PAIRS && a.act != ACT_PARTIAL_F32 && a.N > 512 && a.N % 256 != 0 && a.M >= 256) {
    const int n_main = a.N / 256 * 256;
    GemmArgs am = a, at = a;
    am.N = n_main;
    at.N = a.N - n_main;
    at.W = (const bf16_t*)a.W + (int64_t)n_main * a.ldw;
    at.C = (bf16_t*)a.C + n_main;
    if (a.bias) at.bias = (const bf16_t*)a.bias + n_main;
    if (a.residual) at.residual = (const bf16_t*)a.residual + n_main;
    const GemmPlan pm = plan_gemm(am), pt = plan_gemm(at);
    if (pm.cost + pt.cost + 3.0 < plan.cost) {
      launch_planned(am, pm, st);
      launch_planned(at, pt, st);
      if (a.norm_w) launch_riding_norm(a, st);
      return;
    }
  }
  static const bool fuse_norm = [] { const char* e = getenv("AHA_GEMM_FUSE_NORM"); return e ? atoi(e) != 0 : true; }();
  bool norm_fused = false;
  launch_planned(a, plan, st, a.norm_w && fuse_norm ? &norm_fused : nullptr);
  if (a.norm_w && !norm_fused) launch_riding_norm(a, st);
}

void launch_gemm_grouped(const GemmArgs& a_in, hipStream_t st) {
  if (a_in.M <= 0 || a_in.N <= 0 || a_in.groups <= 0) return;
  static const int e_grp = [] { const char* e = getenv("AHA_GEMM_GROUP"); return e ? atoi(e) : 8; }();
  static const bool n192_on = [] { const char* e = getenv("AHA_GEMM_N192"); return e ? atoi(e) != 0 : true; }();
  static const bool fast_on = [] { const char* e = getenv("AHA_GEMM_GROUPED"); return e ? atoi(e) != 0 : true; }();
  GemmArgs a = a_in;
  a.tile_group = e_grp;
  const bool fast = fast_on && a.groups > 1 && a.M >= BM2 && a.K % BK == 0 && !a.bias && !a.residual && !a.norm_w &&
                    (a.act == ACT_NONE || a.act == ACT_SILU_MUL_PAIRS) && 256.0 * 2.0 * (double)std::max(a.lda, a.ldw) < 1.0e9 &&
                    (g_force_tile == 0 || g_force_tile == 256 || g_force_tile == 192);
  if (!fast) {
    for (int g = 0; g < a.groups; ++g) {
      GemmArgs s = a_in;
      s.groups = 1;
      const int crow = a.c_row0 + g * a.c_gstride;
      s.A = (const bf16_t*)a.A + (int64_t)g * a.a_gstride * a.lda;
      s.C = (bf16_t*)a.C + (int64_t)crow * a.ldc;
      if (a.residual) s.residual = (const bf16_t*)a.residual + (int64_t)crow * a.ldc;
      s.M = std::max(0, std::min(a.M, a.m_total - crow));
      if (s.M > 0) launch_gemm(s, st);
    }
    return;
  }
  // tile width by rounds, as plan_gemm does for one segment: a round of 192-column tiles costs ~0.8 of a round of 256^2 ones
  const int rows = a.groups * ((a.M + BM2 - 1) / BM2), cus = gemm_streamk_cus();
  const int t256 = rows * ((a.N + 255) / 256), t192 = rows * ((a.N + 191) / 192);
  bool n192 = n192_on && 0.8 * ((t192 + cus - 1) / cus) < 1.0 * ((t256 + cus - 1) / cus);
  if (g_force_tile == 192) n192 = true;
  if (g_force_tile == 256) n192 = false;
  const size_t lds = 4 * TILE2_BYTES;
  const int nk = a.K / BK;
  const dim3 grid((unsigned)(n192 ? t192 : t256));
#define AHA_GROUPED(ACT_, NF3_)                                                                                                      \
  do {                                                                                                                                \
    static DevOnce once;                                                                                                              \
    if (auto once_guard = once.first())                                                                                                                 \
      hipFuncSetAttribute((const void*)gemm256q_kernel<ACT_, false, false, true, 0, NF3_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((gemm256q_kernel<ACT_, false, false, true, 0, NF3_, false, true>), grid, dim3(256), lds, st, a, nk, 0);        \
  } while (0)
  if (a.act == ACT_NONE) {
    if (n192) AHA_GROUPED(ACT_NONE, true);
    else AHA_GROUPED(ACT_NONE, false);
  } else {
    if (n192) AHA_GROUPED(ACT_SILU_MUL_PAIRS, true);
    else AHA_GROUPED(ACT_SILU_MUL_PAIRS, false);
  }
#undef AHA_GROUPED
}

static void launch_planned(const GemmArgs& a, const GemmPlan& plan, hipStream_t st, bool* norm_fused) {
  if (norm_fused != nullptr) *norm_fused = false;
  if (plan.streamk && launch_gemm_streamk(a, plan.tile, st)) return;
  if (plan.tile == 256 || plan.tile == 192) {
    const bool n192 = plan.tile == 192;
    switch (a.act) {
      case ACT_NONE: launch256_act<ACT_NONE>(a, plan.splitk, st, norm_fused, n192); break;
      case ACT_GELU_TANH: launch256_act<ACT_GELU_TANH>(a, plan.splitk, st, norm_fused, n192 && plan.splitk > 1); break;
      case ACT_GELU_ERF: launch256_act<ACT_GELU_ERF>(a, plan.splitk, st, norm_fused, n192 && plan.splitk > 1); break;
      case ACT_SILU: launch256_act<ACT_SILU>(a, plan.splitk, st, norm_fused, n192 && plan.splitk > 1); break;
      case ACT_SILU_MUL_PAIRS: launch256_one<ACT_SILU_MUL_PAIRS, false, false>(a, 1, st, nullptr, n192); break;
      case ACT_PARTIAL_F32: launch256_one<ACT_PARTIAL_F32, false, false>(a, 1, st); break;
    }
    return;
  }
  if (plan.tile == 128 && plan.splitk > 1 && a.workspace != nullptr && a.act != ACT_SILU_MUL_PAIRS && a.act != ACT_PARTIAL_F32) {
    switch (a.act) {
      case ACT_NONE: launch_ring_splitk_act<ACT_NONE>(a, plan.splitk, st, norm_fused); break;
      case ACT_GELU_TANH: launch_ring_splitk_act<ACT_GELU_TANH>(a, plan.splitk, st, norm_fused); break;
      case ACT_GELU_ERF: launch_ring_splitk_act<ACT_GELU_ERF>(a, plan.splitk, st, norm_fused); break;
      case ACT_SILU: launch_ring_splitk_act<ACT_SILU>(a, plan.splitk, st, norm_fused); break;
    }
    return;
  }
  const bool tall = plan.tile == 2128;   // 256 x 128 tiles of the eight-wave ring kernel
  const int ntm = tall ? (a.M + 255) / 256 : (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
  dim3 grid(ntm * ntn);
  switch (a.act) {
    case ACT_NONE: launch_act<ACT_NONE>(a, grid, st, tall); break;
    case ACT_GELU_TANH: launch_act<ACT_GELU_TANH>(a, grid, st, tall); break;
    case ACT_GELU_ERF: launch_act<ACT_GELU_ERF>(a, grid, st, tall); break;
    case ACT_SILU: launch_act<ACT_SILU>(a, grid, st, tall); break;
    case ACT_SILU_MUL_PAIRS: launch_one<ACT_SILU_MUL_PAIRS, false, false>(a, grid, st, tall); break;
    case ACT_PARTIAL_F32: launch_one<ACT_PARTIAL_F32, false, false>(a, grid, st, tall); break;
  }
}

}  // namespace aha
