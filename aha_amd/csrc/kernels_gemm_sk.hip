// Persistent, segment-table-driven form of the four-wave 256-row MFMA GEMM (gemm256s_kernel) + its host planner.
//
// Why: at prefill sizes the tile count does not divide the CU count.  M = 1542 (BASELINE cfg 3): gate+up is 576 full + 96 ragged tiles of
// 256^2 on 256 CUs = 2.25 tiles of work per CU done in three tile times; o_proj / down_proj are 112 tiles, split along K into f32 slabs
// plus a reduce pass.  Here ONE workgroup per CU walks a list of SEGMENTS (tile, K-tile range) the host planned for it:
//   * whole tiles first, round by round, in the XCD-aware grouped order of the one-tile-per-block kernels (the tiles in flight on an
//     XCD form a compact rectangle in its L2 and advance through K together);
//   * then the tiles of the last, partly filled round, every one cut along K at the SAME points (so that neighbouring tiles' pieces
//     still share operand panels in the L2) and the pieces dealt to the workers longest-first;
//   * a tile cut into P pieces is finished inside the launch: every piece publishes its f32 partial sums as a 256-KiB chunk in
//     FRAGMENT order (the wave-wide 16-byte stores are whole 1-KiB runs; write-through, sc1), bumps the tile's counter, and the piece
//     that arrives LAST adds the chunks in K order -- a fixed order, so the result does not depend on which piece was last -- and runs
//     the ordinary epilogue chain (Linear -> bf16, bias, activation / gate * up, residual).  Nobody waits for anybody: no spinning,
//     no assumption about dispatch order or co-residency (cdna guide G16: sc1 payload both sides, per-wave drain, one relaxed
//     agent-scope counter).
// The main loop is gemm256_body.h's gemm256q_mainloop, unchanged: a segment is "K tiles [kt0, kt1) of the tile at (m0, n0)".
// The worker count is a launch parameter (AHA_GEMM_RESERVE_CUS / set_gemm_reserved_cus): under tensor parallelism the GEMM can leave
// CUs free for RCCL's kernels on the communication stream.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"
#include "gemm256_body.h"
#include "kernels.h"

namespace aha {

namespace {

constexpr int SK_CHUNK_BYTES = 256 * 256 * 4;   // one piece's partial sums of a 256 x 256 (or 256 x 192) tile, fragment order
constexpr int SK_FLAG_OFF = 4 * TILE2_BYTES - 16;   // "this block arrived last": the last word of the 128 KiB (free outside the k loop)

// fragment (nf, mf), register quad q of wave w: 1 KiB at this byte offset of a chunk; lane l holds bytes l * 16 .. + 16
__device__ __forceinline__ constexpr int sk_frag_off(int nf, int mf, int q) { return ((nf * 4 + mf) * 4 + q) * 1024; }

template <bool NF3>
__device__ __forceinline__ void sk_publish(const f32x16_t (&acc)[4][4], float* chunk, int wave, int lane) {
  constexpr int NFV = NF3 ? 3 : 4;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)chunk, 0, SK_CHUNK_BYTES, 0x00020000);
  const int wv = wave * 65536;
#pragma unroll
  for (int nf = 0; nf < NFV; ++nf)
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4_t v = {__float_as_uint(acc[nf][mf][4 * q]), __float_as_uint(acc[nf][mf][4 * q + 1]), __float_as_uint(acc[nf][mf][4 * q + 2]),
                           __float_as_uint(acc[nf][mf][4 * q + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, wv + sk_frag_off(nf, mf, q), 16 /* sc1: write-through */);
      }
}

// acc = [acc +] chunk[z0] + chunk[z0 + 1] + ... + chunk[z0 + P - 1], added in that order.  One unit = one fragment of all P chunks
// (4 P loads of 16 B per lane); D units are kept in flight (24-32 loads per lane, 24-32 KiB per wave: a single workgroup pulls its
// chunks at the rate the requests it has outstanding allow, nothing else; D = 4 for two chunks made the register allocator move an
// accumulator fragment through scratch in the gate * up instantiation).
template <int P, bool ADD, bool NF3>
__device__ __forceinline__ void sk_reduce(f32x16_t (&acc)[4][4], const float* chunks0, int nparts, int z0, int wave, int lane) {
  constexpr int NFV = NF3 ? 3 : 4, NFRAG = NFV * 4, D = P <= 2 ? 3 : 2;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)chunks0, 0, nparts * SK_CHUNK_BYTES, 0x00020000);
  const int wv = wave * 65536 + z0 * SK_CHUNK_BYTES;
  u32x4_t buf[D][P][4];
  auto issue = [&](int f, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        buf[b][p][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, wv + p * SK_CHUNK_BYTES + sk_frag_off(f >> 2, f & 3, q), 16 /* sc1 */);
  };
#pragma unroll
  for (int f = 0; f < D && f < NFRAG; ++f) issue(f, f);
#pragma unroll
  for (int f = 0; f < NFRAG; ++f) {
    const int b = f % D, nf = f >> 2, mf = f & 3;
    f32x16_t s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = ADD ? acc[nf][mf][r] + __uint_as_float(buf[b][0][r >> 2][r & 3]) : __uint_as_float(buf[b][0][r >> 2][r & 3]);
#pragma unroll
    for (int p = 1; p < P; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] += __uint_as_float(buf[b][p][r >> 2][r & 3]);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nf][mf][r] = s[r];
    if (f + D < NFRAG) issue(f + D, b);
  }
}

// One workgroup per CU (four waves, one per SIMD, 512 registers each, 128 KiB of LDS: exactly one fits).  tab: [G + 1] segment offsets,
// padded to `hdr` ints, then 8 ints per segment {m0, n0, kt0, kt1, nparts, slot, chunk, counter}.
template <int ACT, bool HAS_BIAS, bool HAS_RES, bool NF3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm256s_kernel(GemmArgs a, const int* __restrict__ tab, int hdr,
                                                                                                   float* chunks, unsigned* ctrs) {
  char* const smem = gemm_smem;
  constexpr int TN = NF3 ? 192 : 256, WC = TN / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  asm volatile("" ::: "v255", "a255");   // the whole register file, as gemm256q_kernel (nothing else is placed on this kernel's CUs)
  const int sbeg = __builtin_amdgcn_readfirstlane(tab[blockIdx.x]), send = __builtin_amdgcn_readfirstlane(tab[blockIdx.x + 1]);
  const int* segs = tab + hdr;
  for (int si = sbeg; si < send; ++si) {
    const int4 s0 = *reinterpret_cast<const int4*>(segs + 8 * si);
    const int m0 = __builtin_amdgcn_readfirstlane(s0.x), n0 = __builtin_amdgcn_readfirstlane(s0.y);
    const int kt0 = __builtin_amdgcn_readfirstlane(s0.z), kt1 = __builtin_amdgcn_readfirstlane(s0.w);
    const int nparts = __builtin_amdgcn_readfirstlane(segs[8 * si + 4]);   // (slot / chunk / counter are read where a cut tile needs them)
    f32x16_t acc[4][4];  // [n fragment of 32][m fragment of 32]
    // The lane / wave ids go through an opaque copy at every stage of a segment.  Otherwise the per-lane addresses of ALL stages
    // (fragment reads, staging offsets, chunk offsets, epilogue rows) are loop invariants of the segment loop, get hoisted in front
    // of it and stay live through the k loop -- whose own 500 registers then spill (325 VGPRs of scratch traffic inside the MFMA loop
    // in the first build: every counted wait a full drain).
    // (Likewise the accumulators' initial value: as a constant it becomes a sixteen-register zero vector kept across the segment loop.)
    int l0 = lane, w0 = wave;
    float zero = 0.f;
    asm volatile("" : "+v"(l0), "+s"(w0), "+s"(zero));
    gemm256q_mainloop<ACT, HAS_BIAS, HAS_RES, true, 0, NF3>(a, m0, n0, kt0, kt1, l0, w0, acc, zero);
    __syncthreads();   // every wave has left the k loop (and waited for its own DMAs): the stages are free
    bool finish = true;
    if (nparts > 1) {
      int lane = l0, wave = w0;
      asm volatile("" : "+v"(lane), "+s"(wave));
      const int slot = __builtin_amdgcn_readfirstlane(segs[8 * si + 5]), chunk = __builtin_amdgcn_readfirstlane(segs[8 * si + 6]);
      const int ctr = __builtin_amdgcn_readfirstlane(segs[8 * si + 7]);
      float* tile_chunks = chunks + (size_t)chunk * (SK_CHUNK_BYTES / 4);
      sk_publish<NF3>(acc, tile_chunks + (size_t)slot * (SK_CHUNK_BYTES / 4), wave, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's write-through stores are acknowledged ...
      __syncthreads();                                    // ... before the one counter bump that announces them
      if (tid == 0) {
        const unsigned prev = __hip_atomic_fetch_add(ctrs + ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = prev + 1u == (unsigned)nparts;
        if (last) __hip_atomic_store(ctrs + ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        *reinterpret_cast<int*>(&gemm_smem[SK_FLAG_OFF]) = last ? 1 : 0;
      }
      __syncthreads();
      finish = __builtin_amdgcn_readfirstlane(*reinterpret_cast<int*>(&gemm_smem[SK_FLAG_OFF])) != 0;
      if (finish) {   // the last piece to arrive: the tile's sums in K order (slot 0, 1, ...: the same order whoever is last)
        if (nparts == 2) sk_reduce<2, false, NF3>(acc, tile_chunks, nparts, 0, wave, lane);
        else if (nparts == 3) sk_reduce<3, false, NF3>(acc, tile_chunks, nparts, 0, wave, lane);
        else sk_reduce<4, false, NF3>(acc, tile_chunks, nparts, 0, wave, lane);   // (the planner cuts a tile into at most SK_MAX_PARTS = 4)
      }
    }
    if (finish) {
      int le = l0, we = w0;
      asm volatile("" : "+v"(le), "+s"(we));
      const int wm = we >> 1, wn = we & 1;
      if (ACT == ACT_PARTIAL_F32) epilogue32_rows_f32<WC>(a, acc, m0 + wm * 128, n0 + wn * WC, le, smem + we * (32 * 528));
      else epilogue32_rows<ACT == ACT_PARTIAL_F32 ? ACT_NONE : ACT, HAS_BIAS, HAS_RES, WC>(a, acc, m0 + wm * 128, n0 + wn * WC, le, smem + we * 8448);
    }
    __syncthreads();   // the epilogue's LDS bands and the flag are free before the next segment's prologue stages into them
  }
}

// ---- host: the plan ---------------------------------------------------------------------------------------------------------------
struct SkSeg { int m0, n0, kt0, kt1, nparts, slot, chunk, ctr; };

struct SkPlan {
  bool ok = false;
  int G = 0, tile_n = 256, style = 0, cuts = 1;   // style 0: equal pieces; 1: `cuts` big pieces + one remainder
  int n_chunks = 0, n_ctrs = 0, n_split_tiles = 0;
  double makespan = 0;   // k steps of a full tile on the slowest worker, overheads included
  std::vector<int> off;        // [G + 1]
  std::vector<SkSeg> segs;     // grouped by worker
};

struct SkCost {
  double ragged_floor = 0.23;   // cost of a k step of a tile whose MFMAs are all skipped (staging + barrier skeleton), in full k steps
  double seg = 3.0;             // per segment: pipeline fill + epilogue, in k steps
  double publish = 1.0;         // issuing a chunk's stores / the counter round trip, per piece
  double fixup = 1.0;           // issuing a chunk's loads in the last arriver, per chunk
  // What a cut really costs is chip-wide: every piece of the last round publishes at the same time and every finisher reads back at the
  // same time, 256 KiB each through the fabric (write-through).  Measured on MI355X (scripts/bench_gemm_sk.py, profiles/r04_gemm_sk.md):
  // o_proj cut in two = 50 MB published + read back = +25 us; gate+up's last round = 71 MB = +35 us: ~0.5 us per MB published, i.e. 0.34 k
  // steps (of 1.45 us) per MB, added to the slowest worker.
  double steps_per_mb = 0.34;
};
SkCost sk_cost() {
  static const SkCost c = [] {
    SkCost k;
    if (const char* e = getenv("AHA_GEMM_SK_COST")) sscanf(e, "%lf,%lf,%lf,%lf,%lf", &k.ragged_floor, &k.seg, &k.publish, &k.fixup, &k.steps_per_mb);
    return k;
  }();
  return c;
}

// the XCD-aware grouped tile order of tile_of_block (gemm256_body.h), on the host: virtual block id -> tile
void sk_tile_of_vblock(int M, int N, int tile_n, int group, int vbid, int& m0, int& n0) {
  const int ntm = (M + 255) / 256, ntn = (N + tile_n - 1) / tile_n, nwg = ntm * ntn;
  int bid = vbid;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int GRP = group > 0 ? group : (1 << 20);
  if (ntm <= ntn) {
    const int per = GRP * ntn, grp = bid / per, first = grp * GRP, gsz = std::min(ntm - first, GRP), rr = bid - grp * per;
    m0 = (first + rr % gsz) * 256;
    n0 = (rr / gsz) * tile_n;
  } else {
    const int per = GRP * ntm, grp = bid / per, first = grp * GRP, gsz = std::min(ntn - first, GRP), rr = bid - grp * per;
    n0 = (first + rr % gsz) * tile_n;
    m0 = (rr / gsz) * 256;
  }
}

// k-step cost of the tile at row m0 relative to a full tile: the block runs at the pace of its busier wave row (rows 0-127 of the tile)
double sk_tile_weight(int M, int m0, const SkCost& c) {
  const int rows = std::min(256, M - m0);
  if (rows > 96) return 1.0;
  return c.ragged_floor + (1.0 - c.ragged_floor) * ((rows + 31) / 32) / 4.0;
}

// pieces of a K range of nk tiles: style 0 = `cuts` equal pieces; style 1 = `cuts` pieces of L = 2 nk / (2 cuts + 1) and one remainder
// (two remainders of different tiles make one worker's share).  Every tile of the last round is cut at the same points.
std::vector<int> sk_cut_points(int nk, int style, int cuts) {
  std::vector<int> p{0};
  if (style == 0) {
    for (int i = 1; i < cuts; ++i) p.push_back((int)((int64_t)nk * i / cuts));
  } else {
    const int L = std::max(1, (2 * nk + cuts) / (2 * cuts + 1));
    for (int i = 1; i <= cuts && i * L < nk; ++i) p.push_back(i * L);
  }
  p.push_back(nk);
  p.erase(std::unique(p.begin(), p.end()), p.end());
  return p;
}

SkPlan sk_plan_one(int M, int N, int K, int tile_n, int G, int group, int style, int cuts, const SkCost& c) {
  SkPlan pl;
  pl.G = G; pl.tile_n = tile_n; pl.style = style; pl.cuts = cuts;
  const int ntm = (M + 255) / 256, ntn = (N + tile_n - 1) / tile_n, T = ntm * ntn, nk = K / 64, W = G / 8;
  struct Tile { int m0, n0; double w; };
  std::vector<std::vector<Tile>> per_xcd(8);
  for (int v = 0; v < T; ++v) {
    Tile t;
    sk_tile_of_vblock(M, N, tile_n, group, v, t.m0, t.n0);
    t.w = sk_tile_weight(M, t.m0, c);
    per_xcd[v & 7].push_back(t);
  }
  // (Measured and NOT done: moving the tiles of a ragged last row -- M = 1542: six rows -- to the end of their XCD's list so that the
  // whole rounds hold full tiles only.  A ragged tile is cheap only while the full tiles of its column panel run beside it and pull its
  // W rows into the L2; sixteen of them alone in a last round each stream their panel from HBM at the latency-bound pace of the
  // staging skeleton, ~76 us against 79 us for a FULL 256 x 192 tile: gate+up 313 us against 289 us in the interleaved order,
  // profiles/r04_gemm_sk.md.)
  std::vector<std::vector<SkSeg>> wsegs(G);
  std::vector<double> load(G, 0.0);
  const std::vector<int> cutp = sk_cut_points(nk, style, cuts);
  for (int x = 0; x < 8; ++x) {
    const auto& tl = per_xcd[x];
    const int n = (int)tl.size(), R = n / W;
    for (int i = 0; i < R * W; ++i) {   // whole tiles, round by round: worker j of the XCD takes the XCD's tiles j, j + W, ...
      const int b = x + 8 * (i % W);
      wsegs[b].push_back(SkSeg{tl[i].m0, tl[i].n0, 0, nk, 1, 0, 0, 0});
      load[b] += nk * tl[i].w + c.seg;
    }
    struct Piece { int tile, kt0, kt1, slot, nparts; double cost; };
    std::vector<Piece> pieces;
    std::vector<int> tile_chunk(n, 0), tile_ctr(n, 0);
    for (int i = R * W; i < n; ++i) {
      // a ragged tile whose whole cost is below a full tile's piece stays whole
      const double full_piece = (double)(cutp[1] - cutp[0]);
      const bool whole = cutp.size() == 2 || tl[i].w * nk <= full_piece * 1.25;
      const int np = whole ? 1 : (int)cutp.size() - 1;
      if (np > 1) {
        tile_chunk[i] = pl.n_chunks;
        tile_ctr[i] = pl.n_ctrs++;
        pl.n_chunks += np;
        ++pl.n_split_tiles;
      }
      for (int s = 0; s < np; ++s) {
        const int k0 = whole ? 0 : cutp[s], k1 = whole ? nk : cutp[s + 1];
        double cost = (k1 - k0) * tl[i].w + c.seg;
        if (np > 1) cost += c.publish + c.fixup * (np - 1) / np;   // (the last arriver pays the whole read: spread, it is not known who)
        pieces.push_back(Piece{i, k0, k1, s, np, cost});
      }
    }
    // longest piece first onto the least loaded worker of this XCD (stable: equal pieces keep tile order, so neighbours run together)
    std::stable_sort(pieces.begin(), pieces.end(), [](const Piece& p, const Piece& q) { return p.cost > q.cost; });
    for (const Piece& p : pieces) {
      int best = x;
      for (int j = 1; j < W; ++j)
        if (load[x + 8 * j] < load[best] - 1e-9) best = x + 8 * j;
      wsegs[best].push_back(SkSeg{tl[p.tile].m0, tl[p.tile].n0, p.kt0, p.kt1, p.nparts, p.slot, tile_chunk[p.tile], tile_ctr[p.tile]});
      load[best] += p.cost;
    }
  }
  pl.off.assign(G + 1, 0);
  for (int b = 0; b < G; ++b) {
    pl.off[b + 1] = pl.off[b] + (int)wsegs[b].size();
    pl.segs.insert(pl.segs.end(), wsegs[b].begin(), wsegs[b].end());
  }
  pl.makespan = *std::max_element(load.begin(), load.end()) + c.steps_per_mb * pl.n_chunks * (SK_CHUNK_BYTES / 1048576.0);
  pl.ok = true;
  return pl;
}

std::atomic<int> g_sk_force_cut{0};   // tests (set_streamk_forced_cut): style * 10 + cuts; 0 = the planner's choice

// best cut of the last round for this shape (all whole tiles when the rounds are full)
SkPlan sk_plan(int M, int N, int K, int tile_n, int G, int group, size_t ws_bytes, int max_ctrs) {
  const SkCost c = sk_cost();
  static const int cand[][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {1, 1}, {1, 2}, {1, 3}};   // at most SK_MAX_PARTS = 4 pieces per tile
  SkPlan best;
  const int nk = K / 64;
  for (const auto& cd : cand) {
    const int force_cut = g_sk_force_cut.load();
    if (force_cut > 0 && (cd[0] != force_cut / 10 || cd[1] != force_cut % 10)) continue;
    if (force_cut == 0 && cd[1] > 1 && nk / (cd[0] ? 2 * cd[1] + 1 : cd[1]) < 4) continue;   // pieces of at least 4 K tiles
    SkPlan p = sk_plan_one(M, N, K, tile_n, G, group, cd[0], cd[1], c);
    if ((size_t)p.n_chunks * SK_CHUNK_BYTES > ws_bytes || p.n_ctrs > max_ctrs) continue;
    if (!best.ok || p.makespan < best.makespan - 1e-9) best = std::move(p);
  }
  return best;
}

std::atomic<int> g_reserved_cus{-1};   // set_gemm_reserved_cus; -1: AHA_GEMM_RESERVE_CUS or 0

// CU count of the CURRENT device (a process may run GEMMs on several GPUs: the Python API takes device=): cached per device
int sk_num_cus() {
  static std::mutex mu;
  static std::map<int, int> by_dev;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
  std::lock_guard<std::mutex> lk(mu);
  auto it = by_dev.find(dev);
  if (it != by_dev.end()) return it->second;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
    (void)hipGetLastError();
    cus = 256;
  }
  by_dev[dev] = cus;
  return cus;
}

// ---- plan cache: one table per (shape, tile, workers) on the device, uploaded on the stream that needs it first --------------------------
struct SkEntry {
  SkPlan plan;
  int* d_tab = nullptr;
  int hdr = 0;
  std::vector<int> h_tab;
  hipEvent_t ev = nullptr;
  hipStream_t st = nullptr;
  bool landed = false;
};
// Everything below is per DEVICE (round-4 advisor: one process-wide arena put device 1's tables into device 0's memory): the cache key
// carries the device, every device has its own table arena, and an arena is recycled only after ITS device has been synchronised.
// Entries are shared_ptr and g_sk_mu is held from the lookup THROUGH the launch (launch_gemm_streamk), so no eviction -- cache limit,
// arena wrap, set_streamk_forced_cut -- can delete an entry or hand its arena space to another table between a thread's lookup and
// its kernel launch.
std::mutex g_sk_mu;
typedef std::tuple<int, int, int, int, int, int, int, size_t> SkKey;   // (device, M, N, K, tile_n, workers, group, workspace class)
std::map<SkKey, std::shared_ptr<SkEntry>> g_sk_cache;
struct SkArena { char* base = nullptr; size_t used = 0; };
std::map<int, SkArena> g_sk_arena;
constexpr size_t SK_ARENA_BYTES = (size_t)64 << 20;

// g_sk_mu held.  Drops the cached plans of `dev` (all devices: dev < 0) after synchronising the device(s) that may still read their tables.
void sk_drop_entries_locked(int dev, const SkEntry* keep = nullptr) {
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) (void)hipGetLastError();
  std::vector<int> devs;
  for (auto& kv : g_sk_cache) {
    const int d = std::get<0>(kv.first);
    if ((dev < 0 || d == dev) && std::find(devs.begin(), devs.end(), d) == devs.end()) devs.push_back(d);
  }
  for (int d : devs) {
    if (hipSetDevice(d) != hipSuccess || hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
  }
  if (!devs.empty() && hipSetDevice(cur) != hipSuccess) (void)hipGetLastError();
  for (auto it = g_sk_cache.begin(); it != g_sk_cache.end();) {
    const int d = std::get<0>(it->first);
    if ((dev < 0 || d == dev) && it->second.get() != keep) {
      if (it->second->ev) hipEventDestroy(it->second->ev);
      it = g_sk_cache.erase(it);
    } else {
      ++it;
    }
  }
  for (auto& kv : g_sk_arena)
    if (dev < 0 || kv.first == dev) kv.second.used = 0;
}

// g_sk_mu held by the caller.  upload = false: the plan only (the cost estimate of plan_gemm), no device table.
std::shared_ptr<SkEntry> sk_lookup_locked(int M, int N, int K, int tile_n, int G, int group, size_t ws_bytes, hipStream_t st, bool upload) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
  const size_t ws_class = std::min<size_t>(ws_bytes / SK_CHUNK_BYTES, 1 << 20);
  const SkKey key = std::make_tuple(dev, M, N, K, tile_n, G, group, ws_class);
  auto it = g_sk_cache.find(key);
  std::shared_ptr<SkEntry> e = it == g_sk_cache.end() ? nullptr : it->second;
  if (!e) {
    if (g_sk_cache.size() > 8192) sk_drop_entries_locked(-1);
    e = std::make_shared<SkEntry>();
    e->plan = sk_plan(M, N, K, tile_n, G, group, ws_bytes, SK_MAX_COUNTERS);
    g_sk_cache[key] = e;
  }
  if (!e->plan.ok || !upload) return e;
  if (!e->d_tab) {
    e->hdr = (G + 1 + 3) / 4 * 4;
    e->h_tab.assign(e->hdr + 8 * e->plan.segs.size(), 0);
    memcpy(e->h_tab.data(), e->plan.off.data(), (G + 1) * sizeof(int));
    memcpy(e->h_tab.data() + e->hdr, e->plan.segs.data(), e->plan.segs.size() * sizeof(SkSeg));
    const size_t bytes = (e->h_tab.size() * sizeof(int) + 255) / 256 * 256;
    SkArena& ar = g_sk_arena[dev];
    if (!ar.base && hipMalloc((void**)&ar.base, SK_ARENA_BYTES) != hipSuccess) {   // on the CURRENT device = the one the kernel runs on
      (void)hipGetLastError();
      ar.base = nullptr;
      e->plan.ok = false;
      return e;
    }
    if (ar.used + bytes > SK_ARENA_BYTES) {   // (thousands of shapes later) start this device's arena over: nothing in flight may still read a table
      sk_drop_entries_locked(dev, e.get());
      if (bytes > SK_ARENA_BYTES) { e->plan.ok = false; return e; }
    }
    e->d_tab = reinterpret_cast<int*>(ar.base + ar.used);
    ar.used += bytes;
    // pageable source: the runtime stages the copy before it returns; h_tab lives as long as the entry anyway
    if (hipMemcpyAsync(e->d_tab, e->h_tab.data(), e->h_tab.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(e->ev, st) != hipSuccess) {
      (void)hipGetLastError();
      e->plan.ok = false;
      return e;
    }
    e->st = st;
    return e;
  }
  if (!e->landed && st != e->st) {   // another stream uploaded it: order this stream behind the copy
    if (hipEventQuery(e->ev) == hipSuccess) e->landed = true;
    else (void)hipStreamWaitEvent(st, e->ev, 0);
  }
  return e;
}

template <int ACT, bool B, bool R, bool NF3>
void sk_launch_one(const GemmArgs& a, const SkEntry* e, hipStream_t st) {
  static std::atomic<unsigned long long> done{0};   // the dynamic-LDS attribute is per device: one bit per device id
  const size_t lds = 4 * TILE2_BYTES;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(done.load() & bit)) {
    hipFuncSetAttribute((const void*)gemm256s_kernel<ACT, B, R, NF3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    done.fetch_or(bit);
  }
  hipLaunchKernelGGL((gemm256s_kernel<ACT, B, R, NF3>), dim3(e->plan.G), dim3(256), lds, st, a, (const int*)e->d_tab, e->hdr, (float*)a.workspace,
                     (unsigned*)a.sk_counters);
}

}  // namespace

void set_gemm_reserved_cus(int n) { g_reserved_cus.store(n); }
int get_gemm_reserved_cus() { return g_reserved_cus.load(); }
// The AUTOMATIC reservation a model's communication stream asks for (model.hip ensure_comm_stream): process-global state with several
// possible holders (two tensor-parallel ranks as two models of one process), so it is reference-counted under a mutex -- the first holder
// saves the caller's setting and reserves, the last one to leave restores it (round-5 advisor: saved / restored per model, closing model A
// took the reservation away from model B, and two racing threads could both "save" and one restore 16 for good).  Returns whether the
// caller now holds a share (false: the user chose a number through the API or AHA_GEMM_RESERVE_CUS -- nothing to hold or release).
static std::mutex g_resv_mu;
static int g_resv_holders = 0, g_resv_prev = -1;
bool acquire_gemm_cu_reservation(int cus) {
  std::lock_guard<std::mutex> lk(g_resv_mu);
  if (g_resv_holders > 0) {
    ++g_resv_holders;
    return true;
  }
  if (getenv("AHA_GEMM_RESERVE_CUS") || gemm_streamk_workers() != gemm_streamk_cus()) return false;
  g_resv_prev = g_reserved_cus.load();
  g_reserved_cus.store(cus);
  g_resv_holders = 1;
  return true;
}
void release_gemm_cu_reservation() {
  std::lock_guard<std::mutex> lk(g_resv_mu);
  if (g_resv_holders > 0 && --g_resv_holders == 0) g_reserved_cus.store(g_resv_prev);
}
void set_streamk_forced_cut(int code) {
  std::lock_guard<std::mutex> lk(g_sk_mu);
  if (code != g_sk_force_cut.load()) sk_drop_entries_locked(-1);   // cached plans were made under the other setting
  g_sk_force_cut.store(code);
}

int gemm_streamk_cus() { return sk_num_cus() / 8 * 8; }

int gemm_streamk_workers() {
  int res = g_reserved_cus.load();
  if (res < 0) {
    static const int env = [] { const char* e = getenv("AHA_GEMM_RESERVE_CUS"); return e ? atoi(e) : 0; }();
    res = env;
  }
  const int cus = sk_num_cus();
  int g = (cus - std::max(0, res)) / 8 * 8;
  return std::max(8, std::min(g, cus / 8 * 8));
}

bool streamk_has_kernel(int act, bool has_bias, bool has_res, bool n192) {
  if (n192) return !has_bias && !has_res && (act == ACT_NONE || act == ACT_SILU_MUL_PAIRS);
  switch (act) {
    case ACT_NONE: return true;
    case ACT_SILU_MUL_PAIRS: return false;   // gate * up: 192-column tiles only (on 256 columns the chunk sum + this epilogue spill a fragment)
    case ACT_GELU_TANH: return has_bias && !has_res;
    case ACT_PARTIAL_F32: return !has_bias && !has_res;   // tensor-parallel row-split projections: the kernel the CU reservation is for
    default: return false;
  }
}

// k steps (of a full 256^2 tile) on the slowest worker, or a negative value when the shape cannot be planned (workspace too small for the
// chunks of the cut tiles, no counters).  *n_chunks / *n_split: what the plan would publish.
double streamk_estimate(const GemmArgs& a, int tile_n, int* n_chunks, int* n_split) {
  if (a.K % 64 || !a.workspace || !a.sk_counters) return -1.0;
  std::lock_guard<std::mutex> lk(g_sk_mu);
  const std::shared_ptr<SkEntry> e = sk_lookup_locked(a.M, a.N, a.K, tile_n, gemm_streamk_workers(), a.tile_group, a.workspace_bytes, nullptr, false);
  if (!e || !e->plan.ok) return -1.0;
  if (n_chunks) *n_chunks = e->plan.n_chunks;
  if (n_split) *n_split = e->plan.n_split_tiles;
  return e->plan.makespan;
}

bool launch_gemm_streamk(const GemmArgs& a, int tile_n, hipStream_t st) {
  const bool n192 = tile_n == 192;
  if (!streamk_has_kernel(a.act, a.bias != nullptr, a.residual != nullptr, n192) || a.K % 64 || !a.workspace || !a.sk_counters) return false;
  const int workers = gemm_streamk_workers();
  std::lock_guard<std::mutex> lk(g_sk_mu);   // from the lookup through the launch: see the plan cache
  const std::shared_ptr<SkEntry> ep = sk_lookup_locked(a.M, a.N, a.K, tile_n, workers, a.tile_group, a.workspace_bytes, st, true);
  const SkEntry* e = ep.get();
  if (!e || !e->plan.ok) return false;
  const bool B = a.bias != nullptr, R = a.residual != nullptr;
  if (n192) {
    if (a.act == ACT_NONE) sk_launch_one<ACT_NONE, false, false, true>(a, e, st);
    else sk_launch_one<ACT_SILU_MUL_PAIRS, false, false, true>(a, e, st);
    return true;
  }
  switch (a.act) {
    case ACT_NONE:
      if (B && R) sk_launch_one<ACT_NONE, true, true, false>(a, e, st);
      else if (B) sk_launch_one<ACT_NONE, true, false, false>(a, e, st);
      else if (R) sk_launch_one<ACT_NONE, false, true, false>(a, e, st);
      else sk_launch_one<ACT_NONE, false, false, false>(a, e, st);
      break;
    case ACT_GELU_TANH: sk_launch_one<ACT_GELU_TANH, true, false, false>(a, e, st); break;
    case ACT_PARTIAL_F32: sk_launch_one<ACT_PARTIAL_F32, false, false, false>(a, e, st); break;
    default: return false;
  }
  return true;
}

// tests / tools (host only): the segments the planner hands the workers.  out: 8 ints per segment (SkSeg), grouped by worker; off_out:
// [workers + 1].  Returns the number of segments (the call fills at most `cap` of them), or -1.  info: {workers, chunks, counters,
// split tiles, style, cuts, makespan x 1000}.
int debug_streamk_plan(int M, int N, int K, int tile_n, int workers, int group, size_t ws_bytes, int* out, int cap, int* off_out, int* info) {
  if (K % 64 || workers < 8 || workers % 8 || (tile_n != 256 && tile_n != 192)) return -1;
  const SkPlan p = sk_plan(M, N, K, tile_n, workers, group, ws_bytes, SK_MAX_COUNTERS);
  if (!p.ok) return -1;
  for (int i = 0; i < (int)p.segs.size() && i < cap; ++i) memcpy(out + 8 * i, &p.segs[i], sizeof(SkSeg));
  if (off_out) memcpy(off_out, p.off.data(), (workers + 1) * sizeof(int));
  if (info) {
    info[0] = p.G; info[1] = p.n_chunks; info[2] = p.n_ctrs; info[3] = p.n_split_tiles; info[4] = p.style; info[5] = p.cuts;
    info[6] = (int)(p.makespan * 1000.0);
  }
  return (int)p.segs.size();
}

}  // namespace aha
