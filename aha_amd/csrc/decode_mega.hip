// Persistent decode-step kernel: ONE launch runs the whole token step (SURVEY.md section 8a D0-D12 at S = 1) --
// embedding row, every decoder layer (RMSNorm + QKV matvec, q/k norm + rope + KV append + paged attention, o_proj +
// residual, RMSNorm + gate/up + SiLU*mul, down + residual), final norm and the lm_head matvec with argmax partials.
//
// Why: decode is HBM-bound (15.1 GB of weights per token for Qwen3-VL-8B) but at batch 1 a layer is only ~386 MB, i.e.
// ~60 us at the achievable 6.3 TB/s, cut into 5 dependent launches.  Every kernel boundary drains the memory pipeline
// (tail of the old grid, launch, ramp-up of the new one: ~4-8 us each, ~40% of the step).  Here the grid stays resident
// (2 blocks per CU) and phases are separated by a grid barrier instead; because weights do not depend on activations,
// every block requests the first weight tile of phase p+1 BEFORE it waits for phase p to complete, so HBM keeps
// streaming across the dependency.  The arithmetic is the stand-alone kernels' (gemv_body.h, attn_decode_body.h):
// the two decode paths are bit-identical (tests/test_model_gpu.py::test_decode_fused_launches_equal_launch_per_op).
//
// Grid barrier: see GridBarrier below.  A bounded spin turns a would-be hang into a reported error.
#include <stdlib.h>

#include "attn_decode_body.h"
#include "gemv_body.h"
#include "model.h"

namespace aha {

namespace {

constexpr unsigned SPIN_LIMIT = 1u << 21;  // polls (~1 us each) before a block gives up and raises the error word
constexpr int BAR_GROUP = 32;              // blocks per arrival group (scripts/bench_barrier.hip: 2.3 us at 512 blocks)

// Two-level grid barrier.  A flat counter costs ~15 ns per arrival because same-address atomics serialise at the memory
// side (7.8 us for 512 blocks, measured); here 32 blocks share a group counter, the last arriver of a group bumps the
// root, the group's first block polls the root and raises the group's flag, the other 31 poll the flag.  All words sit
// on their own 128-byte lines and only ever grow, so nothing is reset between launches: the host passes the number of
// barriers completed so far (done0).  Layout (32-bit words): [0] root, [32] sticky error, [64 + 32 g] group counters,
// [4096 + 32 g] group flags.
//
// Activations that cross the barrier are agent-scope atomics (common.h act_*): write-through stores, cache-bypassing
// loads.  Every wave waits for its own outstanding stores (s_waitcnt vmcnt(0)) before the block announces itself; no L2
// write-back / invalidate is needed, so all barrier operations are relaxed.
struct GridBarrier {
  unsigned* mem;
  unsigned nblk, bid;
  unsigned done;  // barriers this block has arrived at (continues across launches)

  static __device__ __forceinline__ unsigned ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  static __device__ __forceinline__ void st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  static __device__ __forceinline__ unsigned add(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

  template <int SLEEP>
  __device__ __forceinline__ void spin_until(const unsigned* p, unsigned target) {
    unsigned* err = mem + 32;
    unsigned polls = 0;
    while ((int)(ld(p) - target) < 0) {
      __builtin_amdgcn_s_sleep(SLEEP);
      if ((++polls & 63u) == 0) {
        if (ld(err) != 0u) return;
        if (polls > SPIN_LIMIT) {
          st(err, 1u);
          return;
        }
      }
    }
  }
  __device__ __forceinline__ void arrive() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), this wave: its activation stores have been acknowledged
    __syncthreads();
    ++done;
    if (threadIdx.x == 0) {
      const unsigned g = bid / BAR_GROUP;
      const unsigned gsize = min((unsigned)BAR_GROUP, nblk - g * BAR_GROUP);
      if (add(mem + 64 + 32 * g) + 1 == done * gsize) add(mem);  // last of the group this round -> root
    }
  }
  // until every block of the grid has arrived `done` times
  __device__ __forceinline__ void wait() {
    if (threadIdx.x == 0) {
      const unsigned g = bid / BAR_GROUP, ng = (nblk + BAR_GROUP - 1) / BAR_GROUP;
      unsigned* flag = mem + 4096 + 32 * g;
      if (bid % BAR_GROUP == 0) {
        spin_until<1>(mem, done * ng);
        st(flag, done);
      } else {
        spin_until<4>(flag, done);
      }
    }
    __syncthreads();
  }
};

template <int QR_, int QU_, int OR_, int OU_, int GR_, int GU_, int DR_, int DU_, int LR_, int LU_>
struct MegaProfile {
  static constexpr int QR = QR_, QU = QU_, OR = OR_, OU = OU_, GR = GR_, GU = GU_, DR = DR_, DU = DU_, LR = LR_, LU = LU_;
};
// rows-per-wave x 512-column chunks in flight per register buffer, per matvec (same choices plan_gemv makes)
using ProfileLarge = MegaProfile<1, 8, 1, 8, 2, 4, 1, 8, 4, 4>;  // hidden >= 2048 (Qwen3-VL-8B: 4096)
using ProfileSmall = MegaProfile<2, 2, 1, 4, 1, 2, 1, 4, 4, 2>;  // hidden 1024 (Qwen3-0.6B, Qwen3-ASR text tower)

template <class P>
__global__ __launch_bounds__(256, 2) void decode_step_kernel(DecodeMegaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  const int bid = blockIdx.x, nblk = gridDim.x;
  GridBarrier bar{a.bar, (unsigned)nblk, (unsigned)bid, a.bar_done0};
  // optional timeline of blocks 0, 100, 300, nblk-1 (AHA_MEGA_TRACE): phase start / barrier passed / compute done
  const int tslot = bid == 0 ? 0 : bid == 100 ? 1 : bid == 300 ? 2 : bid == nblk - 1 ? 3 : -1;
  int tphase = 0;
  auto stamp = [&](int k) {
    if (a.trace != nullptr && tslot >= 0 && threadIdx.x == 0)
      a.trace[((size_t)tslot * (5 * a.n_layers + 1) + tphase) * 3 + k] = wall_clock64();
  };
  auto wait = [&] {
    bar.wait();
    stamp(1);
  };
  const int nq = a.nh * 128, nkv = a.kvh * 128;
  const bf16_t* x_embed = (const bf16_t*)a.embed + (size_t)a.state->token * a.H;  // D0: embedding row of the token

  for (int li = 0; li < a.n_layers; ++li) {
    const DecodeLayerDev L = a.layers[li];
    const void* x_in = li == 0 ? (const void*)x_embed : (const void*)a.x;
    {  // h = RMSNorm(x); qkv = h Wqkv^T                              (qwen3/model.rs:79, modules.rs:538-552)
      GemvArgs g{};
      g.W = L.wqkv; g.x = x_in; g.norm_w = L.in_norm; g.eps = a.eps; g.y = a.qkv; g.N = nq + 2 * nkv; g.K = a.H;
      stamp(0);
      gemv_body<P::QR, P::QU, GEMV_STORE, true>(g, xs, bid, nblk, wait);
      stamp(2); ++tphase;
      bar.arrive();
    }
    if (bid < a.kvh * a.nsplit) {  // q/k norm + rope + KV append + attention over the paged cache (modules.rs:544-574, 757-813)
      AttnDecodeFusedArgs f{};
      f.qkv = a.qkv; f.q_norm_w = L.q_norm; f.k_norm_w = L.k_norm; f.pos = a.state->pos; f.inv_freq = a.inv_freq;
      f.axis_map = a.axis_map; f.kv.page_ptrs = a.page_ptrs; f.kv.layer_off = L.kv_layer_off; f.kv.kvh = a.kvh; f.kv.d = 128;
      f.kv_start = &a.state->kv_start; f.kv_len = &a.state->kv_len; f.part_o = a.part_o; f.part_ml = a.part_ml;
      f.nh = a.nh; f.kvh = a.kvh; f.nsplit = a.nsplit; f.eps = a.eps; f.scale = a.scale;
      f.o = a.attn; f.head_ctr = a.bar + DECODE_HEAD_CTR_WORD; f.ctr_target = a.head_ctr_target + (unsigned)li * (unsigned)a.nsplit;
      stamp(0);
      attn_decode_fused_body<true>(f, smem, bid % a.kvh, bid / a.kvh, a.nsplit, wait);
      stamp(2);
      bar.arrive();
    }
    {  // x = x + attn Wo^T                                           (modules.rs:577, qwen3/model.rs:81)
      GemvArgs g{};
      g.W = L.wo; g.x = a.attn; g.residual = x_in; g.y = a.x; g.N = a.H; g.K = nq;
      // Blocks without an attention unit come straight here: their o_proj tile is requested first, then they pass the
      // qkv barrier and announce themselves at the attention barrier in order (group counters assume that no block
      // arrives at barrier k+1 before every block of its group has arrived at barrier k).
      const bool idle_in_attn = bid >= a.kvh * a.nsplit;
      ++tphase;
      stamp(0);
      gemv_body<P::OR, P::OU, GEMV_RESIDUAL, true>(g, xs, bid, nblk, [&] {
        if (idle_in_attn) {
          bar.wait();
          bar.arrive();
        }
        wait();
      });
      stamp(2); ++tphase;
      bar.arrive();
    }
    {  // act = silu(h Wg^T) * (h Wu^T), h = RMSNorm(x)                (qwen3/model.rs:83, modules.rs:81-84)
      GemvArgs g{};
      g.W = L.wgu; g.x = a.x; g.norm_w = L.post_norm; g.eps = a.eps; g.y = a.act; g.N = a.I; g.K = a.H;
      stamp(0);
      gemv_body<P::GR, P::GU, GEMV_SILU_MUL, true>(g, xs, bid, nblk, wait);
      stamp(2); ++tphase;
      bar.arrive();
    }
    {  // x = x + act Wd^T                                             (modules.rs:85, qwen3/model.rs:86)
      GemvArgs g{};
      g.W = L.wdown; g.x = a.act; g.residual = a.x; g.y = a.x; g.N = a.H; g.K = a.I;
      stamp(0);
      gemv_body<P::DR, P::DU, GEMV_RESIDUAL, true>(g, xs, bid, nblk, wait);
      stamp(2); ++tphase;
      bar.arrive();
    }
  }
  {  // logits = RMSNorm(x) W_lm^T (+ per-block argmax partials)          (qwen3/model.rs:186-188, generate.rs:75-84)
    GemvArgs g{};
    g.W = a.lm_head; g.x = a.x; g.norm_w = a.final_norm; g.eps = a.eps; g.N = a.vocab; g.K = a.H;
    g.y_f32 = a.logits; g.blk_max = a.blk_max; g.blk_idx = a.blk_idx; g.h_out = a.h_out;
    stamp(0);
    gemv_body<P::LR, P::LU, GEMV_LOGITS, true>(g, xs, bid, nblk, wait);
    stamp(2);
  }
}

// ---- attention + o_proj in one launch (opt-in: AHA_DECODE_AO=1) ---------------------------------------------------------------
// A full grid barrier costs as much as a kernel boundary (profiles/r01_decode_mega_timeline.md), but the attention ->
// o_proj dependency is few-to-many: the kvh blocks that merged a kv head's splits produce, all blocks consume.  One launch
// of the o_proj grid: the first kvh*nsplit blocks run the fused attention body, the merging block of each kv head bumps
// a counter (kvh same-address atomics), every block requests its first o_proj weight tile immediately and waits for the
// counter only before the prologue that reads the attention output.  Removes one kernel boundary per layer and
// hides the attention latency behind the o_proj weight stream.  Blocks are dispatched in index order, so the producers
// are always resident before any consumer spins.
template <int R, int U>
__global__ __launch_bounds__(256, 2) void attn_oproj_kernel(AttnDecodeFusedArgs f, GemvArgs g, unsigned* sync, unsigned target) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x, nblk = gridDim.x;
  unsigned* ctr = sync + DECODE_AO_CTR_WORD;
  unsigned* err = sync + DECODE_MEGA_BAR_ERR_WORD;
  if (bid < f.kvh * f.nsplit) {
    if (attn_decode_fused_body<true>(f, smem, bid % f.kvh, bid / f.kvh, f.nsplit, [] {})) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's slice of the attention output has been acknowledged
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  gemv_body<R, U, GEMV_RESIDUAL, true>(g, reinterpret_cast<float*>(smem), bid, nblk, [&] {
    if (threadIdx.x == 0) {
      unsigned polls = 0;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if ((++polls & 63u) == 0) {
          if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (polls > SPIN_LIMIT) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
    }
    __syncthreads();
  });
}

template <class P>
int occupancy_of(size_t lds) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, decode_step_kernel<P>, 256, lds) != hipSuccess) return 0;
  return nb;
}

}  // namespace

size_t decode_mega_lds_bytes(int H, int I, int nq) {
  const int kmax = H > I ? (H > nq ? H : nq) : (I > nq ? I : nq);
  const size_t xs = (size_t)((kmax + 511) / 512) * 512 * 4 + 64;
  return xs > (size_t)ATTN_DECODE_FUSED_LDS ? xs : (size_t)ATTN_DECODE_FUSED_LDS;
}

int decode_mega_max_blocks_per_cu(int H, size_t lds) {
  return H >= 2048 ? occupancy_of<ProfileLarge>(lds) : occupancy_of<ProfileSmall>(lds);
}

size_t attn_oproj_lds_bytes(int nq) {
  const size_t xs = (size_t)((nq + 511) / 512) * 512 * 4 + 64;
  return xs > (size_t)ATTN_DECODE_FUSED_LDS ? xs : (size_t)ATTN_DECODE_FUSED_LDS;
}

// g: the o_proj matvec (GEMV_RESIDUAL, x = f.o).  target: value the counter reaches when this launch's kvh merging
// blocks have all arrived.  Grid: the matvec's own persistent grid (>= kvh*nsplit).
void launch_attn_oproj(const AttnDecodeFusedArgs& f, const GemvArgs& g, unsigned* sync, unsigned target, hipStream_t st) {
  const int nchunks = (g.K + 511) / 512;
  const int ntiles = (g.N + 3) / 4;
  int grid = ntiles < 512 ? ntiles : 512;
  if (grid < f.kvh * f.nsplit) grid = f.kvh * f.nsplit;
  const size_t lds = attn_oproj_lds_bytes(g.K);
  if (nchunks >= 8)
    hipLaunchKernelGGL((attn_oproj_kernel<1, 8>), dim3(grid), dim3(256), lds, st, f, g, sync, target);
  else if (nchunks >= 4)
    hipLaunchKernelGGL((attn_oproj_kernel<1, 4>), dim3(grid), dim3(256), lds, st, f, g, sync, target);
  else
    hipLaunchKernelGGL((attn_oproj_kernel<1, 2>), dim3(grid), dim3(256), lds, st, f, g, sync, target);
}

void launch_decode_mega(const DecodeMegaArgs& a, int grid, size_t lds, hipStream_t st) {
  if (a.H >= 2048)
    hipLaunchKernelGGL(decode_step_kernel<ProfileLarge>, dim3(grid), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL(decode_step_kernel<ProfileSmall>, dim3(grid), dim3(256), lds, st, a);
}

}  // namespace aha
