// Decode "chain" engine: o_proj -> gate/up -> down -> (next layer's) qkv of ONE token in ONE persistent launch.
//
// The launch-per-op path spends ~40% of a decoder layer outside weight streaming (profiles/r01_decode_mega_timeline.md):
// gaps between dependent launches and, above all, the first-tile latency of every matvec -- a block cannot request
// weights of op i+1 before op i's launch has ended.  Weights do not depend on activations, so here the two are decoupled
// (the loader/consumer structure of /opt/skills/guides/MI355X_MICROARCH.md, "prefetch-credit" / "ldsdma-fill"):
//
//   * one workgroup per CU, 4 waves.  Wave 0 is the LOADER: it walks the static list of 16-KiB weight slots this CU
//     needs for all ops of the launch and DMAs them (global_load_lds, non-temporal) into an 8-slot LDS ring, running
//     ahead of the consumers by up to 128 KiB (~5 us of stream), across op boundaries.
//   * waves 1-3 are CONSUMERS: per op they gather the input vector into LDS, apply the fused RMSNorm, then take row
//     groups round-robin, dot the rows of each landed slot with the vector (same per-lane order of fmaf as
//     gemv_body.h, so results are BIT-IDENTICAL to the launch-per-op path), and run the epilogue.
//   * ops hand their output vector to every CU as 8-byte granules {2 x bf16, 32-bit tag} written with ONE
//     agent-scope (sc1, write-through) store each; consumers poll the granules themselves (no separate flag, no
//     barrier, no L2 write-back/invalidate).  The tag is launch-unique, so stale data of earlier launches never
//     matches.  Residuals that stay on the same CU (down_proj's x1 rows) never leave LDS.
//
// Row ownership: CU c owns logical rows [c * n_out/NCU, (c+1) * n_out/NCU) of every op.  A logical row is one weight
// row (K/512 pieces of 1 KiB), or for gate/up the pair (gate row j, up row j) of the fused 16-row-interleaved matrix.
// A row GROUP is G (even) logical rows = S ring slots of up to 16 pieces; one consumer wave owns a whole group.
//
// Every wait is bounded: a wave that gives up raises the error word and all loops drain (results invalid, reported).
#include <stdlib.h>

#include "common.h"
#include "gemv_body.h"
#include "model.h"

namespace aha {

namespace {

constexpr int CH_RING = 8;               // ring slots
constexpr int CH_SLOT = 16384;           // bytes per slot
constexpr int CH_PIECE = 1024;           // bytes per DMA instruction (64 lanes x 16 B) = 512 bf16 of one row
constexpr int CH_OWN_MAX = 128;          // max logical rows per CU per op kept for same-CU residuals
constexpr unsigned CH_SPIN = 1u << 22;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct OpGeom {
  int cpr;      // 512-element chunks per weight row
  int ppr;      // pieces per logical row (2 * cpr for gate/up pairs)
  int rpc;      // logical rows per CU
  int G, S;     // rows per group, slots per group
  int ngroups;  // groups per CU
};

__device__ __forceinline__ OpGeom geom_of(const ChainOp& op, int ncu) {
  OpGeom g;
  g.cpr = op.K >> 9;
  g.ppr = op.kind == GEMV_SILU_MUL ? 2 * g.cpr : g.cpr;
  g.rpc = op.n_out / ncu;
  int G = (16 / g.ppr) & ~1;
  if (G < 2) G = 2;
  if (G > g.rpc) G = g.rpc;
  g.G = G;
  g.S = (G * g.ppr + 15) >> 4;
  g.ngroups = g.rpc / G;
  return g;
}

struct Ctl {                 // LDS control block
  unsigned landed[CH_RING];  // generation of the fill that has landed in each ring slot
  unsigned done[CH_RING];    // generation of the fill the consumers have finished reading
  unsigned cbar;             // consumer-wave barrier counter
  unsigned err;              // local copy of "give up"
};

// Control words are read / written with inline asm: for an ordinary LDS access the compiler's waitcnt pass assumes it may
// alias an outstanding LDS-DMA destination and puts `s_waitcnt vmcnt(0)` in front of it, which would drain the loader's
// whole prefetch queue at every poll.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
  return v;
}
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}

// spin until *p >= target (wrap-safe); false if the launch has been told to give up
__device__ __forceinline__ bool lds_wait_ge(const unsigned* p, unsigned target, Ctl* ctl, unsigned* gerr) {
  unsigned polls = 0;
  while ((int)(lds_ld(p) - target) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if ((++polls & 255u) == 0) {
      if (lds_ld(&ctl->err) != 0u) return false;
      if (__hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || polls > CH_SPIN) {
        __hip_atomic_store(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_st(&ctl->err, 1u);
        return false;
      }
    }
  }
  return true;
}

// acc += a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (gfx950 VOP2; this compiler has no builtin for it).  One
// instruction per two MACs and no unpacking: the consumer's instruction count per 16-KiB slot drops from ~400 to ~70.
// Its internal rounding differs from two fmaf's, so this path is NOT bit-identical to gemv_body.h (tests compare it
// within the f32-accumulation tolerance; AHA_CHAIN_EXACT=1 selects the fmaf path).
__device__ __forceinline__ float dot2c_bf16(unsigned a, unsigned b, float acc) {
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}
// Last dot of a chain.  A DOT result may be read by a DIFFERENT VALU instruction only 3 wait states later
// (GCNHazardRecognizer DotWriteDifferentVALURead); the compiler inserts those for its own instructions but cannot see
// inside inline asm, so the wait states are part of the asm (without them the following v_add reads stale sums).
__device__ __forceinline__ float dot2c_bf16_last(unsigned a, unsigned b, float acc) {
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}

// wait until at most n DMA instructions of this wave are outstanding (n rounded DOWN to an encodable constant)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
  if (n >= 48) __builtin_amdgcn_s_waitcnt(0xCF70);       // vmcnt(48)
  else if (n >= 40) __builtin_amdgcn_s_waitcnt(0x8F78);  // vmcnt(40)
  else if (n >= 32) __builtin_amdgcn_s_waitcnt(0x8F70);  // vmcnt(32)
  else if (n >= 24) __builtin_amdgcn_s_waitcnt(0x4F78);  // vmcnt(24)
  else if (n >= 16) __builtin_amdgcn_s_waitcnt(0x4F70);  // vmcnt(16)
  else if (n >= 8) __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8)
  else __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0)
}

__global__ __launch_bounds__(256, 1) void decode_chain_kernel(ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                               // CH_RING x CH_SLOT
  bf16_t* vec = reinterpret_cast<bf16_t*>(smem + CH_RING * CH_SLOT);  // input vector of the current op (bf16, K)
  bf16_t* own = vec + a.kmax;                                      // [n_ops][CH_OWN_MAX] own outputs (bf16)
  Ctl* ctl = reinterpret_cast<Ctl*>(own + CH_MAX_OPS * CH_OWN_MAX);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps slot / row bookkeeping in SGPRs
  const int cu = blockIdx.x, ncu = gridDim.x;

  if (tid < CH_RING) { ctl->landed[tid] = 0; ctl->done[tid] = 0; }
  if (tid == 0) { ctl->cbar = 0; ctl->err = 0; }
  __syncthreads();

  // optional timeline (AHA_CHAIN_TRACE): blocks 0 and ncu/2+1; per op [loader first issue, loader last issue, gathered,
  // normed, rows done] in 100 MHz ticks
  const int tslot = cu == 0 ? 0 : (cu == ncu / 2 + 1 ? 1 : -1);
  auto stamp = [&](int oi, int k) {
    if (a.trace != nullptr && tslot >= 0 && lane == 0) a.trace[(tslot * CH_MAX_OPS + oi) * 5 + k] = wall_clock64();
  };
  if (wave == 0) {
    // ============================================ LOADER ============================================================
    unsigned fill = 0;
    int pend1 = 0, pend2 = 0;  // DMA instructions of fills fill-1 and fill-2 (issued, not yet confirmed)
    // n contiguous 1-KiB pieces src -> dst.  The instruction's immediate offset advances the global AND the LDS address, so
    // four pieces share one (base, M0) pair: ~2 scalar instructions per piece instead of ~40 (the loader's issue rate is
    // bound by its own instruction count: one wave issues one instruction per 4 cycles).
    auto issue_run = [&](const bf16_t* src, char* dst, int n) {
      if (a.dbg & 2) return;
      while (n >= 4) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 2 /* nt */);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 1024, 2);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 2048, 2);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 3072, 2);
        src += 2048;
        dst += 4096;
        n -= 4;
      }
      if (n >= 2) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 2);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 1024, 2);
        src += 1024;
        dst += 2048;
        n -= 2;
      }
      if (n) __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 2);
    };
    for (int oi = 0; oi < a.n_ops; ++oi) {
      const ChainOp& op = a.op[oi];
      const OpGeom g = geom_of(op, ncu);
      const bf16_t* W = (const bf16_t*)op.W + lane * 8;
      const int pieces_per_group = g.G * g.ppr;
      const int range_len = g.G * g.cpr;  // a group is 1 (plain) or 2 (gate rows, then up rows) contiguous memory ranges
      stamp(oi, 0);
      for (int grp = 0; grp < g.ngroups; ++grp) {
        const int64_t lrow0 = (int64_t)cu * g.rpc + grp * g.G;
        const bf16_t* r0;
        const bf16_t* r1 = nullptr;
        if (op.kind == GEMV_SILU_MUL) {  // fused matrix: 16-row blocks alternating gate / up; G rows stay inside one block
          const int64_t fr = (lrow0 >> 4) * 32 + (lrow0 & 15);
          r0 = W + fr * op.K;
          r1 = W + (fr + 16) * op.K;
        } else {
          r0 = W + lrow0 * op.K;
        }
        for (int s = 0; s < g.S; ++s) {
          const int rs = fill & (CH_RING - 1);
          const unsigned round = fill / CH_RING;
          if (!lds_wait_ge(&ctl->done[rs], round, ctl, a.err)) return;
          const int q0 = s * 16;
          const int np = min(16, pieces_per_group - q0);
          char* dst = ring + rs * CH_SLOT;
          // pieces q0 .. q0+np-1 of the group: range 0 holds pieces [0, range_len), range 1 the rest
          const int n0 = max(0, min(np, range_len - q0));
          if (n0 > 0) issue_run(r0 + (int64_t)q0 * 512, dst, n0);
          if (np > n0) issue_run(r1 + (int64_t)(q0 + n0 - range_len) * 512, dst + n0 * CH_PIECE, np - n0);
          // fill-3 has landed once only the instructions of the last three fills may still be in flight (the hardware
          // counter holds 63: up to four 16-instruction fills are outstanding)
          if (fill >= 3) {
            wait_vmcnt_le(pend2 + pend1 + np);
            lds_st(&ctl->landed[(fill - 3) & (CH_RING - 1)], (fill - 3) / CH_RING + 1);
          }
          pend2 = pend1;
          pend1 = np;
          ++fill;
        }
      }
      stamp(oi, 1);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): everything has landed
    for (unsigned k = 3; k >= 1; --k)
      if (fill >= k) lds_st(&ctl->landed[(fill - k) & (CH_RING - 1)], (fill - k) / CH_RING + 1);
    return;
  }

  // ============================================== CONSUMERS ==========================================================
  const int cw = wave - 1;        // 0..2
  const int ct = cw * 64 + lane;  // 0..191
  unsigned cphase = 0;
  auto cons_sync = [&]() -> bool {  // barrier among the three consumer waves (LDS counter; the loader is not involved)
    ++cphase;
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes are done
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(lds_addr(&ctl->cbar)), "v"(1u) : "memory");
    return lds_wait_ge(&ctl->cbar, 3u * cphase, ctl, a.err);
  };

  unsigned fill_base = 0;
  for (int oi = 0; oi < a.n_ops; ++oi) {
    const ChainOp& op = a.op[oi];
    const OpGeom g = geom_of(op, ncu);
    const int K = op.K;
    // norm weights are constants: request them before the gather so their latency hides behind it
    u32x4_t nwv[8];
    const bool nw_pre = op.norm_w != nullptr && (K >> 3) <= 8 * 192;
    if (nw_pre) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int v = ct + j * 192;
        nwv[j] = v < (K >> 3) ? ld16((const bf16_t*)op.norm_w + v * 8) : u32x4_t{0u, 0u, 0u, 0u};
      }
    }
    // ---- 1. input vector -> LDS (bf16) ------------------------------------------------------------------------------
    if (op.in_plain != nullptr) {  // written by an earlier launch: plain 16-byte loads
      const bf16_t* x = (const bf16_t*)op.in_plain;
      for (int v = ct; v < (K >> 3); v += 192) *reinterpret_cast<u32x4_t*>(vec + v * 8) = ld16(x + v * 8);
    } else {  // produced by the previous op of this launch on all CUs: poll the tagged granules
      const unsigned long long* gin = a.op[oi - 1].gran;
      const unsigned want = a.tag_base + (unsigned)(oi - 1);
      const int ng = K >> 1;
      for (int g0 = ct; g0 < ng; g0 += 192 * 16) {
        unsigned long long v[16];
        unsigned pending = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int gi = g0 + j * 192;
          if (gi < ng) {
            v[j] = __hip_atomic_load(gin + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pending |= 1u << j;
          }
        }
        unsigned polls = 0;
        while (pending) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            if (pending & (1u << j)) {
              if ((unsigned)(v[j] >> 32) == want) {
                *reinterpret_cast<uint32_t*>(vec + (g0 + j * 192) * 2) = (uint32_t)v[j];
                pending &= ~(1u << j);
              }
            }
          }
          if (!pending) break;
          __builtin_amdgcn_s_sleep(2);
          if ((++polls & 63u) == 0) {
            if (lds_ld(&ctl->err) != 0u) return;
            if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || polls > CH_SPIN) {
              __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              lds_st(&ctl->err, 1u);
              return;
            }
          }
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (pending & (1u << j)) v[j] = __hip_atomic_load(gin + g0 + j * 192, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (!cons_sync()) return;
    if (cw == 0) stamp(oi, 2);
    // ---- 2. fused RMSNorm (qwen3/model.rs:79,83), same summation tree as gemv_body.h: 256 virtual threads, each over
    //         its vectors v = t, t+256, ...; wave sums of the 4 virtual waves; ((s0+s1)+s2)+s3 ---------------------------
    if (op.norm_w != nullptr) {
      float part[4];
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        float ss = 0.f;
        for (int v = w4 * 64 + lane; v < (K >> 3); v += 256) {
          const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(vec + v * 8);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float f0 = lo_bf(xv[j]), f1 = hi_bf(xv[j]);
            ss = fmaf(f0, f0, ss);
            ss = fmaf(f1, f1, ss);
          }
        }
        part[w4] = wave_sum(ss);
      }
      const float tot = part[0] + part[1] + part[2] + part[3];
      const float rinv = 1.0f / sqrtf(tot / (float)K + op.eps);
      if (!cons_sync()) return;  // every wave has read the raw vector before anyone overwrites it
      const bf16_t* nw = (const bf16_t*)op.norm_w;
      auto norm8 = [&](int v, u32x4_t wv) {
        const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(vec + v * 8);
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = pack_bf(rbf(lo_bf(xv[j]) * rinv * lo_bf(wv[j])), rbf(hi_bf(xv[j]) * rinv * hi_bf(wv[j])));
        *reinterpret_cast<u32x4_t*>(vec + v * 8) = o;
      };
      if (nw_pre) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int v = ct + j * 192;
          if (v < (K >> 3)) norm8(v, nwv[j]);
        }
      } else {
        for (int v = ct; v < (K >> 3); v += 192) norm8(v, ld16(nw + v * 8));
      }
      if (!cons_sync()) return;
    }
    if (cw == 0) stamp(oi, 3);
    // ---- 3. row groups -------------------------------------------------------------------------------------------------
    const unsigned tag = a.tag_base + (unsigned)oi;
    for (int grp = cw; grp < g.ngroups; grp += 3) {
      // Pieces arrive in order: rows 0..G-1 (gate rows for gate/up), then the G up rows, chunk by chunk: one running
      // accumulator per lane; a completed row is reduced across the wave and kept by lane (row index) in v0 (v1: up rows).
      float cur = 0.f, v0 = 0.f, v1 = 0.f;
      int left = g.cpr, ai = 0, chunk = 0;
      const int pieces_per_group = g.G * g.ppr;
      for (int s = 0; s < g.S; ++s) {
        const unsigned f = fill_base + (unsigned)(grp * g.S + s);
        const int rs = f & (CH_RING - 1);
        const bool tr = a.trace != nullptr && tslot == 0 && oi == 1 && cw == 0 && grp < 24 && lane == 0;
        if (tr) a.trace[2 * CH_MAX_OPS * 5 + (grp / 3 * 2 + s) * 3 + 0] = wall_clock64();
        if (!lds_wait_ge(&ctl->landed[rs], f / CH_RING + 1, ctl, a.err)) return;
        if (tr) a.trace[2 * CH_MAX_OPS * 5 + (grp / 3 * 2 + s) * 3 + 1] = wall_clock64();
        const char* src = ring + rs * CH_SLOT;
        const int np = min(16, pieces_per_group - s * 16);
        // the chunk index of every piece of the slot is known up front: all 32 LDS reads can be in flight together
        auto row_done = [&]() {  // a row('s half) is complete: reduce across the wave, park it with its lane
          const float tot = wave_sum(cur);
          if (ai < g.G) {
            if (lane == ai) v0 = tot;
          } else if (lane == ai - g.G) {
            v1 = tot;
          }
          ++ai;
          cur = 0.f;
          left = g.cpr;
          chunk = 0;
        };
        if (np == 16 && (g.cpr & 7) == 0 && !(a.dbg & 1)) {
          // fast path (K a multiple of 4096): rows end only at multiples of 8 pieces -> two straight-line halves, all 32
          // LDS reads of the slot in flight before the first fma
          const int c1 = left == 8 ? 0 : chunk + 8;  // chunk index of the second half
          u32x4_t wreg[16], xreg[16];
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            const int cp = p < 8 ? chunk + p : c1 + (p - 8);
            wreg[p] = *reinterpret_cast<const u32x4_t*>(src + p * CH_PIECE + lane * 16);
            xreg[p] = *reinterpret_cast<const u32x4_t*>(vec + cp * 512 + lane * 8);
          }
          if (a.exact) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
              for (int p = h * 8; p < h * 8 + 8; ++p) {
                const u32x4_t w = wreg[p], xv = xreg[p];
                cur = fmaf(lo_bf(w[0]), lo_bf(xv[0]), cur); cur = fmaf(hi_bf(w[0]), hi_bf(xv[0]), cur);
                cur = fmaf(lo_bf(w[1]), lo_bf(xv[1]), cur); cur = fmaf(hi_bf(w[1]), hi_bf(xv[1]), cur);
                cur = fmaf(lo_bf(w[2]), lo_bf(xv[2]), cur); cur = fmaf(hi_bf(w[2]), hi_bf(xv[2]), cur);
                cur = fmaf(lo_bf(w[3]), lo_bf(xv[3]), cur); cur = fmaf(hi_bf(w[3]), hi_bf(xv[3]), cur);
              }
              chunk += 8;
              left -= 8;
              if (left == 0) row_done();
            }
          } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;  // four independent chains, summed by ordinary adds
#pragma unroll
              for (int p = h * 8; p < h * 8 + 8; ++p) {
                c0 = dot2c_bf16(wreg[p][0], xreg[p][0], c0);
                c1 = dot2c_bf16(wreg[p][1], xreg[p][1], c1);
                c2 = dot2c_bf16(wreg[p][2], xreg[p][2], c2);
                c3 = p == h * 8 + 7 ? dot2c_bf16_last(wreg[p][3], xreg[p][3], c3) : dot2c_bf16(wreg[p][3], xreg[p][3], c3);
              }
              cur += (c0 + c1) + (c2 + c3);
              chunk += 8;
              left -= 8;
              if (left == 0) row_done();
            }
          }
        } else {
          u32x4_t wreg[16], xreg[16];
          int cp = chunk;  // chunk index of piece p: runs on from the previous slot, restarts at 0 every cpr pieces
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            if (p < np && !(a.dbg & 1)) {
              wreg[p] = *reinterpret_cast<const u32x4_t*>(src + p * CH_PIECE + lane * 16);
              xreg[p] = *reinterpret_cast<const u32x4_t*>(vec + cp * 512 + lane * 8);
            }
            if (++cp == g.cpr) cp = 0;
          }
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            if (p < np && !(a.dbg & 1)) {
              const u32x4_t w = wreg[p], xv = xreg[p];
              cur = fmaf(lo_bf(w[0]), lo_bf(xv[0]), cur); cur = fmaf(hi_bf(w[0]), hi_bf(xv[0]), cur);
              cur = fmaf(lo_bf(w[1]), lo_bf(xv[1]), cur); cur = fmaf(hi_bf(w[1]), hi_bf(xv[1]), cur);
              cur = fmaf(lo_bf(w[2]), lo_bf(xv[2]), cur); cur = fmaf(hi_bf(w[2]), hi_bf(xv[2]), cur);
              cur = fmaf(lo_bf(w[3]), lo_bf(xv[3]), cur); cur = fmaf(hi_bf(w[3]), hi_bf(xv[3]), cur);
              ++chunk;
              if (--left == 0) row_done();
            }
          }
        }
        if (lane == 0) lds_st(&ctl->done[rs], f / CH_RING + 1);  // all reads of the slot have been consumed above
        if (tr) a.trace[2 * CH_MAX_OPS * 5 + (grp / 3 * 2 + s) * 3 + 2] = wall_clock64();
      }
      // epilogue: lane r < G owns logical row r of the group; the reference's rounding chain; one granule per row pair
      float out = 0.f;
      const int lrow = grp * g.G + lane;
      const int64_t row = (int64_t)cu * g.rpc + lrow;
      if (lane < g.G) {
        const float lin = rbf(v0);  // candle_nn::Linear output tensor (bf16)
        if (op.kind == GEMV_SILU_MUL) {
          const float gte = rbf(silu_f(lin));  // gate_proj -> act_fn   (modules.rs:82)
          const float up = rbf(v1);            // up_proj               (modules.rs:83)
          out = rbf(gte * up);                 // lhs * rhs             (modules.rs:84)
        } else if (op.kind == GEMV_RESIDUAL) {
          const float res = op.res_plain != nullptr ? bf2f(((const bf16_t*)op.res_plain)[row])
                                                    : bf2f(own[op.res_own_op * CH_OWN_MAX + lrow]);
          out = rbf(res + lin);
        } else {
          out = lin;
        }
        own[oi * CH_OWN_MAX + lrow] = f2bf(out);
        if (op.out_plain != nullptr) ((bf16_t*)op.out_plain)[row] = f2bf(out);
      }
      const float nxt = __shfl_down(out, 1, 64);
      if (op.gran != nullptr && lane < g.G && (lane & 1) == 0) {
        const unsigned long long v = (unsigned long long)pack_bf(out, nxt) | ((unsigned long long)tag << 32);
        __hip_atomic_store(op.gran + (row >> 1), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    fill_base += (unsigned)(g.ngroups * g.S);
    if (!cons_sync()) return;  // vec / own are reused by the next op
    if (cw == 0) stamp(oi, 4);
  }
}

}  // namespace

size_t decode_chain_lds_bytes(int kmax) {
  return (size_t)CH_RING * CH_SLOT + (size_t)kmax * 2 + (size_t)CH_MAX_OPS * CH_OWN_MAX * 2 + sizeof(Ctl) + 64;
}

// Can this op run on the engine with `ncu` workgroups?  (K in 512-element pieces, an even number of rows per CU that the
// group size divides, at most 8 rows per group, rows kept for same-CU residuals fit.)
bool decode_chain_op_ok(int n_out, int K, int kind, int ncu) {
  if (K % 512 != 0 || n_out % ncu != 0) return false;
  const int cpr = K / 512, ppr = kind == GEMV_SILU_MUL ? 2 * cpr : cpr, rpc = n_out / ncu;
  if (rpc < 2 || rpc > CH_OWN_MAX) return false;
  int G = (16 / ppr) & ~1;
  if (G < 2) G = 2;
  if (G > rpc) G = rpc;
  if (G > 8 || (G & 1) || rpc % G != 0) return false;
  return true;
}

void launch_decode_chain(const ChainArgs& a, int ncu, hipStream_t st) {
  const size_t lds = decode_chain_lds_bytes(a.kmax);
  static bool once = false;
  if (!once) {
    hipFuncSetAttribute((const void*)decode_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    once = true;
  }
  hipLaunchKernelGGL(decode_chain_kernel, dim3(ncu), dim3(256), lds, st, a);
}

}  // namespace aha
