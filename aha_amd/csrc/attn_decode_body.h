// Body of the fused decode attention kernel (attn_decode_fused_kernel, kernels_attn.hip).
#pragma once
#include <type_traits>
#include "attn_common.h"

namespace aha {

constexpr int ATTN_DECODE_NW = 4;  // waves (= KV units) per workgroup
constexpr int ATTN_DECODE_FUSED_LDS = 16 * 128 * 2 + 128 * 2 + 128 * 2 + ATTN_DECODE_NW * 128 * 16 * 4 + 2 * ATTN_DECODE_NW * 16 * 4;  // bytes

// ---- decode, fused: q/k RMSNorm + (M-)RoPE + KV append + split-KV attention + in-block merge --------------------
// One launch replaces qknorm_rope_kernel + attn_decode_kernel + the combine: every block redoes the (tiny) norm/rope of
// its kv head's g query heads and of the new key in LDS (QKNormAttention::forward, modules.rs:538-557), block (kvhd, 0)
// appends the new K/V to the cache page (modules.rs:558-566), all blocks attend over the OLD tokens from the pages, unit
// 0 adds the new token from LDS, the 4 waves of a block are merged through LDS, one un-normalised partial per
// (split, head) is published, and the LAST split block of the kv head to arrive merges the splits and writes the bf16
// attention output.
//
// Latency structure (batch 1: the launch is a chain of dependent round trips, not bandwidth):
//   * every length-dependent scalar (cache length, slot, rope table of the step) arrives as a kernel ARGUMENT or from a
//     table the step's first kernel wrote -- no dependent scalar loads, no sincos, in front of the first page request;
//   * the prologue's own inputs (qkv, norm weights, rope table: L2 hits) are requested first, the unit's first page
//     right behind them (returns are in order within a wave), and the norm/rope arithmetic runs under the page fetch;
//   * in the page loop K and V of a page are separate register buffers re-requested as soon as their MFMAs have
//     consumed them (K(next) right after QK^T, V(next) right after P.V), so a wave always has 16-32 KB in flight.
// smem: ATTN_DECODE_FUSED_LDS bytes, 16-byte aligned.  Returns true in the one block per kv head that wrote the output.
__device__ __forceinline__ bool attn_decode_fused_body(const AttnDecodeFusedArgs& a, char* smem, const int kvhd, const int split,
                                                       const int nsplit) {
  constexpr int NW = ATTN_DECODE_NW, NT = NW * 64;
  bf16_t* qs = reinterpret_cast<bf16_t*>(smem);                 // [16][128]
  bf16_t* ksn = qs + 16 * 128;                                  // [128]
  bf16_t* vsn = ksn + 128;                                      // [128]
  float* mo = reinterpret_cast<float*>(vsn + 128);              // per wave O^T [NW][d 128][q 16]
  float* mm = mo + NW * 128 * 16;                               // [NW][16]
  float* mlz = mm + NW * 16;                                    // [NW][16]
  const int tid = (int)threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, G = lane >> 4, c = lane & 15;
  const int g = a.nh / a.kvh;
  const int nunits = nsplit * NW, unit = split * NW + wave;
  const int L = a.kv_len_v, slot_new = a.kv_start_v;
  const int L_old = L - 1;  // tokens already in the pages; the new one is handled from LDS
  const int npages = (L_old + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;

  // optional timeline (AHA_ATTN_TRACE): block (kv head 0, split 0) and (kv head 0, last split), thread 0, 100 MHz stamps:
  // start, prologue done, pages done, partial published, arrival known, end
  const int tslot = (kvhd == 0 && split == 0) ? 0 : (kvhd == 0 && split == nsplit - 1 ? 1 : -1);
  auto stamp = [&](int k) {
    if (a.trace != nullptr && tslot >= 0 && tid == 0) a.trace[tslot * 6 + k] = wall_clock64();
  };
  stamp(0);

  // ---- requests, in the order their data is needed ------------------------------------------------------------------
  const bf16_t* qkv = (const bf16_t*)a.qkv;
  // first (for most waves: only) prologue item of this wave: hs < g: q head kvhd*g+hs ; hs == g: the k head
  const bool p_is_k = wave == g;
  const bool p_have = wave <= g;
  const bf16_t* p_src = p_is_k ? qkv + (int64_t)(a.nh + kvhd) * 128 : qkv + (int64_t)(kvhd * g + min(wave, g - 1)) * 128;
  const bf16_t* p_nw = (const bf16_t*)(p_is_k ? a.k_norm_w : a.q_norm_w);
  const bf16_t px0 = p_src[lane], px1 = p_src[lane + 64];
  const bf16_t pw0 = p_nw[lane], pw1 = p_nw[lane + 64];
  const float cs = a.rope[lane], sn = a.rope[64 + lane];   // bf16-representable values (rope_step_kernel)
  bf16_t vnew = 0;
  if (tid < 128) vnew = qkv[(int64_t)(a.nh + a.kvh + kvhd) * 128 + tid];

  u32x4_t kf[4][4], vf[8][2];
  // Addresses: wave-uniform 64-bit base (scalar registers) + ONE 32-bit per-lane offset (lane * 16), the rest immediates:
  // pages are fragment-major (common.h), every load is `global_load_dwordx4 v, v_off, s[base] offset:frag*1024` = 1 KB
  // contiguous per wave instruction.
  typedef const __attribute__((address_space(1))) char* gchar_t;
  const uint32_t l_off = (uint32_t)lane * 16;
  auto load_k = [&](uint64_t base) {
    gchar_t kb = reinterpret_cast<gchar_t>(base + (uint64_t)kvhd * KV_PAGE_TOKENS * 256);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
        kf[sub][k4] = __builtin_nontemporal_load(reinterpret_cast<gptr16_t>(kb + (sub * 4 + k4) * 1024 + l_off));
  };
  auto load_v = [&](uint64_t base) {
    gchar_t vb = reinterpret_cast<gchar_t>(base + (uint64_t)a.kvh * KV_PAGE_TOKENS * 256 + (uint64_t)kvhd * 128 * (KV_PAGE_TOKENS * 2));
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        vf[ds][kk] = __builtin_nontemporal_load(reinterpret_cast<gptr16_t>(vb + (ds * 2 + kk) * 1024 + l_off));
  };
  // This unit's page pointers (pages unit, unit + nunits, ...): ONE vector load up front, lane i holding the i-th of them,
  // broadcast per iteration with v_readlane.  (A load of page_ptrs[page] inside the loop is a VECTOR load -- the kernel also
  // stores, so the compiler will not use the scalar cache -- and its wait drains the K/V requests queued behind it.)
  uint64_t my_pages = 0;
  auto fetch_page_ptrs = [&](int first_it) {
    const int pg = unit + (first_it + lane) * nunits;
    my_pages = pg < npages ? (uint64_t)(a.kv.page_ptrs[pg] + a.kv.layer_off) : 0;
  };
  auto page_base = [&](int i) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)my_pages, i & 63), hi = __builtin_amdgcn_readlane((uint32_t)(my_pages >> 32), i & 63);
    return ((uint64_t)hi << 32) | lo;
  };
  fetch_page_ptrs(0);
  int page = unit, it = 0;
  if (page < npages) {
    const uint64_t b0 = page_base(0);
    __builtin_amdgcn_sched_barrier(0);  // K strictly ahead of V (as in the loop): the loop's counted waits assume that order
    load_k(b0);
    __builtin_amdgcn_sched_barrier(0);
    load_v(b0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- prologue: norm + rope of the g q heads and the k head; v raw ------------------------------------------------
  {
    auto norm_rope = [&](bf16_t bx0, bf16_t bx1, bf16_t bw0, bf16_t bw1, bf16_t* dst) {
      float x0 = bf2f(bx0), x1 = bf2f(bx1);
      const float ss = wave_sum(fmaf(x0, x0, x1 * x1));  // explicit: `a*a + b*b` can be fused two ways
      const float rinv = 1.0f / sqrtf(ss / 128.0f + a.eps);
      x0 = rbf(x0 * rinv * bf2f(bw0));
      x1 = rbf(x1 * rinv * bf2f(bw1));
      const bf16_t y0 = f2bf(rbf(x0 * cs) + rbf(-x1 * sn));
      const bf16_t y1 = f2bf(rbf(x1 * cs) + rbf(x0 * sn));
      dst[lane] = y0;
      dst[lane + 64] = y1;
    };
    if (p_have) norm_rope(px0, px1, pw0, pw1, p_is_k ? ksn : qs + wave * 128);
    for (int hs = wave + NW; hs <= g; hs += NW) {  // more heads than waves (g >= 4): a second round for some waves
      const bool is_k = hs == g;
      const bf16_t* src = is_k ? qkv + (int64_t)(a.nh + kvhd) * 128 : qkv + (int64_t)(kvhd * g + hs) * 128;
      const bf16_t* nw = (const bf16_t*)(is_k ? a.k_norm_w : a.q_norm_w);
      norm_rope(src[lane], src[lane + 64], nw[lane], nw[lane + 64], is_k ? ksn : qs + hs * 128);
    }
    if (tid < 128) vsn[tid] = vnew;
  }
  __syncthreads();
  stamp(1);
  if (split == 0) {  // append (k roped, v raw) for the following steps
    // global address space spelled out: a flat store here would make every later wait in the kernel a vmcnt(0)
    typedef __attribute__((address_space(1))) bf16_t* gbf_t;
    const int pg = slot_new / KV_PAGE_TOKENS, t = slot_new % KV_PAGE_TOKENS;
    const uint64_t base = (uint64_t)(a.kv.page_ptrs[pg] + a.kv.layer_off);
    gbf_t kd = reinterpret_cast<gbf_t>(base) + (int64_t)kvhd * KV_PAGE_TOKENS * 128;
    gbf_t vd = reinterpret_cast<gbf_t>(base) + (int64_t)a.kvh * KV_PAGE_TOKENS * 128 + (int64_t)kvhd * 128 * KV_PAGE_TOKENS;
    for (int i = tid; i < 128; i += NT) {
      kd[kpage_elem(t, i, 4)] = ksn[i];
      vd[vpage_elem(t, i)] = vsn[i];
    }
  }

  bf16x8_t qf[4];
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    u32x4_t v = *reinterpret_cast<const u32x4_t*>(qs + min(c, g - 1) * 128 + k4 * 32 + G * 8);
    if (c >= g) v = u32x4_t{0u, 0u, 0u, 0u};
    qf[k4] = as_frag(v);
  }
  float m = -INFINITY, l = 0.f;
  f32x4_t o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // One page of this unit.  MORE (compile time): the unit has another page after this one -- its K is requested right
  // after QK^T and its V right after P.V, unconditionally, so the register buffers are plain loop-carried values (a
  // run-time `if (more) load` makes them phis that the compiler resolves with copies and a vmcnt(0) at the loop end).
  auto do_page = [&](auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    uint64_t nb = 0;
    if (MORE) {
      ++it;
      if ((it & 63) == 0) fetch_page_ptrs(it);  // > 64 pages per unit: next batch of pointers (contexts beyond 1 M tokens)
      nb = page_base(it);
    }
    f32x4_t st[4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      st[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) st[sub] = mfma16(as_frag(kf[sub][k4]), qf[k4], st[sub]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MORE) load_k(nb);   // K of the unit's next page: in flight during softmax + P.V of this one
    __builtin_amdgcn_sched_barrier(0);
    float alpha;
    bf16x8_t pf[2];
    const int t0 = page * KV_PAGE_TOKENS;
    softmax_tile(st, a.scale, [&](int t) { return t0 + t < L_old; }, G, m, l, alpha, pf);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      o[ds] *= alpha;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) o[ds] = mfma16(as_frag(vf[ds][kk]), pf[kk], o[ds]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MORE) load_v(nb);   // V of the next page: in flight during its QK^T and softmax
    __builtin_amdgcn_sched_barrier(0);
    page += nunits;
  };
  if (page < npages) {
    while (page + nunits < npages) do_page(std::true_type{});
    do_page(std::false_type{});
  }
  stamp(2);
  if (unit == 0) {  // the new token: score from LDS, one more online-softmax step
    float dot = 0.f;
    const bf16_t* qr = qs + min(c, g - 1) * 128 + G * 32;
    const bf16_t* kr = ksn + G * 32;
#pragma unroll
    for (int j = 0; j < 32; ++j) dot = fmaf(bf2f(qr[j]), bf2f(kr[j]), dot);
    dot = group_sum(dot);
    const float s = rbf(rbf(dot) * a.scale);
    const float m_new = fmaxf(m, s);
    const float alpha = __expf(m - m_new);
    const float p = rbf(__expf(s - m_new));  // P feeds the MFMA as bf16 on the page path: same rounding here
    l = l * alpha + (G == 0 ? __expf(s - m_new) : 0.f);
    m = m_new;
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[ds][r] = fmaf(p, bf2f(vsn[ds * 16 + G * 4 + r]), o[ds][r] * alpha);
  }
  l = group_sum(l);

  // ---- merge the 4 waves through LDS, leave one partial per (split, head) ---------------------------------------------
  {
    float* wo = mo + wave * (128 * 16);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
#pragma unroll
      for (int r = 0; r < 4; ++r) wo[(ds * 16 + G * 4 + r) * 16 + c] = o[ds][r];
    if (G == 0) {
      mm[wave * 16 + c] = m;
      mlz[wave * 16 + c] = l;
    }
  }
  __syncthreads();
  const bool single = nsplit == 1;
  // one thread per (head, 4 dims): the published partial is ONE 16-byte write-through store per thread (a dword sc1 store is
  // one fabric write each: ~6x the time per byte of a dwordx4 one)
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(a.part_o, 0, (int)((size_t)nsplit * a.nh * 128 * 4), 0x00020000);
  for (int item = tid; item < g * 32; item += NT) {
    const int q = item >> 5, d0 = (item & 31) * 4;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, mm[w * 16 + q]);
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, ls = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float mw = mm[w * 16 + q];
      const float wt = (mw == -INFINITY) ? 0.f : __expf(mw - M);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(wt, mo[w * (128 * 16) + (d0 + e) * 16 + q], acc[e]);
      ls = fmaf(wt, mlz[w * 16 + q], ls);
    }
    const int head = kvhd * g + q;
    if (single) {  // the whole cache went through this block: normalise and write the attention output tensor (bf16)
      const float inv = 1.0f / ls;
      uint2 w2;
      w2.x = pack_bf(acc[0] * inv, acc[1] * inv);
      w2.y = pack_bf(acc[2] * inv, acc[3] * inv);
      *reinterpret_cast<uint2*>((bf16_t*)a.o + head * 128 + d0) = w2;
    } else {
      const u32x4_t v = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_w, (uint32_t)(((size_t)split * a.nh + head) * 128 + d0) * 4, 0, 16 /* sc1 */);
      if (d0 == 0) {
        const unsigned long long mlp = ((unsigned long long)__float_as_uint(ls) << 32) | __float_as_uint(M);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.part_ml + ((int64_t)split * a.nh + head) * 2), mlp, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (single) {
    stamp(5);
    return true;
  }
  stamp(3);

  // ---- the LAST split block of this kv head to finish merges all splits of its g heads --------------------------------
  // Partials cross blocks (possibly XCDs) inside one launch: write-through (agent-scope) stores above, every wave waits for
  // their acknowledgement, then one relaxed atomic per block on the head's counter decides who arrived last; the reader
  // uses agent-scope loads (no fence needed on either side: cdna guide G16 R1, sc1 payload both sides).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* s_last = reinterpret_cast<int*>(qs);  // q fragments are in registers since the page loop; LDS region is free
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(a.head_ctr + 32 * kvhd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = (prev + 1u == a.ctr_target) ? 1 : 0;
  }
  __syncthreads();
  stamp(4);
  if (*s_last == 0) return false;
  // The merging block is ONE workgroup, so this phase is bound by how many load instructions it issues and by dependent round
  // trips to memory (the partials were published write-through: they are in memory, not in any L2).  Hence: 16-byte
  // write-through-coherent (sc1) buffer loads, one thread per (head, 4 dims) -- nsplit wave-loads per wave instead of
  // 4 * nsplit -- and ONE round for up to 16 splits: the (max, sum) pairs of all splits of the g heads (shared through LDS)
  // and the thread's first 16 partial vectors are requested together; the global max, the split weights and 1/sum are
  // derived from LDS while the vectors are in flight.
  float* w_lds = mo;               // [g][nsplit] weights  (the per-wave O^T staging area is free again)
  float* inv_lds = mo + 16 * 256;  // [g] 1 / sum
  float2* ml_lds = reinterpret_cast<float2*>(mo + 8 * 256);  // [g][nsplit] (max, sum)
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.part_o, 0, (int)((size_t)nsplit * a.nh * 128 * 4), 0x00020000);
  const int nvec = g * 32;                      // (head, 4-dim group) items of this kv head
  const uint32_t split_stride = (uint32_t)a.nh * 128 * 4;
  f32x4_t pv[16];
  auto load_batch = [&](int item, int s0) {     // partial vectors of splits s0 .. s0 + 15 (clamped) for `item`
    const uint32_t off = (uint32_t)((kvhd * g + (item >> 5)) * 128 + (item & 31) * 4) * 4;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + (uint32_t)min(s0 + j, nsplit - 1) * split_stride, 0, 16 /* sc1 */);
      pv[j] = f32x4_t{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
  };
  {
    float2 mlv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // g * nsplit <= 1024 (host check): up to 4 pairs per thread
      const int i = tid + r * NT;
      const int q = min(i / nsplit, g - 1), sidx = i - (i / nsplit) * nsplit;
      mlv[r] = act_ldf2<true>(a.part_ml + ((size_t)sidx * a.nh + kvhd * g + q) * 2);
    }
    if (tid < nvec) load_batch(tid, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = tid + r * NT;
      if (i < g * nsplit) ml_lds[i] = mlv[r];
    }
  }
  __syncthreads();
  for (int q = wave; q < g; q += NW) {  // one wave per head
    float M = -INFINITY;
    for (int sidx = lane; sidx < nsplit; sidx += 64) M = fmaxf(M, ml_lds[q * nsplit + sidx].x);
    M = wave_max(M);
    float ls = 0.f;
    for (int sidx = lane; sidx < nsplit; sidx += 64) {
      const float2 v = ml_lds[q * nsplit + sidx];
      const float wgt = (v.x == -INFINITY) ? 0.f : __expf(v.x - M);
      w_lds[q * nsplit + sidx] = wgt;
      ls = fmaf(wgt, v.y, ls);
    }
    ls = wave_sum(ls);
    if (lane == 0) inv_lds[q] = 1.0f / ls;
  }
  __syncthreads();
  for (int item = tid; item < nvec; item += NT) {
    const int q = item >> 5;
    const float* wq = w_lds + q * nsplit;
    f32x4_t f = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < nsplit; s0 += 16) {
      if (item != tid || s0 > 0) load_batch(item, s0);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float wgt = (s0 + j < nsplit) ? wq[s0 + j] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = fmaf(wgt, pv[j][e], f[e]);
      }
    }
    const float inv = inv_lds[q];
    uint2 w;
    w.x = pack_bf(f[0] * inv, f[1] * inv);
    w.y = pack_bf(f[2] * inv, f[3] * inv);
    *reinterpret_cast<uint2*>((bf16_t*)a.o + (kvhd * g) * 128 + item * 4) = w;   // attention output tensor (bf16)
  }
  stamp(5);
  return true;
}

}  // namespace aha
