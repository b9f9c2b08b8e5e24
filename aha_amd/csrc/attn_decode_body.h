// Body of the fused decode attention, shared by attn_decode_fused_kernel (kernels_attn.hip) and the persistent
// decode-step kernel (decode_mega.hip).
#pragma once
#include "attn_common.h"

namespace aha {

constexpr int ATTN_DECODE_FUSED_LDS = 16 * 128 * 2 + 128 * 2 + 128 * 2 + 4 * 128 * 16 * 4 + 2 * 4 * 16 * 4;  // bytes

// ---- decode, fused: q/k RMSNorm + (M-)RoPE + KV append + split-KV attention + in-block merge --------------------
// One launch replaces qknorm_rope_kernel + attn_decode_kernel + most of the combine: every block redoes the (tiny)
// norm/rope of its kv head's g query heads and of the new key in LDS (QKNormAttention::forward, modules.rs:538-557),
// block (kvhd, 0) appends the new K/V to the cache page (modules.rs:558-566), all blocks attend over the OLD tokens
// from the pages, unit 0 adds the new token from LDS, the 4 waves of a block are merged through LDS, and one
// un-normalised partial per (split, head) is left for the o_proj matvec's prologue to merge (kernels_gemv.hip).
// smem: ATTN_DECODE_FUSED_LDS bytes, 16-byte aligned.  (kvhd, split) of nsplit: this block's KV head and KV split.
// after_prefetch() runs after the unit's first page has been requested and before qkv is read.
// Returns true in the one block per kv head that wrote the final attention output of the head's g query heads.
// General form: NW waves cooperate as one "block" (wave = 0..NW-1, tid = wave * 64 + lane); sync() is a barrier among exactly
// those waves (__syncthreads() for a whole workgroup; the chain engine's consumer-wave barrier otherwise); emit(head, d, v)
// receives the final attention output (called for consecutive d by consecutive tid).  LDS: attn_decode_lds_bytes(NW).
constexpr int attn_decode_lds_bytes(int nw) { return 16 * 128 * 2 + 128 * 2 + 128 * 2 + nw * 128 * 16 * 4 + 2 * nw * 16 * 4; }

template <bool COH, int NW, class AfterPrefetch, class Sync, class Emit>
__device__ __forceinline__ bool attn_decode_fused_body_t(const AttnDecodeFusedArgs& a, char* smem, const int kvhd, const int split,
                                                         const int nsplit, const int wave, const int tid,
                                                         AfterPrefetch&& after_prefetch, Sync&& sync, Emit&& emit) {
  constexpr int NT = NW * 64;
  bf16_t* qs = reinterpret_cast<bf16_t*>(smem);                 // [16][128]
  bf16_t* ksn = qs + 16 * 128;                                  // [128]
  bf16_t* vsn = ksn + 128;                                      // [128]
  float* mo = reinterpret_cast<float*>(vsn + 128);              // per wave O^T [NW][d 128][q 16]
  float* mm = mo + NW * 128 * 16;                               // [NW][16]
  float* mlz = mm + NW * 16;                                    // [NW][16]
  const int lane = tid & 63, G = lane >> 4, c = lane & 15;
  const int g = a.nh / a.kvh;
  const int nunits = nsplit * NW, unit = split * NW + wave;
  const int L = *a.kv_len, slot_new = *a.kv_start;
  const int L_old = L - 1;  // tokens already in the pages; the new one is handled from LDS
  const int npages = (L_old + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;

  // This unit's first page goes out BEFORE the norm/rope prologue: its loads do not depend on q, and the prologue's
  // own dependent chain (qkv -> wave_sum -> sincos -> LDS) then overlaps the page fetch instead of preceding it.
  u32x4_t kf[4][4], vf[8][2];
  auto load_page = [&](int page) {
    const char* base = reinterpret_cast<const char*>(a.kv.page_ptrs[page] + a.kv.layer_off);
    const char* kb = base + (size_t)kvhd * KV_PAGE_TOKENS * 256;
    const char* vb = base + (size_t)a.kvh * KV_PAGE_TOKENS * 256 + (size_t)kvhd * 128 * (KV_PAGE_TOKENS * 2);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) kf[sub][k4] = ld_nt16(kb + (size_t)(sub * 16 + c) * 256 + (k4 * 32 + G * 8) * 2);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) vf[ds][kk] = ld_nt16(vb + (size_t)(ds * 16 + c) * (KV_PAGE_TOKENS * 2) + (kk * 32 + G * 8) * 2);
  };
  // optional timeline (AHA_ATTN_TRACE): block (kv head 0, split 0) and (kv head 0, last split), thread 0, 100 MHz stamps:
  // start, prologue done, pages done, partial published, arrival known, end
  const int tslot = (kvhd == 0 && split == 0) ? 0 : (kvhd == 0 && split == nsplit - 1 ? 1 : -1);
  auto stamp = [&](int k) {
    if (a.trace != nullptr && tslot >= 0 && tid == 0) a.trace[tslot * 6 + k] = wall_clock64();
  };
  stamp(0);
  int page = unit;
  if (page < npages) load_page(page);
  after_prefetch();  // grid barrier of the persistent decode kernel: qkv of this step is complete past this point

  // ---- prologue: norm + rope of the g q heads and the k head; v raw ------------------------------------------------
  {
    const bf16_t* qkv = (const bf16_t*)a.qkv;
    for (int hs = wave; hs <= g; hs += NW) {  // hs < g: q head kvhd*g+hs ; hs == g: the k head
      const bool is_k = hs == g;
      const bf16_t* src = is_k ? qkv + (int64_t)(a.nh + kvhd) * 128 : qkv + (int64_t)(kvhd * g + hs) * 128;
      const bf16_t* nw = (const bf16_t*)(is_k ? a.k_norm_w : a.q_norm_w);
      float x0 = bf2f(act_ld_bf<COH>(src + lane)), x1 = bf2f(act_ld_bf<COH>(src + lane + 64));
      const float ss = wave_sum(fmaf(x0, x0, x1 * x1));  // explicit: `a*a + b*b` can be fused two ways
      const float rinv = 1.0f / sqrtf(ss / 128.0f + a.eps);
      x0 = rbf(x0 * rinv * bf2f(nw[lane]));
      x1 = rbf(x1 * rinv * bf2f(nw[lane + 64]));
      const float ang = (float)a.pos[a.axis_map[lane]] * a.inv_freq[lane];
      const float cs = rbf(cosf(ang)), sn = rbf(sinf(ang));
      const bf16_t y0 = f2bf(rbf(x0 * cs) + rbf(-x1 * sn));
      const bf16_t y1 = f2bf(rbf(x1 * cs) + rbf(x0 * sn));
      bf16_t* dst = is_k ? ksn : qs + hs * 128;
      dst[lane] = y0;
      dst[lane + 64] = y1;
    }
    for (int i = tid; i < 128; i += NT) vsn[i] = act_ld_bf<COH>(qkv + (int64_t)(a.nh + a.kvh + kvhd) * 128 + i);
  }
  sync();
  stamp(1);
  if (split == 0) {  // append (k roped, v raw) for the following steps
    const int pg = slot_new / KV_PAGE_TOKENS, t = slot_new % KV_PAGE_TOKENS;
    bf16_t* base = reinterpret_cast<bf16_t*>(a.kv.page_ptrs[pg] + a.kv.layer_off);
    bf16_t* vd = base + (int64_t)a.kvh * KV_PAGE_TOKENS * 128 + (int64_t)kvhd * 128 * KV_PAGE_TOKENS;
    for (int i = tid; i < 128; i += NT) {
      base[((int64_t)kvhd * KV_PAGE_TOKENS + t) * 128 + i] = ksn[i];
      vd[(int64_t)i * KV_PAGE_TOKENS + v_slot(t)] = vsn[i];
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

  while (page < npages) {
    f32x4_t st[4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      st[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) st[sub] = mfma16(as_frag(kf[sub][k4]), qf[k4], st[sub]);
    }
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
    page += nunits;
    if (page < npages) load_page(page);
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
  sync();
  const bool single = nsplit == 1;
  for (int it = tid; it < g * 128; it += NT) {
    const int q = it >> 7, d = it & 127;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, mm[w * 16 + q]);
    float acc = 0.f, ls = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float mw = mm[w * 16 + q];
      const float wt = (mw == -INFINITY) ? 0.f : __expf(mw - M);
      acc = fmaf(wt, mo[w * (128 * 16) + d * 16 + q], acc);
      ls = fmaf(wt, mlz[w * 16 + q], ls);
    }
    const int head = kvhd * g + q;
    if (single) {  // the whole cache went through this block: normalise and emit the attention output tensor (bf16)
      emit(head, d, acc * (1.0f / ls));
    } else {
      act_stf<true>(a.part_o + ((int64_t)split * a.nh + head) * 128 + d, acc);
      if (d == 0) {
        act_stf<true>(a.part_ml + ((int64_t)split * a.nh + head) * 2 + 0, M);
        act_stf<true>(a.part_ml + ((int64_t)split * a.nh + head) * 2 + 1, ls);
      }
    }
  }
  if (single) {
    stamp(5);
    return true;
  }
  stamp(3);

  // ---- the LAST split block of this kv head to finish merges all splits of its g heads --------------------------------
  // Partials cross blocks (possibly XCDs) inside one launch: agent-scope stores above, every wave waits for their
  // acknowledgement, then one relaxed atomic per block on the head's counter decides who arrived last.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  sync();
  int* s_last = reinterpret_cast<int*>(qs);  // q fragments are in registers since the page loop; LDS region is free
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(a.head_ctr + 32 * kvhd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = (prev + 1u == a.ctr_target) ? 1 : 0;
  }
  sync();
  stamp(4);
  if (*s_last == 0) return false;
  for (int it = tid; it < g * 128; it += NT) {
    const int q = it >> 7, d = it & 127;
    const int head = kvhd * g + q;
    float M = -INFINITY, ls = 0.f, f = 0.f;
    // 8 splits per round, all 16 loads of a round issued before any is used; running (M, ls, f) rescaled between rounds
    for (int s0 = 0; s0 < nsplit; s0 += 8) {
      float2 ml[8];
      float po[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int sidx = min(s0 + j, nsplit - 1);
        const size_t hb = (size_t)sidx * a.nh + head;
        ml[j] = act_ldf2<true>(a.part_ml + hb * 2);
        po[j] = act_ldf<true>(a.part_o + hb * 128 + d);
        if (s0 + j >= nsplit) ml[j].x = -INFINITY;
      }
      float Mc = M;
#pragma unroll
      for (int j = 0; j < 8; ++j) Mc = fmaxf(Mc, ml[j].x);
      if (Mc == -INFINITY) continue;
      const float resc = (M == -INFINITY) ? 0.f : __expf(M - Mc);
      ls *= resc;
      f *= resc;
      M = Mc;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float wgt = (ml[j].x == -INFINITY) ? 0.f : __expf(ml[j].x - M);
        ls = fmaf(wgt, ml[j].y, ls);
        f = fmaf(wgt, po[j], f);
      }
    }
    emit(head, d, f * (1.0f / ls));  // attention output tensor (rounded to bf16 by the receiver)
  }
  stamp(5);
  return true;
}

// The whole-workgroup form used by attn_decode_fused_kernel and the persistent decode-step kernel: 4 waves, __syncthreads,
// output written to a.o as bf16.
template <bool COH, class AfterPrefetch>
__device__ __forceinline__ bool attn_decode_fused_body(const AttnDecodeFusedArgs& a, char* smem, const int kvhd, const int split,
                                                       const int nsplit, AfterPrefetch&& after_prefetch) {
  return attn_decode_fused_body_t<COH, 4>(
      a, smem, kvhd, split, nsplit, (int)(threadIdx.x >> 6), (int)threadIdx.x, after_prefetch, [] { __syncthreads(); },
      [&](int head, int d, float v) { act_st_bf<COH>((bf16_t*)a.o + head * 128 + d, f2bf(v)); });
}

}  // namespace aha
