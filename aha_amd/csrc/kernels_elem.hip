// Row-wise / index kernels of the Qwen3 decoder stack: embedding gather (D0), RMSNorm rows (D3),
// q/k-norm + rotary + KV append (D4/D5/D6, M1), argmax (D11).  All HBM-bound: 16-byte accesses, one wave per row.
#include "common.h"
#include "kernels.h"

namespace aha {

// ---- D0: embedding gather ------------------------------------------------------------------------------------
// candle_nn::Embedding via Qwen3Model::embedding_token_id (/root/reference/src/models/qwen3/model.rs:191-193)
__global__ __launch_bounds__(256) void embed_gather_kernel(const bf16_t* __restrict__ table, const uint32_t* __restrict__ ids,
                                                           bf16_t* __restrict__ out, int S, int H) {
  const int row = blockIdx.x;
  const uint32_t id = ids[row];
  const u32x4_t* src = reinterpret_cast<const u32x4_t*>(table + (size_t)id * H);
  u32x4_t* dst = reinterpret_cast<u32x4_t*>(out + (size_t)row * H);
  for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = src[i];
}
void launch_embed_gather(const void* table, const uint32_t* ids, void* out, int S, int H, hipStream_t st) {
  if (S <= 0) return;
  hipLaunchKernelGGL(embed_gather_kernel, dim3(S), dim3(256), 0, st, (const bf16_t*)table, ids, (bf16_t*)out, S, H);
}

// ---- D3: RMSNorm over rows ------------------------------------------------------------------------------------
// candle_nn::RmsNorm (/root/reference/src/models/qwen3/model.rs:53-62,79,83,186): y = x / sqrt(mean(x^2)+eps) * w.
// One wave per row; the row is read once (kept in registers for dim <= 8192), f32 sum, single bf16 rounding.
template <int VPL>  // 16-byte vectors per lane (dim = VPL*512 max)
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ y, int64_t rows, int dim, int64_t ldx,
                                                           int64_t ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  bf16_t* yr = y + row * ldy;
  const int nvec = dim / 8;
  u32x4_t v[VPL];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
      v[i] = ld16(xr + vi * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = lo_bf(v[i][j]), b = hi_bf(v[i][j]);
        float t = fmaf(a, a, b * b);   // == a * a + b * b as the compiler contracts it
        // The four terms are added by four scalar v_add_f32 in order.  Left to itself clang packs the running sum into v_pk_add_f32
        // pairs and finishes with `v_pk_add_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[1,0]` (lo = A.lo + B.hi): on gfx950 that form
        // was measured to add B.lo instead in lanes 16-31 / 48-63 whenever the wave shares a SIMD with a wave of another kernel
        // running on a concurrent stream (profiles/r03_simd_coresidency.md; tests/test_isa_cpu.py keeps it out of every kernel).
        asm volatile("" : "+v"(t));
        ss += t;
      }
    }
  }
  ss = wave_sum(ss);
  const float rinv = 1.0f / sqrtf(ss / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
      const u32x4_t wv = ld16(w + vi * 8);
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf(lo_bf(v[i][j]) * rinv * lo_bf(wv[j]), hi_bf(v[i][j]) * rinv * hi_bf(wv[j]));
      *reinterpret_cast<u32x4_t*>(yr + vi * 8) = o;
    }
  }
}
void launch_rmsnorm_rows(const void* x, const void* w, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                         float eps, hipStream_t st) {
  if (rows <= 0) return;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int vpl = (dim / 8 + 63) / 64;
#define RMS_CASE(V)                                                                                              \
  hipLaunchKernelGGL(rmsnorm_rows_kernel<V>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, \
                     rows, dim, ldx, ldy, eps)
  if (vpl <= 1) RMS_CASE(1);
  else if (vpl <= 2) RMS_CASE(2);
  else if (vpl <= 4) RMS_CASE(4);
  else if (vpl <= 8) RMS_CASE(8);
  else if (vpl <= 10) RMS_CASE(10);
  else RMS_CASE(16);
#undef RMS_CASE
}

// ---- D4/D5/D6/M1: per-head q/k RMSNorm + rotary embedding + KV append ----------------------------------------
// QKNormAttention::forward (/root/reference/src/models/common/modules.rs:530-566): q_norm/k_norm over head_dim,
// apply_rotary_pos_emb (/root/reference/src/position_embed/rope.rs:96-132) with the half-split rotate_half
// (rope.rs:15-22), then the roped K and the raw V are appended to the cache (modules.rs:558-566).
// cos/sin follow RoPE::forward (rope.rs:593-612) / Qwen3VLTextRotaryEmbedding::forward (rope.rs:541-580):
// angle = f32(pos[axis(i)]) * inv_freq[i], cos/sin in f32, cast to bf16 before use; each of q*cos, rot(q)*sin and
// their sum is a materialised bf16 tensor in the reference, so each is rounded here too.
// One wave per (token, head): lane l owns elements l and l+64 of the 128-wide head == one rotate_half pair.
constexpr int ROPE_SLOTS_PER_WAVE = 8;  // cos/sin depend on (token, lane) only: one wave computes them once for 8 head slots
__global__ __launch_bounds__(256) void qknorm_rope_kernel(RopeArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nslot = a.nh + 2 * a.kvh;
  const int ngrp = (nslot + ROPE_SLOTS_PER_WAVE - 1) / ROPE_SLOTS_PER_WAVE;
  if (wid >= (int64_t)a.S * ngrp) return;
  const int s = (int)(wid / ngrp);
  const int slot0 = (int)(wid % ngrp) * ROPE_SLOTS_PER_WAVE;
  const int ax = a.axis_map[lane];
  const float ang = (float)a.pos[(int64_t)ax * a.pos_ld + s] * a.inv_freq[lane];
  const float c = rbf(cosf(ang)), sn = rbf(sinf(ang));
  const float qw0 = bf2f(((const bf16_t*)a.q_norm_w)[lane]), qw1 = bf2f(((const bf16_t*)a.q_norm_w)[lane + 64]);
  const float kw0 = bf2f(((const bf16_t*)a.k_norm_w)[lane]), kw1 = bf2f(((const bf16_t*)a.k_norm_w)[lane + 64]);
  const bf16_t* row = (const bf16_t*)a.qkv + (int64_t)s * a.ld;
  const int nloc = min(ROPE_SLOTS_PER_WAVE, nslot - slot0);
  float xa[ROPE_SLOTS_PER_WAVE], xb[ROPE_SLOTS_PER_WAVE];
#pragma unroll
  for (int j = 0; j < ROPE_SLOTS_PER_WAVE; ++j) {  // all loads of the group in flight together
    if (j < nloc) {
      xa[j] = bf2f(row[(int64_t)(slot0 + j) * 128 + lane]);
      xb[j] = bf2f(row[(int64_t)(slot0 + j) * 128 + lane + 64]);
    }
  }
#pragma unroll
  for (int j = 0; j < ROPE_SLOTS_PER_WAVE; ++j) {
    if (j >= nloc) break;
    const int slot = slot0 + j;
    float x0 = xa[j], x1 = xb[j];
    const bool is_q = slot < a.nh;
    const bool is_k = !is_q && slot < a.nh + a.kvh;
    if (is_q || is_k) {
      const float ss = wave_sum(fmaf(x0, x0, x1 * x1));
      const float rinv = 1.0f / sqrtf(ss / 128.0f + a.eps);
      x0 = rbf(x0 * rinv * (is_q ? qw0 : kw0));
      x1 = rbf(x1 * rinv * (is_q ? qw1 : kw1));
      const float y0 = rbf(rbf(x0 * c) + rbf(-x1 * sn));
      const float y1 = rbf(rbf(x1 * c) + rbf(x0 * sn));
      x0 = y0;
      x1 = y1;
    }
    const bf16_t b0 = f2bf(x0), b1 = f2bf(x1);
    if (is_q) {
      bf16_t* dst = (bf16_t*)a.q_out + (int64_t)s * a.nh * 128 + (int64_t)slot * 128;
      dst[lane] = b0;
      dst[lane + 64] = b1;
      continue;
    }
    const int h = is_k ? slot - a.nh : slot - a.nh - a.kvh;
    if (a.kv.page_ptrs == nullptr) {
      bf16_t* dst = (bf16_t*)(is_k ? a.k_out : a.v_out) + (int64_t)s * a.kvh * 128 + (int64_t)h * 128;
      dst[lane] = b0;
      dst[lane + 64] = b1;
      continue;
    }
    // the cache offset the host named for this call (chunked / context-parallel prefill: the segment's first position) wins over the
    // step state's, exactly as in the rows kernel (round-4 advisor: with AHA_ROPE_ROWS=0 a second segment landed on pages [0, len))
    const int tok = (a.kv_start_host >= 0 ? a.kv_start_host : *a.kv_start) + s;
    const int page = tok / KV_PAGE_TOKENS, t = tok % KV_PAGE_TOKENS;
    char* base = reinterpret_cast<char*>(a.kv.page_ptrs[page] + a.kv.layer_off);
    if (is_k) {
      bf16_t* dst = reinterpret_cast<bf16_t*>(base) + (int64_t)h * KV_PAGE_TOKENS * 128;
      dst[kpage_elem(t, lane, 4)] = b0;
      dst[kpage_elem(t, lane + 64, 4)] = b1;
    } else {
      bf16_t* dst = reinterpret_cast<bf16_t*>(base) + (int64_t)a.kvh * KV_PAGE_TOKENS * 128 + (int64_t)h * 128 * KV_PAGE_TOKENS;
      dst[vpage_elem(t, lane)] = b0;
      dst[vpage_elem(t, lane + 64)] = b1;
    }
  }
}
// ---- the same op for a prefill (many tokens, paged destination, cache offset known on the host), 16 bytes per lane -------------
// qknorm_rope_kernel moves 2 bytes per lane and instruction (lane l owns elements l and l+64 of a head) and recomputes cosf / sinf
// of every angle in every layer: 23 us per call on 38 MB.  Here a 16-lane group owns one head of one token (lane sub holds dims
// 8 sub .. 8 sub + 7: one 16-byte load and one 16-byte store, the store being a whole piece of the fragment-major K page), a wave
// covers 4 tokens x a run of head slots, the rotate_half partner (dims +-64) is the lane 8 further on in the 16-lane row (DPP
// row_ror:8), and cos / sin come from a table computed once per prefill (launch_rope_table): 16.5 us.
// Bit-identical to qknorm_rope_kernel by construction: the sum of squares adds the same 64 partial terms
// fma(x[l], x[l], x[l+64]^2) along the same butterfly (partners l^32, l^16, l^8, l^4, l^2, l^1: here rows of 8 lanes x 8
// registers, i.e. three DPP steps then three in-register steps), and every later expression is the old one
// (tests/test_model_gpu.py::test_row_vectorised_rope_kernel_is_bit_identical).
// V rows take a second block role: 128 lanes per (page, kv head) load 8 tokens x 8 dims each, transpose 8 x 8 in registers
// and store eight 16-byte pieces of the fragment-major V page (a piece = one dim x 8 token slots); pieces that
// are only partly covered by this call's tokens fall back to 2-byte stores.
template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
// ROPE_ROWS_CHUNK = head slots per wave in the q / k role: 8 (40 q + k heads of the 8B model = 5 waves per 4 tokens, 24 of the 0.6B = 3)
// against 10: 15.2 vs 17.0 us at cfg 3 (more waves in flight; 5: 15.6 us).  Leading the grid with the V blocks: no difference.
template <int ROPE_ROWS_CHUNK>
__global__ __launch_bounds__(256) void qknorm_rope_rows_kernel(RopeArgs a, int n_qk_blocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nqk = a.nh + a.kvh;
  const int bid = (int)blockIdx.x;
  if (bid < n_qk_blocks) {
    // ---- role A: q / k heads ----
    const int first = a.skip_q ? a.nh : 0;   // skip_q: the attention kernel norms + rotates the q heads in its Q load
    const int nchunk = (nqk - first + ROPE_ROWS_CHUNK - 1) / ROPE_ROWS_CHUNK;
    const int64_t wid = (int64_t)bid * 4 + wave;
    const int tg = (int)(wid / nchunk), chunk = (int)(wid % nchunk);
    const int tq = lane >> 4, sub = lane & 15, sp = sub & 7;
    const int tok = tg * 4 + tq;
    if (tg * 4 >= a.S) return;
    const bool tok_ok = tok < a.S;
    const int tokc = tok_ok ? tok : a.S - 1;
    float c[8], sn[8], qw[8], kw[8];
    if (a.rope_tab) {
      const bf16_t* tp = (const bf16_t*)a.rope_tab + (int64_t)tokc * 128 + sp * 8;
      const u32x4_t c4 = ld16(tp), s4 = ld16(tp + 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[2 * j] = lo_bf(c4[j]); c[2 * j + 1] = hi_bf(c4[j]);
        sn[2 * j] = lo_bf(s4[j]); sn[2 * j + 1] = hi_bf(s4[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = sp * 8 + j;
        const float ang = (float)a.pos[(int64_t)a.axis_map[i] * a.pos_ld + tokc] * a.inv_freq[i];
        c[j] = rbf(cosf(ang));
        sn[j] = rbf(sinf(ang));
      }
    }
    {
      const u32x4_t q4 = ld16((const bf16_t*)a.q_norm_w + sub * 8), k4 = ld16((const bf16_t*)a.k_norm_w + sub * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qw[2 * j] = lo_bf(q4[j]); qw[2 * j + 1] = hi_bf(q4[j]);
        kw[2 * j] = lo_bf(k4[j]); kw[2 * j + 1] = hi_bf(k4[j]);
      }
    }
    const bf16_t* row = (const bf16_t*)a.qkv + (int64_t)tokc * a.ld + sub * 8;
    const int slot0 = first + chunk * ROPE_ROWS_CHUNK, slot1 = min(nqk, slot0 + ROPE_ROWS_CHUNK);
    const int ctok = a.kv_start_host + tokc;
    const int page = ctok / KV_PAGE_TOKENS, t = ctok % KV_PAGE_TOKENS;
    char* kbase = reinterpret_cast<char*>(a.kv.page_ptrs[page] + a.kv.layer_off);
    const bool hi_half = sub >= 8;
    u32x4_t raw[ROPE_ROWS_CHUNK];
#pragma unroll
    for (int i = 0; i < ROPE_ROWS_CHUNK; ++i)
      if (slot0 + i < slot1) raw[i] = ld16(row + (int64_t)(slot0 + i) * 128);
#pragma unroll
    for (int i = 0; i < ROPE_ROWS_CHUNK; ++i) {
      const int slot = slot0 + i;
      if (slot >= slot1) break;
      const bool is_q = slot < a.nh;
      float x[8], px[8], s[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[2 * j] = lo_bf(raw[i][j]); x[2 * j + 1] = hi_bf(raw[i][j]); }
#pragma unroll
      for (int j = 0; j < 8; ++j) px[j] = dpp_rot<0x128>(x[j]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float lo = hi_half ? px[j] : x[j], hi = hi_half ? x[j] : px[j];
        s[j] = fmaf(lo, lo, hi * hi);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += dpp_rot<0x124>(s[j]);   // partner l ^ 32
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += dpp_rot<0x122>(s[j]);   // l ^ 16
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += dpp_rot<0x121>(s[j]);   // l ^ 8
      const float u0 = s[0] + s[4], u1 = s[1] + s[5], u2 = s[2] + s[6], u3 = s[3] + s[7];   // l ^ 4
      const float v0 = u0 + u2, v1 = u1 + u3;                                                // l ^ 2
      const float ss = v0 + v1;                                                              // l ^ 1
      const float rinv = 1.0f / sqrtf(ss / 128.0f + a.eps);
      float xn[8], pn[8], y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[j] = rbf(x[j] * rinv * (is_q ? qw[j] : kw[j]));
#pragma unroll
      for (int j = 0; j < 8; ++j) pn[j] = dpp_rot<0x128>(xn[j]);
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = rbf(rbf(xn[j] * c[j]) + rbf((hi_half ? pn[j] : -pn[j]) * sn[j]));
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack_bf(y[2 * j], y[2 * j + 1]);
      if (!tok_ok) continue;
      if (is_q) {
        *reinterpret_cast<u32x4_t*>((bf16_t*)a.q_out + ((int64_t)tok * a.nh + slot) * 128 + sub * 8) = o;
      } else {
        bf16_t* dst = reinterpret_cast<bf16_t*>(kbase) + (int64_t)(slot - a.nh) * KV_PAGE_TOKENS * 128;
        *reinterpret_cast<u32x4_t*>(dst + kpage_elem(t, sub * 8, 4)) = o;
      }
    }
    return;
  }
  // ---- role B: V rows, 128 lanes per (page of this call, kv head) ----
  const int64_t u = ((int64_t)(bid - n_qk_blocks) * 256 + threadIdx.x);
  const int unit = (int)(u >> 7), lu = (int)(u & 127);
  const int p0 = a.kv_start_host / KV_PAGE_TOKENS, p1 = (a.kv_start_host + a.S - 1) / KV_PAGE_TOKENS;
  const int page = p0 + unit / a.kvh, h = unit % a.kvh;
  if (page > p1) return;
  const int kk = lu >> 6, G = (lu >> 4) & 3, dchunk = lu & 15;
  uint32_t in[8][4];
  bool ok[8];
  bool all = true;
#pragma unroll
  for (int e = 0; e < 8; ++e) {   // slot order inside the piece: e = sub1 * 4 + j  <->  token kk*32 + sub1*16 + G*4 + j
    const int tokp = kk * 32 + (e >> 2) * 16 + G * 4 + (e & 3);
    const int srow = page * KV_PAGE_TOKENS + tokp - a.kv_start_host;
    ok[e] = srow >= 0 && srow < a.S;
    all &= ok[e];
    const u32x4_t v = ld16((const bf16_t*)a.qkv + (int64_t)min(max(srow, 0), a.S - 1) * a.ld + (int64_t)(a.nh + a.kvh + h) * 128 + dchunk * 8);
    in[e][0] = v[0]; in[e][1] = v[1]; in[e][2] = v[2]; in[e][3] = v[3];
  }
  bf16_t* vb = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(a.kv.page_ptrs[page] + a.kv.layer_off)) +
               (int64_t)a.kvh * KV_PAGE_TOKENS * 128 + (int64_t)h * 128 * KV_PAGE_TOKENS;
#pragma unroll
  for (int dd = 0; dd < 8; ++dd) {   // dim d = dchunk*8 + dd: word dd>>1, half dd&1 of every token's 16 bytes
    const int d = dchunk * 8 + dd;
    u32x4_t o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t a0 = in[2 * w][dd >> 1], a1 = in[2 * w + 1][dd >> 1];
      o[w] = (dd & 1) ? ((a0 >> 16) | (a1 & 0xffff0000u)) : ((a0 & 0xffffu) | (a1 << 16));
    }
    bf16_t* dst = vb + ((((d >> 4) * 2 + kk) * 64 + G * 16 + (d & 15)) << 3);
    if (all) {
      *reinterpret_cast<u32x4_t*>(dst) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (ok[e]) dst[e] = (bf16_t)((e & 1) ? (o[e >> 1] >> 16) : (o[e >> 1] & 0xffffu));
    }
  }
}

__global__ __launch_bounds__(256) void rope_table_kernel(const int32_t* __restrict__ pos, int64_t pos_ld, const float* __restrict__ inv_freq,
                                                         const int32_t* __restrict__ axis_map, int S, bf16_t* __restrict__ tab) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)S * 64) return;
  const int tok = (int)(g >> 6), i = (int)(g & 63);
  const float ang = (float)pos[(int64_t)axis_map[i] * pos_ld + tok] * inv_freq[i];
  tab[(int64_t)tok * 128 + i] = f2bf(cosf(ang));
  tab[(int64_t)tok * 128 + 64 + i] = f2bf(sinf(ang));
}
void launch_rope_table(const int32_t* pos, int64_t pos_ld, const float* inv_freq, const int32_t* axis_map, int S, void* tab, hipStream_t st) {
  if (S <= 0) return;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)(((int64_t)S * 64 + 255) / 256)), dim3(256), 0, st, pos, pos_ld, inv_freq, axis_map, S,
                     (bf16_t*)tab);
}

void launch_qknorm_rope(const RopeArgs& a, hipStream_t st) {
  static const bool rows_on = [] { const char* e = getenv("AHA_ROPE_ROWS"); return e ? atoi(e) != 0 : true; }();
  if (rows_on && a.kv_start_host >= 0 && a.kv.page_ptrs != nullptr && a.d == 128 && a.S >= 16) {
    static const int chunk_env = [] { const char* e = getenv("AHA_ROPE_CHUNK"); return e ? atoi(e) : 8; }();
    // K heads only (skip_q): two heads per wave -- eight would leave S / 4 waves walking all kv heads one after the other (cfg 3: 386 waves on
    // 1024 SIMDs, ~3.7 us of dependent vector work each)
    static const int chunk_k = [] { const char* e = getenv("AHA_ROPE_CHUNK_K"); return e ? atoi(e) : 2; }();
    const int chunk = a.skip_q ? (chunk_k == 8 ? 8 : 2) : (chunk_env == 10 ? 10 : 8);
    const int nqk = (a.skip_q ? 0 : a.nh) + a.kvh, nchunk = (nqk + chunk - 1) / chunk;
    const int64_t qk_waves = (int64_t)((a.S + 3) / 4) * nchunk;
    const int n_qk_blocks = (int)((qk_waves + 3) / 4);
    const int npages = (a.kv_start_host + a.S - 1) / KV_PAGE_TOKENS - a.kv_start_host / KV_PAGE_TOKENS + 1;
    const int n_v_blocks = (npages * a.kvh * 128 + 255) / 256;
    const dim3 grid((unsigned)(n_qk_blocks + n_v_blocks));
    if (chunk == 2) hipLaunchKernelGGL(qknorm_rope_rows_kernel<2>, grid, dim3(256), 0, st, a, n_qk_blocks);
    else if (chunk == 8) hipLaunchKernelGGL(qknorm_rope_rows_kernel<8>, grid, dim3(256), 0, st, a, n_qk_blocks);
    else hipLaunchKernelGGL(qknorm_rope_rows_kernel<10>, grid, dim3(256), 0, st, a, n_qk_blocks);
    return;
  }
  const int64_t waves = (int64_t)a.S * ((a.nh + 2 * a.kvh + ROPE_SLOTS_PER_WAVE - 1) / ROPE_SLOTS_PER_WAVE);
  if (waves <= 0) return;
  hipLaunchKernelGGL(qknorm_rope_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
}

// contiguous token-major K,V (L, kvh*128) -> pages.  One wave per (token, kv head, K|V).
__global__ __launch_bounds__(256) void kv_pack_pages_kernel(const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                            KvLayer kv, int L) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)L * kv.kvh * 2) return;
  const int which = (int)(wid % 2);
  const int h = (int)((wid / 2) % kv.kvh);
  const int tok = (int)(wid / 2 / kv.kvh);
  const bf16_t* src = (which ? v : k) + ((int64_t)tok * kv.kvh + h) * 128;
  const int page = tok / KV_PAGE_TOKENS, t = tok % KV_PAGE_TOKENS;
  bf16_t* base = reinterpret_cast<bf16_t*>(kv.page_ptrs[page] + kv.layer_off);
  if (!which) {
    bf16_t* dst = base + (int64_t)h * KV_PAGE_TOKENS * 128;
    dst[kpage_elem(t, lane, 4)] = src[lane];
    dst[kpage_elem(t, lane + 64, 4)] = src[lane + 64];
  } else {
    bf16_t* dst = base + (int64_t)kv.kvh * KV_PAGE_TOKENS * 128 + (int64_t)h * 128 * KV_PAGE_TOKENS;
    dst[vpage_elem(t, lane)] = src[lane];
    dst[vpage_elem(t, lane + 64)] = src[lane + 64];
  }
}
void launch_kv_pack_pages(const void* k, const void* v, KvLayer kv, int L, hipStream_t st) {
  const int64_t waves = (int64_t)L * kv.kvh * 2;
  if (waves <= 0) return;
  hipLaunchKernelGGL(kv_pack_pages_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const bf16_t*)k,
                     (const bf16_t*)v, kv, L);
}

// ---- D11: greedy argmax (first maximal index, as candle's argmax / Sampling::ArgMax) ---------------------------
__device__ __forceinline__ void argmax_combine(float& bv, uint32_t& bi, float v, uint32_t i) {
  if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}
__device__ __forceinline__ void wave_argmax(float& bv, uint32_t& bi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const uint32_t oi = __shfl_xor(bi, o, 64);
    argmax_combine(bv, bi, ov, oi);
  }
}
__global__ __launch_bounds__(256) void argmax_partials_kernel(const float* __restrict__ pv, const uint32_t* __restrict__ pi,
                                                              int n, uint32_t* __restrict__ out) {
  __shared__ float sv[4];
  __shared__ uint32_t si[4];
  float bv = -INFINITY;
  uint32_t bi = 0xffffffffu;
  for (int i = threadIdx.x; i < n; i += 256) argmax_combine(bv, bi, pv[i], pi[i]);
  wave_argmax(bv, bi);
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) argmax_combine(bv, bi, sv[w], si[w]);
    *out = bi;
  }
}
// vocab-parallel lm_head: pairs[2r] = this rank's max logit, pairs[2r+1] = its GLOBAL index (exact in f32: < 2^24), every
// other slot 0, so that a sum all-reduce over ranks leaves all T pairs everywhere
__global__ __launch_bounds__(256) void argmax_pair_kernel(const float* __restrict__ pv, const uint32_t* __restrict__ pi, int n,
                                                          int row0, float* __restrict__ pairs, int rank, int T) {
  __shared__ float sv[4];
  __shared__ uint32_t si[4];
  float bv = -INFINITY;
  uint32_t bi = 0xffffffffu;
  for (int i = threadIdx.x; i < n; i += 256) argmax_combine(bv, bi, pv[i], pi[i]);
  wave_argmax(bv, bi);
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if ((int)threadIdx.x < 2 * T) pairs[threadIdx.x] = 0.f;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) argmax_combine(bv, bi, sv[w], si[w]);
    pairs[2 * rank] = bv;
    pairs[2 * rank + 1] = (float)(bi + (uint32_t)row0);
  }
}
__global__ void argmax_pick_kernel(const float* __restrict__ pairs, int T, uint32_t* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float bv = -INFINITY;
  uint32_t bi = 0xffffffffu;
  for (int r = 0; r < T; ++r) argmax_combine(bv, bi, pairs[2 * r], (uint32_t)pairs[2 * r + 1]);
  *out = bi;
}
void launch_argmax_pair(const float* blk_max, const uint32_t* blk_idx, int n, int row0, float* pairs, int rank, int T, hipStream_t st) {
  hipLaunchKernelGGL(argmax_pair_kernel, dim3(1), dim3(256), 0, st, blk_max, blk_idx, n, row0, pairs, rank, T);
}
void launch_argmax_pick(const float* pairs, int T, uint32_t* out, hipStream_t st) {
  hipLaunchKernelGGL(argmax_pick_kernel, dim3(1), dim3(64), 0, st, pairs, T, out);
}
void launch_argmax_partials(const float* blk_max, const uint32_t* blk_idx, int n, uint32_t* out, hipStream_t st) {
  hipLaunchKernelGGL(argmax_partials_kernel, dim3(1), dim3(256), 0, st, blk_max, blk_idx, n, out);
}
__global__ __launch_bounds__(256) void argmax_f32_stage1(const float* __restrict__ x, int64_t n, float* __restrict__ pv,
                                                         uint32_t* __restrict__ pi) {
  __shared__ float sv[4];
  __shared__ uint32_t si[4];
  float bv = -INFINITY;
  uint32_t bi = 0xffffffffu;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    // NaN never wins (v > bv is false), matching a partial_cmp-style scan that keeps the first non-NaN max
    argmax_combine(bv, bi, v, (uint32_t)i);
  }
  wave_argmax(bv, bi);
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) argmax_combine(bv, bi, sv[w], si[w]);
    pv[blockIdx.x] = bv;
    pi[blockIdx.x] = bi;
  }
}
void launch_argmax_f32(const float* x, int64_t n, float* ws_max, uint32_t* ws_idx, uint32_t* out, hipStream_t st) {
  const int blocks = (int)((n + 4095) / 4096 < 256 ? (n + 4095) / 4096 : 256);
  hipLaunchKernelGGL(argmax_f32_stage1, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, x, n, ws_max, ws_idx);
  launch_argmax_partials(ws_max, ws_idx, blocks > 0 ? blocks : 1, out, st);
}

}  // namespace aha

// ---- tensor-parallel seam: x = bf16(x + bf16(sum)) after the all-reduce of f32 partial projections ---------------------
namespace aha {
__global__ __launch_bounds__(256) void residual_add_f32_kernel(bf16_t* __restrict__ x, const float* __restrict__ s, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 256 * 4) {
    const float4 v = *reinterpret_cast<const float4*>(s + i);
    uint2 xv = *reinterpret_cast<const uint2*>(x + i);
    xv.x = pack_bf(lo_bf(xv.x) + rbf(v.x), hi_bf(xv.x) + rbf(v.y));
    xv.y = pack_bf(lo_bf(xv.y) + rbf(v.z), hi_bf(xv.y) + rbf(v.w));
    *reinterpret_cast<uint2*>(x + i) = xv;
  }
}
// sequence-parallel prefill: broadcast of one bf16 row through the f32 all-reduce seam (zeros from the ranks that do not own it)
__global__ __launch_bounds__(256) void row_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = x ? bf2f(x[i]) : 0.f;
}
__global__ __launch_bounds__(256) void f32_to_row_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = f2bf(x[i]);
}
void launch_row_to_f32(const void* x_or_null, float* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(row_to_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const bf16_t*)x_or_null, out, n);
}
void launch_f32_to_row(const float* x, void* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(f32_to_row_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, (bf16_t*)out, n);
}
// the same over a column block: x[r, c] (row pitch ldx) += bf16(s[r, c]) (row pitch lds), rows x cols, cols % 4 == 0
__global__ __launch_bounds__(256) void residual_add_f32_cols_kernel(bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ s, int64_t lds,
                                                                    int64_t rows, int cols) {
  const int q = cols >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * q; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    const float4 v = *reinterpret_cast<const float4*>(s + r * lds + c);
    uint2 xv = *reinterpret_cast<const uint2*>(x + r * ldx + c);
    xv.x = pack_bf(lo_bf(xv.x) + rbf(v.x), hi_bf(xv.x) + rbf(v.y));
    xv.y = pack_bf(lo_bf(xv.y) + rbf(v.z), hi_bf(xv.y) + rbf(v.w));
    *reinterpret_cast<uint2*>(x + r * ldx + c) = xv;
  }
}
void launch_residual_add_f32_cols(void* x, int64_t ldx, const float* sum, int64_t lds, int64_t rows, int cols, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return;
  const int64_t blocks = (rows * (cols >> 2) + 255) / 256;
  hipLaunchKernelGGL(residual_add_f32_cols_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, (bf16_t*)x, ldx, sum, lds,
                     rows, cols);
}
void launch_residual_add_f32(void* x, const float* sum, int64_t n, hipStream_t st) {
  if (n <= 0) return;
  const int64_t blocks = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(residual_add_f32_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, (bf16_t*)x, sum, n);
}
// Test tool (aha_hip_debug_poison_lds): fills the LDS of every CU with seeded garbage.  A kernel that consumes LDS it did not stage
// in the same launch computes from whatever the previous kernel on that CU left there -- identical, hence invisible, when the same
// launch is repeated; with a different poison in front of each launch it shows up as
// run-to-run differences.  1024 blocks x 64 KiB: several blocks per CU, every byte of the 160 KiB written by one of them.
__global__ __launch_bounds__(256) void poison_lds_kernel(uint32_t seed, uint32_t* sink) {
  extern __shared__ uint32_t lds_words[];
  const int n = 64 * 1024 / 4;
  uint32_t x = seed * 2654435761u + blockIdx.x * 40503u + threadIdx.x;
  for (int i = threadIdx.x; i < n; i += 256) {
    x = x * 1664525u + 1013904223u;
    lds_words[i] = (x & 0x7f7f7f7fu) | 0x3f003f00u;   // finite bf16 / f32 patterns of order 1
  }
  __syncthreads();
  if (sink != nullptr && lds_words[(seed + threadIdx.x) % n] == 0xdeadbeefu) *sink = x;   // keep the stores alive
}
void launch_poison_lds(uint32_t seed, hipStream_t st) {
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 64 * 1024, st, seed, (uint32_t*)nullptr);
}
}  // namespace aha
