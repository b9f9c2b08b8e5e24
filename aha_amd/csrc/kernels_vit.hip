// Row-wise kernels of the Qwen3-VL vision tower (SURVEY.md section 8a V2, V3, V4 staging, V5/V6 LayerNorm, M3).
#include "common.h"
#include "kernels.h"

namespace aha {

// ---- LayerNorm rows (candle_nn::LayerNorm via get_layer_norm, /root/reference/src/models/common/modules.rs:867-875;
// used at qwen3vl/model.rs:355-366 and 167-178).  One wave per row, row held in registers, f32 statistics.
template <int VPL>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                             int64_t rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * dim;
  bf16_t* yr = y + row * dim;
  const int nvec = dim / 8;
  float f[VPL][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
      const u32x4_t v = ld16(xr + vi * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) { f[i][2 * j] = lo_bf(v[j]); f[i][2 * j + 1] = hi_bf(v[j]); }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; q += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = i * 64 + lane;
    if (vi < nvec) {
      const u32x4_t wv = ld16(w + vi * 8), bv = ld16(b + vi * 8);
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf((f[i][2 * j] - mean) * rstd * lo_bf(wv[j]) + lo_bf(bv[j]),
                       (f[i][2 * j + 1] - mean) * rstd * hi_bf(wv[j]) + hi_bf(bv[j]));
      *reinterpret_cast<u32x4_t*>(yr + vi * 8) = o;
    }
  }
}
void launch_layernorm_rows(const void* x, const void* w, const void* b, void* y, int64_t rows, int dim, float eps,
                           hipStream_t st) {
  if (rows <= 0) return;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int vpl = (dim / 8 + 63) / 64;
#define LN_CASE(V)                                                                                                   \
  hipLaunchKernelGGL(layernorm_rows_kernel<V>, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, \
                     (bf16_t*)y, rows, dim, eps)
  if (vpl <= 1) LN_CASE(1);
  else if (vpl <= 3) LN_CASE(3);
  else if (vpl <= 9) LN_CASE(9);
  else LN_CASE(16);
#undef LN_CASE
}

// ---- V2: learned 2-D position embedding, bilinear (fast_pos_embed_interpolate, qwen3vl/model.rs:512-639) ----------
// pos = ((bf16(E[i0]*w0) + bf16(E[i1]*w1)) + bf16(E[i2]*w2)) + bf16(E[i3]*w3), each add rounded; x = bf16(x + pos).
// The corner indices / weights are built on the host exactly as the reference does (f32 linspace, u32 truncation).
__global__ __launch_bounds__(256) void pos_embed_add_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ table,
                                                            const int32_t* __restrict__ idx, const float* __restrict__ wt,
                                                            int64_t N, int D) {
  const int64_t n = blockIdx.x;
  const bf16_t* e[4];
  float w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e[i] = table + (int64_t)idx[i * N + n] * D;
    w[i] = rbf(wt[i * N + n]);  // weight_tensor.to_dtype(model dtype)
  }
  bf16_t* xr = x + n * D;
  for (int v = threadIdx.x; v < D / 8; v += 256) {
    u32x4_t ev[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ev[i] = ld16(e[i] + v * 8);
    const u32x4_t xv = ld16(xr + v * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = rbf(lo_bf(ev[0][j]) * w[0]), hi = rbf(hi_bf(ev[0][j]) * w[0]);
#pragma unroll
      for (int i = 1; i < 4; ++i) {
        lo = rbf(lo + rbf(lo_bf(ev[i][j]) * w[i]));
        hi = rbf(hi + rbf(hi_bf(ev[i][j]) * w[i]));
      }
      o[j] = pack_bf(lo_bf(xv[j]) + lo, hi_bf(xv[j]) + hi);
    }
    *reinterpret_cast<u32x4_t*>(xr + v * 8) = o;
  }
}
void launch_pos_embed_add(void* x, const void* table, const int32_t* idx, const float* wt, int64_t N, int D,
                          hipStream_t st) {
  if (N <= 0) return;
  hipLaunchKernelGGL(pos_embed_add_kernel, dim3((unsigned)N), dim3(256), 0, st, (bf16_t*)x, (const bf16_t*)table, idx, wt, N, D);
}

// ---- V3/V4 staging: 2-D rotary on q,k (apply_rotary_pos_emb_vision, rope.rs:75-94; rot_pos_emb qwen3vl/model.rs:641-690)
// and repack into the attention kernel's operand layouts.  One wave per (token, head).  head_dim hd = 72:
// rotate_half pairs element e with e + hd/2; angle(e) = (e < hd/4 ? row : col) * inv_freq[e % (hd/4)] for e < hd/2,
// and the same again for e >= hd/2 (emb = cat(rotary, rotary), model.rs:704).
// 8 bytes per lane: a (token, q | k) unit is 9 lanes that walk over the heads, lane g holding elements 4g..4g+3 AND their rotate_half partners
// 36 + 4g.. (two 8-byte loads, no cross-lane traffic: 36 elements are 4.5 sixteen-byte pieces, so the pair is not a lane shift);
// every store is 8 bytes (half a piece of the fragment-major K page, or of the padded q row).  The first 6 lanes of a unit also
// zero the pad elements 72..95.  cos / sin come from the per-image table (launch_vit_rope_table).
// V rows take the second block role: 80 lanes per (page, head) -- (token-slot half kk, lane group G, 8-dim chunk) -- load 8 tokens x
// 8 dims, transpose 8 x 8 in registers and store eight 16-byte pieces of the fragment-major V block; pad dim 72 is 1.0 on real tokens
// (the row-sum column, see below), pad dims 73..79 and the slots of a tail page past its last token are zeros (they must be finite).
// (The first version moved 2 bytes per lane and instruction and ran at 1.5 TB/s: 29 us per call at N = 4096, 2.7 ms at 8 x 2048^2.)
__global__ __launch_bounds__(256) void vit_rope_pack_kernel(VitRopeArgs a, int n_qk_blocks) {
  const int lane = threadIdx.x & 63;
  const int hd = a.hd, half = hd / 2;   // 72, 36
  const int D = a.nh * hd;
  if ((int)blockIdx.x < n_qk_blocks) {
    // a 9-lane unit = (token, q | k); it walks over the heads, 4 at a time in flight
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t unit = wid * 7 + lane / 9;
    const int g = lane % 9;
    if (lane >= 63 || unit >= (int64_t)a.N * 2) return;
    const int which = (int)(unit & 1), n = (int)(unit >> 1);
    const int e0 = 4 * g;
    const float4 cs01 = *reinterpret_cast<const float4*>(a.cs_tab + ((int64_t)n * half + e0) * 2);
    const float4 cs23 = *reinterpret_cast<const float4*>(a.cs_tab + ((int64_t)n * half + e0 + 2) * 2);
    const float c[4] = {cs01.x, cs01.z, cs23.x, cs23.z}, s[4] = {cs01.y, cs01.w, cs23.y, cs23.w};
    const bf16_t* src = (const bf16_t*)a.qkv + (int64_t)n * 3 * D + (int64_t)which * D + e0;
    bf16_t* dq = (bf16_t*)a.q_out + (int64_t)n * a.nh * VIT_DQK;
    const int slot = a.slot_of[n];
    bf16_t* dk = reinterpret_cast<bf16_t*>(a.kv.page_ptrs[a.page_of[n]] + a.kv.layer_off);
    const int k0 = kpage_elem(slot, e0, VIT_DQK / 32), k1 = kpage_elem(slot, e0 + half, VIT_DQK / 32), kz = kpage_elem(slot, hd + 4 * (g % 6), VIT_DQK / 32);
    const uint2 zero = {0u, 0u};
    for (int h0 = 0; h0 < a.nh; h0 += 4) {
      uint2 x0w[4], x1w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (h0 + i < a.nh) {
          x0w[i] = *reinterpret_cast<const uint2*>(src + (int64_t)(h0 + i) * hd);
          x1w[i] = *reinterpret_cast<const uint2*>(src + (int64_t)(h0 + i) * hd + half);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int h = h0 + i;
        if (h >= a.nh) break;
        const float x0[4] = {lo_bf(x0w[i].x), hi_bf(x0w[i].x), lo_bf(x0w[i].y), hi_bf(x0w[i].y)};
        const float x1[4] = {lo_bf(x1w[i].x), hi_bf(x1w[i].x), lo_bf(x1w[i].y), hi_bf(x1w[i].y)};
        float y0[4], y1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          y0[j] = rbf(x0[j] * c[j]) + rbf(-x1[j] * s[j]);
          y1[j] = rbf(x1[j] * c[j]) + rbf(x0[j] * s[j]);
        }
        const uint2 o0 = {pack_bf(y0[0], y0[1]), pack_bf(y0[2], y0[3])}, o1 = {pack_bf(y1[0], y1[1]), pack_bf(y1[2], y1[3])};
        if (which == 0) {
          bf16_t* qd = dq + (int64_t)h * VIT_DQK;
          *reinterpret_cast<uint2*>(qd + e0) = o0;
          *reinterpret_cast<uint2*>(qd + e0 + half) = o1;
          if (g < (VIT_DQK - 72) / 4) *reinterpret_cast<uint2*>(qd + hd + 4 * g) = zero;
        } else {
          bf16_t* kd = dk + (int64_t)h * KV_PAGE_TOKENS * VIT_DQK;
          *reinterpret_cast<uint2*>(kd + k0) = o0;
          *reinterpret_cast<uint2*>(kd + k1) = o1;
          if (g < (VIT_DQK - 72) / 4) *reinterpret_cast<uint2*>(kd + kz) = zero;
        }
      }
    }
    return;
  }
  // ---- V: 80 lanes per (page, head) ----
  const int64_t u = (int64_t)(blockIdx.x - n_qk_blocks) * 256 + threadIdx.x;
  const int64_t unit = u / 80;
  const int lu = (int)(u % 80);
  if (unit >= (int64_t)a.n_pages * a.nh) return;
  const int page = (int)(unit / a.nh), h = (int)(unit % a.nh);
  const int kk = lu / 40, G = (lu % 40) / 10, dchunk = lu % 10;
  const int first = a.page_first[page], cnt = a.page_cnt[page];
  uint32_t in[8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {   // slot order inside a piece: e = sub1 * 4 + j  <->  token slot kk*32 + sub1*16 + G*4 + j
    const int tslot = kk * 32 + (e >> 2) * 16 + G * 4 + (e & 3);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (tslot < cnt && dchunk < 9)
      v = ld16((const bf16_t*)a.qkv + (int64_t)(first + tslot) * 3 * D + 2 * (int64_t)D + (int64_t)h * hd + dchunk * 8);
    // pad dim 72 = 1.0 on every real token (round 5): V^T . P^T then delivers the softmax row sum in output row 72 -- rescaled with the
    // accumulators, from the SAME rounded probabilities as the numerator -- and the f32-chain attention kernel drops its own sum
    // (kernels_attn.hip LSUM).  Rows 72..79 of the output are never stored.
    if (tslot < cnt && dchunk == 9) v[0] = 0x3F80u;
    in[e][0] = v[0]; in[e][1] = v[1]; in[e][2] = v[2]; in[e][3] = v[3];
  }
  bf16_t* vd = reinterpret_cast<bf16_t*>(a.kv.page_ptrs[page] + a.kv.layer_off) + (int64_t)a.nh * KV_PAGE_TOKENS * VIT_DQK +
               (int64_t)h * VIT_DV * KV_PAGE_TOKENS;
#pragma unroll
  for (int dd = 0; dd < 8; ++dd) {
    const int d = dchunk * 8 + dd;
    u32x4_t o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t a0 = in[2 * w][dd >> 1], a1 = in[2 * w + 1][dd >> 1];
      o[w] = (dd & 1) ? ((a0 >> 16) | (a1 & 0xffff0000u)) : ((a0 & 0xffffu) | (a1 << 16));
    }
    *reinterpret_cast<u32x4_t*>(vd + ((((d >> 4) * 2 + kk) * 64 + G * 16 + (d & 15)) << 3)) = o;
  }
}
__global__ __launch_bounds__(256) void vit_rope_table_kernel(const int32_t* __restrict__ rowcol, const float* __restrict__ inv_freq, int N,
                                                             int hd, float* __restrict__ tab) {
  const int half = hd / 2, quarter = hd / 4;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)N * half) return;
  const int n = (int)(g / half), lane = (int)(g % half);
  const int pos = (lane < quarter) ? rowcol[2 * n] : rowcol[2 * n + 1];
  const float ang = (float)pos * inv_freq[lane % quarter];
  tab[g * 2] = rbf(cosf(ang));
  tab[g * 2 + 1] = rbf(sinf(ang));
}
void launch_vit_rope_table(const int32_t* rowcol, const float* inv_freq, int N, int hd, float* tab, hipStream_t st) {
  if (N <= 0) return;
  hipLaunchKernelGGL(vit_rope_table_kernel, dim3((unsigned)(((int64_t)N * (hd / 2) + 255) / 256)), dim3(256), 0, st, rowcol, inv_freq, N, hd, tab);
}
void launch_vit_rope_pack(const VitRopeArgs& a, hipStream_t st) {
  if (a.N <= 0) return;
  const int64_t units = (int64_t)a.N * 2, waves = (units + 6) / 7;
  const int n_qk_blocks = (int)((waves + 3) / 4);
  const int n_v_blocks = (int)(((int64_t)a.n_pages * a.nh * 80 + 255) / 256);
  hipLaunchKernelGGL(vit_rope_pack_kernel, dim3((unsigned)(n_qk_blocks + n_v_blocks)), dim3(256), 0, st, a, n_qk_blocks);
}

// ---- M3: dst[rows[i], :] = src[i, :]  (masked_scatter_dim0, tensor_utils.rs:294-321)
//          dst[rows[i], :] = bf16(dst + src[i, :])  (mask_index_add, tensor_utils.rs:466-470)
__global__ __launch_bounds__(256) void scatter_rows_kernel(bf16_t* __restrict__ dst, const bf16_t* __restrict__ src,
                                                           const int32_t* __restrict__ rows, int D, int add) {
  const int64_t i = blockIdx.x;
  bf16_t* d = dst + (int64_t)rows[i] * D;
  const bf16_t* s = src + i * D;
  for (int v = threadIdx.x; v < D / 8; v += 256) {
    u32x4_t sv = ld16(s + v * 8);
    if (add) {
      const u32x4_t dv = ld16(d + v * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) sv[j] = pack_bf(lo_bf(dv[j]) + lo_bf(sv[j]), hi_bf(dv[j]) + hi_bf(sv[j]));
    }
    *reinterpret_cast<u32x4_t*>(d + v * 8) = sv;
  }
}
void launch_scatter_rows(void* dst, const void* src, const int32_t* rows, int64_t n, int D, int add, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, (bf16_t*)dst, (const bf16_t*)src, rows, D, add);
}

}  // namespace aha

// ---- V0: image -> patch rows (Qwen3VLProcessor::process_images, /root/reference/src/models/qwen3vl/processor.rs:174-251;
// img_transform, /root/reference/src/utils/img_utils.rs:272-293).  u8 HWC -> f32 * (1/255) -> (x - mean) / std -> bf16,
// the frame duplicated to T = 2, rows in merge-window order (t, bh, bw, ih, iw), columns ordered (c, t_in_patch, py, px).
// One block per patch row; HBM-bound (3 B in -> 2 x 6 B out per pixel).
namespace aha {
__global__ __launch_bounds__(192) void image_to_patches_kernel(const uint8_t* __restrict__ img, bf16_t* __restrict__ out,
                                                               int H, int W, int patch, int merge, float m0, float m1,
                                                               float m2, float s0, float s1, float s2) {
  const int gw = W / patch, bwn = gw / merge;
  const int n = blockIdx.x;
  const int iw = n % merge, ih = (n / merge) % merge, bw = (n / (merge * merge)) % bwn, bh = n / (merge * merge * bwn);
  const int y0 = (bh * merge + ih) * patch, x0 = (bw * merge + iw) * patch;
  const int pp = patch * patch;
  const float inv255 = 1.0f / 255.0f;
  for (int e = threadIdx.x; e < 3 * pp; e += blockDim.x) {
    const int c = e / pp, py = (e % pp) / patch, px = e % patch;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    const float v = (float)img[((size_t)(y0 + py) * W + (x0 + px)) * 3 + c];
    const bf16_t b = f2bf((v * inv255 + 0.0f - mean) / sd);
    bf16_t* o = out + (size_t)n * (6 * pp) + (size_t)c * 2 * pp + py * patch + px;
    o[0] = b;    // temporal slot 0
    o[pp] = b;   // temporal slot 1 (duplicated frame, processor.rs:240)
  }
}
// Video frames -> patch rows (Qwen3VLProcessor::process_videos, processor.rs:253-281): (T, H, W, 3) u8 RGB frames -> rows
// (t, bh, bw, ih, iw) x columns (c, frame-in-pair, py, px), a temporal patch = 2 consecutive frames, an odd last frame repeated
// (process_vision_tensor, processor.rs:176-186).  Unlike img_transform (f32 until the final cast) the reference normalises videos
// in the MODEL dtype: u8 -> bf16, affine(1/255, 0), broadcast_sub(mean), broadcast_div(std), every op rounding to bf16
// (scalars cast to bf16 first) -- restated op by op here.
__global__ __launch_bounds__(192) void video_to_patches_kernel(const uint8_t* __restrict__ frames, bf16_t* __restrict__ out, int T,
                                                               int H, int W, int patch, int merge, float m0, float m1, float m2,
                                                               float s0, float s1, float s2) {
  const int gw = W / patch, gh = H / patch, bwn = gw / merge;
  const int t = blockIdx.x / (gh * gw), n = blockIdx.x % (gh * gw);
  const int iw = n % merge, ih = (n / merge) % merge, bw = (n / (merge * merge)) % bwn, bh = n / (merge * merge * bwn);
  const int y0 = (bh * merge + ih) * patch, x0 = (bw * merge + iw) * patch;
  const int pp = patch * patch;
  const float inv255 = rbf(1.0f / 255.0f);
  for (int e = threadIdx.x; e < 3 * pp; e += blockDim.x) {
    const int c = e / pp, py = (e % pp) / patch, px = e % patch;
    const float mean = rbf(c == 0 ? m0 : (c == 1 ? m1 : m2)), sd = rbf(c == 0 ? s0 : (c == 1 ? s1 : s2));
    bf16_t* o = out + (size_t)blockIdx.x * (6 * pp) + (size_t)c * 2 * pp + py * patch + px;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int f = min(2 * t + k, T - 1);
      const float v = (float)frames[(((size_t)f * H + (y0 + py)) * W + (x0 + px)) * 3 + c];
      o[k * pp] = f2bf(rbf(rbf(rbf(v * inv255) + 0.0f) - mean) / sd);
    }
  }
}
void launch_video_to_patches(const uint8_t* frames, void* out, int T, int H, int W, int patch, int merge, const float* mean,
                             const float* stdv, hipStream_t st) {
  const int n = ((T + 1) / 2) * (H / patch) * (W / patch);
  if (n <= 0) return;
  hipLaunchKernelGGL(video_to_patches_kernel, dim3(n), dim3(192), 0, st, frames, (bf16_t*)out, T, H, W, patch, merge, mean[0],
                     mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
}
void launch_image_to_patches(const uint8_t* img, void* out, int H, int W, int patch, int merge, const float* mean,
                             const float* stdv, hipStream_t st) {
  const int n = (H / patch) * (W / patch);
  if (n <= 0) return;
  hipLaunchKernelGGL(image_to_patches_kernel, dim3(n), dim3(192), 0, st, img, (bf16_t*)out, H, W, patch, merge, mean[0],
                     mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
}
}  // namespace aha
