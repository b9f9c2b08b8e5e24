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
constexpr int VIT_HEADS_PER_WAVE = 4;  // cos/sin depend on (patch, lane) only
__global__ __launch_bounds__(256) void vit_rope_pack_kernel(VitRopeArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ngrp = (a.nh + VIT_HEADS_PER_WAVE - 1) / VIT_HEADS_PER_WAVE;
  if (wid >= (int64_t)a.N * ngrp) return;
  const int n = (int)(wid / ngrp), h0 = (int)(wid % ngrp) * VIT_HEADS_PER_WAVE;
  const int hd = a.hd, half = hd / 2, quarter = hd / 4;
  const int D = a.nh * hd;
  const int page = a.page_of[n], slot = a.slot_of[n];
  bf16_t* pbase = reinterpret_cast<bf16_t*>(a.kv.page_ptrs[page] + a.kv.layer_off);
  float c = 0.f, s = 0.f;
  if (lane < half) {
    const int pos = (lane < quarter) ? a.rowcol[2 * n] : a.rowcol[2 * n + 1];
    const float ang = (float)pos * a.inv_freq[lane % quarter];
    c = rbf(cosf(ang));
    s = rbf(sinf(ang));
  }
  for (int h = h0; h < min(h0 + VIT_HEADS_PER_WAVE, a.nh); ++h) {
    const bf16_t* src = (const bf16_t*)a.qkv + (int64_t)n * 3 * D + (int64_t)h * hd;
    bf16_t* qd = (bf16_t*)a.q_out + ((int64_t)n * a.nh + h) * VIT_DQK;
    bf16_t* kd = pbase + (int64_t)h * KV_PAGE_TOKENS * VIT_DQK;   // fragment-major K block of the head (common.h kpage_elem)
    bf16_t* vd = pbase + (int64_t)a.nh * KV_PAGE_TOKENS * VIT_DQK + (int64_t)h * VIT_DV * KV_PAGE_TOKENS;
    if (lane < half) {
      const int e = lane;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const bf16_t* p = src + (int64_t)which * D;
        const float x0 = bf2f(p[e]), x1 = bf2f(p[e + half]);
        const bf16_t y0 = f2bf(rbf(x0 * c) + rbf(-x1 * s));
        const bf16_t y1 = f2bf(rbf(x1 * c) + rbf(x0 * s));
        if (which) {
          kd[kpage_elem(slot, e, VIT_DQK / 32)] = y0;
          kd[kpage_elem(slot, e + half, VIT_DQK / 32)] = y1;
        } else {
          qd[e] = y0;
          qd[e + half] = y1;
        }
      }
    }
    // zero the pad lanes of q and k rows [hd, VIT_DQK)
    if (lane < VIT_DQK - hd) {
      qd[hd + lane] = 0;
      kd[kpage_elem(slot, hd + lane, VIT_DQK / 32)] = 0;
    }
    // V: fragment-major (common.h vpage_elem); pad rows [hd, VIT_DV) zero
    const bf16_t* vp = src + 2 * (int64_t)D;
    for (int e = lane; e < VIT_DV; e += 64) vd[vpage_elem(slot, e)] = (e < hd) ? vp[e] : (bf16_t)0;
  }
}
void launch_vit_rope_pack(const VitRopeArgs& a, hipStream_t st) {
  const int64_t waves = (int64_t)a.N * ((a.nh + VIT_HEADS_PER_WAVE - 1) / VIT_HEADS_PER_WAVE);
  if (waves <= 0) return;
  hipLaunchKernelGGL(vit_rope_pack_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
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
void launch_image_to_patches(const uint8_t* img, void* out, int H, int W, int patch, int merge, const float* mean,
                             const float* stdv, hipStream_t st) {
  const int n = (H / patch) * (W / patch);
  if (n <= 0) return;
  hipLaunchKernelGGL(image_to_patches_kernel, dim3(n), dim3(192), 0, st, img, (bf16_t*)out, H, W, patch, merge, mean[0],
                     mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
}
}  // namespace aha
