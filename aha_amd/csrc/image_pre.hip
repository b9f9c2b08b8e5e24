// V0-pre: the resize in front of the Qwen3-VL patchifier -- `Qwen3VLProcessor::process_img` (reference
// src/models/qwen3vl/processor.rs:150-171): `img_smart_resize` (src/utils/img_utils.rs:294-331, host arithmetic) and
// `DynamicImage::resize_exact(w, h, FilterType::CatmullRom)` = crate `image` 0.25.10 `imageops::resize`: a vertical pass into an
// f32 image, then a horizontal pass, per output sample the normalised CatmullRom weights of the taps left..right (kernel
// argument scaled by max(ratio, 1)), f32 accumulation in tap order, clamp to [0, 255], round half away from zero.
//
// The tap tables (left index, count, weights) are built on the host in f32 with the operation order of the restatement in
// oracle/image_pre.py; the two passes run on the GPU with un-fused multiplies and adds (fp contract off) in tap order, so the u8 result is
// bit-identical to that restatement.  HBM-bound: the f32 intermediate is written and read once.
#include <math.h>

#include <algorithm>
#include <string>
#include <vector>

#include "model.h"

namespace aha {
namespace {

float bc_cubic_spline(float x) {  // B = 0, C = 0.5
  const float a = fabsf(x), b = 0.0f, c = 0.5f;
  float k;
  if (a < 1.0f) k = (12.0f - 9.0f * b - 6.0f * c) * (a * a * a) + (-18.0f + 12.0f * b + 6.0f * c) * (a * a) + (6.0f - 2.0f * b);
  else if (a < 2.0f) k = (-b - 6.0f * c) * (a * a * a) + (6.0f * b + 30.0f * c) * (a * a) + (-12.0f * b - 48.0f * c) * a + (8.0f * b + 24.0f * c);
  else k = 0.0f;
  return k / 6.0f;
}

struct Taps {
  std::vector<int> left, count, offset;
  std::vector<float> w;
};

// the loop head shared by vertical_sample / horizontal_sample; volatile keeps every f32 operation a separate rounding
Taps build_taps(int n_in, int n_out) {
  Taps t;
  const float ratio = (float)n_in / (float)n_out;
  const float sratio = ratio < 1.0f ? 1.0f : ratio;
  const float src_support = 2.0f * sratio;
  for (int o = 0; o < n_out; ++o) {
    volatile float inp = ((float)o + 0.5f) * ratio;
    long left = (long)floorf(inp - src_support);
    left = std::min<long>(std::max<long>(left, 0), n_in - 1);
    long right = (long)ceilf(inp + src_support);
    right = std::min<long>(std::max<long>(right, left + 1), n_in);
    inp = inp - 0.5f;
    t.left.push_back((int)left);
    t.count.push_back((int)(right - left));
    t.offset.push_back((int)t.w.size());
    volatile float sum = 0.0f;
    const size_t base = t.w.size();
    for (long i = left; i < right; ++i) {
      volatile float arg = ((float)i - inp) / sratio;
      const float w = bc_cubic_spline(arg);
      t.w.push_back(w);
      sum = sum + w;
    }
    for (size_t i = base; i < t.w.size(); ++i) t.w[i] = t.w[i] / sum;
  }
  return t;
}

// tmp[oy][x][c] = sum_i w[oy][i] * src[left[oy] + i][x][c]
__global__ void resize_vertical_kernel(const uint8_t* src, float* tmp, int W3, const int* left, const int* count, const int* offset,
                                       const float* w, int new_h) {
#pragma clang fp contract(off)  // t += p * w is two roundings in the crate; HIP's __fmul_rn / __fadd_rn are plain operators
  const int xc = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (xc >= W3) return;
  const int l = left[oy], n = count[oy];
  const float* ww = w + offset[oy];
  float t = 0.0f;
  for (int i = 0; i < n; ++i) {
    const float p = (float)src[(size_t)(l + i) * W3 + xc] * ww[i];
    t = t + p;
  }
  tmp[(size_t)oy * W3 + xc] = t;
}

// dst[oy][ox][c] = round(clamp(sum_i w[ox][i] * tmp[oy][left[ox] + i][c], 0, 255))
__global__ void resize_horizontal_kernel(const float* tmp, uint8_t* dst, int W, int new_w, const int* left, const int* count,
                                         const int* offset, const float* w) {
#pragma clang fp contract(off)
  const int oxc = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (oxc >= new_w * 3) return;
  const int ox = oxc / 3, c = oxc - ox * 3;
  const int l = left[ox], n = count[ox];
  const float* ww = w + offset[ox];
  const float* row = tmp + (size_t)oy * W * 3;
  float t = 0.0f;
  for (int i = 0; i < n; ++i) {
    const float p = row[(size_t)(l + i) * 3 + c] * ww[i];
    t = t + p;
  }
  t = fminf(fmaxf(t, 0.0f), 255.0f);
  dst[(size_t)oy * new_w * 3 + oxc] = (uint8_t)roundf(t);  // f32::round: half away from zero
}

int upload(const Taps& t, int** d_int, float** d_w, hipStream_t st) {
  const size_t n = t.left.size();
  std::vector<int> pack(3 * n);
  std::copy(t.left.begin(), t.left.end(), pack.begin());
  std::copy(t.count.begin(), t.count.end(), pack.begin() + n);
  std::copy(t.offset.begin(), t.offset.end(), pack.begin() + 2 * n);
  AHA_HIP_CHECK(hipMalloc((void**)d_int, pack.size() * 4));
  AHA_HIP_CHECK(hipMalloc((void**)d_w, t.w.size() * 4));
  AHA_HIP_CHECK(hipMemcpy(*d_int, pack.data(), pack.size() * 4, hipMemcpyHostToDevice));
  AHA_HIP_CHECK(hipMemcpy(*d_w, t.w.data(), t.w.size() * 4, hipMemcpyHostToDevice));
  (void)st;
  return AHA_OK;
}

}  // namespace

// host-only view of the tap tables (tests compare them bit for bit with the restatement without a GPU)
int debug_resize_taps(int n_in, int n_out, int32_t* left, int32_t* count, float* weights, int64_t weights_cap) {
  const Taps t = build_taps(n_in, n_out);
  if ((int64_t)t.w.size() > weights_cap) {
    set_error("debug_resize_taps: weights buffer too small");
    return AHA_ERR_INVALID;
  }
  std::copy(t.left.begin(), t.left.end(), left);
  std::copy(t.count.begin(), t.count.end(), count);
  std::copy(t.w.begin(), t.w.end(), weights);
  return (int)t.w.size();
}

// img_smart_resize (img_utils.rs:294-331): f32 beta, round / floor / ceil to a multiple of `factor`
int img_smart_resize(uint32_t h, uint32_t w, uint32_t factor, uint32_t min_pixels, uint32_t max_pixels, uint32_t* h_out, uint32_t* w_out) {
  if (h == 0 || w == 0 || factor == 0) {
    set_error("img_smart_resize: zero dimension");
    return AHA_ERR_INVALID;
  }
  if (std::max(h, w) / std::min(h, w) > 200) {
    set_error("absolute aspect ratio mush be smaller than 200");  // the reference's message (img_utils.rs:302-306)
    return AHA_ERR_INVALID;
  }
  auto round_by = [&](uint32_t v) { return (uint32_t)roundf((float)v / (float)factor) * factor; };
  auto floor_by = [&](float v) { return (uint32_t)floorf(v / (float)factor) * factor; };
  auto ceil_by = [&](float v) { return (uint32_t)ceilf(v / (float)factor) * factor; };
  uint32_t hb = std::max(factor, round_by(h)), wb = std::max(factor, round_by(w));
  if ((uint64_t)hb * wb > max_pixels) {
    const float beta = sqrtf((float)(h * w) / (float)max_pixels);
    hb = std::max(factor, floor_by((float)h / beta));
    wb = std::max(factor, floor_by((float)w / beta));
  } else if ((uint64_t)hb * wb < min_pixels) {
    const float beta = sqrtf((float)min_pixels / (float)(h * w));
    hb = ceil_by((float)h * beta);
    wb = ceil_by((float)w * beta);
  }
  *h_out = hb;
  *w_out = wb;
  return AHA_OK;
}

// ---- the video path's host arithmetic (host only; decoding and the swscale resize are ffmpeg's and stay with the caller) --------
// video_smart_resize (/root/reference/src/utils/video_utils.rs:9-59): the frame size get_video_data scales to.  The reference
// multiplies in u32 (t_bar * h_bar * w_bar, num_frames * height * width: past 2^32 a debug build panics and a release build
// wraps); the products are taken in 64 bits here.
int video_smart_resize(uint32_t num_frames, uint32_t h, uint32_t w, uint32_t temporal_factor, uint32_t factor, uint32_t min_pixels,
                       uint32_t max_pixels, uint32_t video_ratio, uint32_t* h_out, uint32_t* w_out) {
  if (factor == 0 || temporal_factor == 0) {
    set_error("video_smart_resize: zero factor");
    return AHA_ERR_INVALID;
  }
  if (num_frames < temporal_factor) {
    set_error(std::to_string(num_frames) + " must be larger than temporal_factor " + std::to_string(temporal_factor));
    return AHA_ERR_INVALID;
  }
  if (h < factor || w < factor) {
    set_error("height:" + std::to_string(h) + " or width:" + std::to_string(w) + " must be larger than factor:" + std::to_string(factor));
    return AHA_ERR_INVALID;
  }
  if (std::max(h, w) / std::min(h, w) > 200) {
    set_error("absolute aspect ratio mush be smaller than 200, got " + std::to_string(std::max(h, w) / std::min(h, w)));
    return AHA_ERR_INVALID;
  }
  uint32_t f = factor;
  if (video_ratio) {  // lcm(image_factor, ratio): swscale wants a multiple of 16 (processor.rs:492-505)
    uint32_t a = factor, b = video_ratio;
    while (b) { const uint32_t t = a % b; a = b; b = t; }
    f = factor / a * video_ratio;
  }
  auto round_by = [](uint32_t v, uint32_t k) { return (uint32_t)roundf((float)v / (float)k) * k; };
  uint32_t hb = round_by(h, f), wb = round_by(w, f);
  const uint64_t tb = round_by(num_frames, temporal_factor);
  const float vol = (float)((uint64_t)num_frames * h * w);
  if (tb * hb * wb > max_pixels) {
    const float beta = sqrtf(vol / (float)max_pixels);
    hb = std::max(f, (uint32_t)floorf((float)h / beta / (float)f) * f);
    wb = std::max(f, (uint32_t)floorf((float)w / beta / (float)f) * f);
  } else if (tb * hb * wb < min_pixels) {
    const float beta = sqrtf((float)min_pixels / vol);
    hb = (uint32_t)ceilf((float)h * beta / (float)f) * f;
    wb = (uint32_t)ceilf((float)w * beta / (float)f) * f;
  }
  *h_out = hb;
  *w_out = wb;
  return AHA_OK;
}

// Which decoded frames get_video_data keeps (/root/reference/src/models/qwen3vl/processor.rs:481-489,518-535): nframes =
// round(frames / rate * fps) clamped to [min_frames, max_frames] and to the frame count (it only sizes the resize), and every
// frame whose id is a multiple of round(frames / nframes).
int video_sample_frames(uint32_t total_frames, float rate, uint32_t fps, uint32_t min_frames, uint32_t max_frames, uint32_t* nframes,
                        uint32_t* interval) {
  if (total_frames == 0 || !(rate > 0.f)) {
    set_error("video_sample_frames: no frames / no frame rate");
    return AHA_ERR_INVALID;
  }
  uint32_t n = (uint32_t)roundf((float)total_frames / rate * (float)fps);
  n = std::min(std::min(std::max(n, min_frames), max_frames), total_frames);
  if (n == 0) {
    set_error("video_sample_frames: zero frames requested");
    return AHA_ERR_INVALID;
  }
  *nframes = n;
  *interval = (uint32_t)roundf((float)total_frames / (float)n);
  return AHA_OK;
}

// calculate_timestamps (processor.rs:283-307): frame indices padded to a multiple of t_merge with the last one, index / fps in
// f32, the mean of the first and last time of every group.  Returns the number of stamps (written up to cap).
int64_t video_timestamps(const uint32_t* frame_indices, size_t n, float fps, uint32_t t_merge, float* out, size_t cap) {
  if (!frame_indices || n == 0 || t_merge == 0 || !(fps > 0.f)) {
    set_error("video_timestamps: bad argument");
    return AHA_ERR_INVALID;
  }
  const size_t padded = (n + t_merge - 1) / t_merge * t_merge;
  auto ts = [&](size_t i) { return (float)frame_indices[std::min(i, n - 1)] / fps; };
  size_t k = 0;
  for (size_t i = 0; i < padded; i += t_merge, ++k)
    if (out && k < cap) out[k] = (ts(i) + ts(i + t_merge - 1)) / 2.0f;
  return (int64_t)k;
}

int image_resize(const uint8_t* src, int H, int W, uint8_t* dst, int new_h, int new_w, hipStream_t st) {
  if (new_h == H && new_w == W) {  // imageops::resize copies when the dimensions are unchanged
    AHA_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)H * W * 3, hipMemcpyDeviceToDevice, st));
    return AHA_OK;
  }
  const Taps tv = build_taps(H, new_h), th = build_taps(W, new_w);
  int *dv = nullptr, *dh = nullptr;
  float *wv = nullptr, *wh = nullptr, *tmp = nullptr;
  int rc = upload(tv, &dv, &wv, st);
  if (!rc) rc = upload(th, &dh, &wh, st);
  hipError_t e = rc ? hipSuccess : hipMalloc((void**)&tmp, (size_t)new_h * W * 3 * 4);
  if (!rc && e == hipSuccess) {
    const int W3 = W * 3;
    hipLaunchKernelGGL(resize_vertical_kernel, dim3((W3 + 255) / 256, new_h), dim3(256), 0, st, src, tmp, W3, dv, dv + new_h,
                       dv + 2 * new_h, wv, new_h);
    hipLaunchKernelGGL(resize_horizontal_kernel, dim3((new_w * 3 + 255) / 256, new_h), dim3(256), 0, st, tmp, dst, W, new_w, dh,
                       dh + new_w, dh + 2 * new_w, wh);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the tap tables and the intermediate are freed below
  }
  hipFree(dv); hipFree(dh); hipFree(wv); hipFree(wh); hipFree(tmp);
  if (rc) return rc;
  AHA_HIP_CHECK(e);
  return AHA_OK;
}

}  // namespace aha
