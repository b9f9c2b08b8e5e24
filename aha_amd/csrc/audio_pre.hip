// A0-pre: the resampling step in front of the Whisper feature extractor -- `resample_audio_from_vec_f32`
// (reference src/utils/audio_utils.rs:590-616): interleaved PCM -> channel mean -> `resample_simple` (audio_utils.rs:247-255) =
// `resample` with sinc interpolation, Hann window, lowpass_filter_width 6, rolloff 0.99 (`get_sinc_resample_kernel`
// audio_utils.rs:66-151, `apply_sinc_resample_kernel` :154-214).  The Qwen3-ASR processor asks for 16 kHz mono
// (src/models/qwen3_asr/processor.rs:76,85).
//
// The taps are built on the host in f32 with the reference's operation order (they are a few hundred KB at most); the
// strided convolution runs on the GPU: out[i * new + j] = sum_k taps[j][k] * padded[i * orig + k].  HBM/LDS-bound FIR, f32.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "model.h"

namespace aha {
namespace {

// mono[f] = mean over channels (Tensor::mean_keepdim(1) on (frames, channels): f32 sum in channel order, then / channels)
__global__ void channel_mean_kernel(const float* pcm, float* mono, int64_t frames, int channels) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= frames) return;
  float s = 0.f;
  for (int c = 0; c < channels; ++c) s += pcm[f * channels + c];
  mono[f] = s / (float)channels;
}

// One block = RES_ROWS input hops (rows of `new_f` outputs each).  The block's input span (RES_ROWS * orig + klen samples,
// zero outside [0, length)) is staged in LDS once; thread t computes outputs j = t, t + 256, ... of every row.
constexpr int RES_ROWS = 4;
__global__ __launch_bounds__(256) void resample_kernel(const float* x, int64_t length, const float* taps, int orig, int new_f,
                                                       int klen, int width, float* out, int64_t out_len) {
  extern __shared__ float xs[];
  const int64_t row0 = (int64_t)blockIdx.x * RES_ROWS;
  const int span = (RES_ROWS - 1) * orig + klen;
  const int64_t base = row0 * orig - width;  // padded[p] = x[p - width]
  for (int i = threadIdx.x; i < span; i += 256) {
    const int64_t p = base + i;
    xs[i] = (p >= 0 && p < length) ? x[p] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < new_f; j += 256) {
    const float* tj = taps + (size_t)j * klen;
    float acc[RES_ROWS] = {};
    for (int k = 0; k < klen; ++k) {
      const float w = tj[k];
#pragma unroll
      for (int r = 0; r < RES_ROWS; ++r) acc[r] = fmaf(w, xs[r * orig + k], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < RES_ROWS; ++r) {
      const int64_t o = (row0 + r) * new_f + j;
      if (o < out_len) out[o] = acc[r];
    }
  }
}

}  // namespace

// get_sinc_resample_kernel (audio_utils.rs:66-151), SincInterpHann branch, every intermediate an f32 tensor as in the reference
void sinc_resample_taps(int64_t orig, int64_t new_f, int lowpass_filter_width, double rolloff, std::vector<float>& taps, int& width,
                        int& klen) {
  const double base_freq = (double)std::min(orig, new_f) * rolloff;
  width = (int)ceil((double)lowpass_filter_width * (double)orig / base_freq);
  klen = 2 * width + (int)orig;
  taps.resize((size_t)new_f * klen);
  const float inv_orig = (float)(1.0 / (double)orig), inv_new = (float)(1.0 / (double)new_f);
  const float base_f = (float)base_freq, lpw = (float)lowpass_filter_width;
  const float win_mul = (float)(M_PI / (double)lowpass_filter_width / 2.0), pi_f = (float)M_PI;
  const float scale = (float)(base_freq / (double)orig);
  for (int64_t j = 0; j < new_f; ++j) {
    const float tj = (float)(-j) * inv_new;            // arange_step(0, -new, -1).affine(1/new)
    for (int k = 0; k < klen; ++k) {
      const float idx = (float)(k - width) * inv_orig;  // arange(-width, width + orig).affine(1/orig)
      float t = (tj + idx) * base_f;                    // broadcast_add, affine(base_freq)
      t = fminf(fmaxf(t, -lpw), lpw);                   // clamp
      const float c = cosf(t * win_mul);
      const float window = c * c;                       // cos().sqr()
      const float ts = t * pi_f;
      const float sinc = ts == 0.f ? 1.f : sinf(ts) / ts;
      taps[(size_t)j * klen + k] = sinc * window * scale;
    }
  }
}

// host-only view of the polyphase taps (orig / new already divided by their gcd), for the CPU test tier
int64_t debug_resample_taps(int64_t orig, int64_t new_f, float* taps, int64_t cap, int32_t* width, int32_t* klen) {
  std::vector<float> t;
  int w = 0, k = 0;
  sinc_resample_taps(orig, new_f, 6, 0.99, t, w, k);
  if ((int64_t)t.size() > cap) {
    set_error("debug_resample_taps: buffer too small");
    return AHA_ERR_INVALID;
  }
  std::copy(t.begin(), t.end(), taps);
  *width = w;
  *klen = k;
  return (int64_t)t.size();
}

int64_t resample_output_len(int64_t length, int64_t orig_sr, int64_t target_sr) {
  if (orig_sr == target_sr) return length;
  const int64_t g = std::gcd(orig_sr, target_sr);
  const int64_t orig = orig_sr / g, new_f = target_sr / g;
  // conv output rows = (length + 2*width + orig - klen) / orig + 1 = length / orig + 1; narrow to ceil(new * length / orig)
  const int64_t rows = length / orig + 1;
  const int64_t target = (int64_t)ceil((double)new_f * (double)length / (double)orig);
  return std::min(target, rows * new_f);
}

int64_t audio_resample(aha_ctx* ctx, const float* pcm, int64_t n_frames, int channels, int orig_sr, int target_sr, float* out,
                       int64_t out_cap) {
  const int64_t out_len = resample_output_len(n_frames, orig_sr, target_sr);
  if (!out) return out_len;
  if (out_cap < out_len) {
    set_error("audio_resample: output buffer too small");
    return AHA_ERR_INVALID;
  }
  if (out_len == 0) return 0;
  if (channels == 1 && orig_sr == target_sr) {  // `resample` returns the waveform unchanged (audio_utils.rs:227-229)
    memcpy(out, pcm, (size_t)n_frames * 4);
    return out_len;
  }
  AHA_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  float *d_pcm = nullptr, *d_mono = nullptr, *d_out = nullptr, *d_taps = nullptr;
  auto cleanup = [&] {
    hipFree(d_pcm); hipFree(d_mono); hipFree(d_out); hipFree(d_taps);
  };
  const size_t pcm_bytes = (size_t)n_frames * channels * 4;
  hipError_t e = hipMalloc((void**)&d_pcm, pcm_bytes);
  if (e == hipSuccess && channels > 1) e = hipMalloc((void**)&d_mono, (size_t)n_frames * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pcm, pcm, pcm_bytes, hipMemcpyHostToDevice, st);
  const float* mono = d_pcm;
  if (e == hipSuccess && channels > 1) {
    hipLaunchKernelGGL(channel_mean_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, st, d_pcm, d_mono, n_frames, channels);
    mono = d_mono;
  }
  if (e == hipSuccess && orig_sr == target_sr) {
    e = hipMemcpyAsync(out, mono, (size_t)n_frames * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    AHA_HIP_CHECK(e);
    return out_len;
  }
  const int64_t g = std::gcd((int64_t)orig_sr, (int64_t)target_sr);
  const int64_t orig = orig_sr / g, new_f = target_sr / g;
  std::vector<float> taps;
  int width = 0, klen = 0;
  sinc_resample_taps(orig, new_f, 6, 0.99, taps, width, klen);
  const size_t lds = ((size_t)(RES_ROWS - 1) * orig + klen) * 4;
  if (lds > 160 * 1024) {
    cleanup();
    set_error("audio_resample: sample-rate ratio too irregular for the LDS-staged kernel (orig / gcd too large)");
    return AHA_ERR_UNSUPPORTED;
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d_taps, taps.size() * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_out, (size_t)out_len * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    if (lds > 64 * 1024) e = hipFuncSetAttribute((const void*)resample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (e == hipSuccess) {
    const int64_t rows = (out_len + new_f - 1) / new_f;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((rows + RES_ROWS - 1) / RES_ROWS)), dim3(256), lds, st, mono, n_frames, d_taps,
                       (int)orig, (int)new_f, klen, width, d_out, out_len);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)out_len * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  AHA_HIP_CHECK(e);
  return out_len;
}

}  // namespace aha
