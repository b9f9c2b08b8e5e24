// Qwen3-ASR audio path kernels: Whisper log-mel frontend, the strided 3x3 convolution
// stack as im2col feeds for the MFMA GEMM, sinusoidal position add, generic K/V page packing for the encoder attention.
#include "common.h"
#include "kernels.h"

namespace aha {

// ---- A0: frames x Hann -> |DFT_400|^2 -> mel(201 -> 128) -> log10 (feature_extraction_whisper.rs:93-115) -----------------
// One block per FOUR STFT frames (round 6; one per frame before).  The 400-point real DFT is evaluated directly (201 bins x 400 taps, f32 FMA,
// twiddles from a 400-entry LDS table of (cos, sin) pairs indexed by k*n mod 400): 0.2 GFLOP for 30 s of audio, no FFT library.  The table
// reads are the cost -- lane k reads entry k*n mod 400, a many-way bank conflict for most n -- so one 8-byte read now serves four frames'
// taps (their samples are one broadcast 16-byte read), and likewise a mel weight serves four frames: 136 -> ~40 us at 3000 frames.
// Every (frame, bin) and (frame, mel) sum keeps its order (n = 0..399, k = 0..200): bit-identical to the one-frame kernel.
// Frame f covers padded samples [160 f, 160 f + 400) of reflect-padded input (pad 200 each side); the right pad carries
// the reference's indexing quirk (tensor_utils.rs:525-549): it mirrors original samples [L-400, L-200), not [L-201, L-1).
constexpr int MEL_FPB = 4;   // frames per block
__global__ __launch_bounds__(256) void logmel_power_kernel(const float* __restrict__ x, int64_t L, const float* __restrict__ window,
                                                           const float* __restrict__ twid,  // (400,2): cos, sin of 2 pi j / 400
                                                           const float* __restrict__ melfb, // (201,128)
                                                           float* __restrict__ out,         // (128, F) log10 mel
                                                           float* __restrict__ frame_max, int F) {
  __shared__ float4 fr[400];     // [tap][frame of the block]
  __shared__ float2 tw[400];
  __shared__ float4 pw[208];     // [bin][frame of the block]
  __shared__ float red[MEL_FPB][4];
  const int f0 = blockIdx.x * MEL_FPB, tid = threadIdx.x;
  for (int n = tid; n < 400; n += 256) {
    float v[MEL_FPB];
#pragma unroll
    for (int q = 0; q < MEL_FPB; ++q) {
      const int64_t p = (int64_t)160 * (f0 + q) + n;  // index into the padded signal of length L + 400
      int64_t src;
      if (p < 200) src = 200 - p;                       // left reflect: padded[p] = x[200 - p]
      else if (p < 200 + L) src = p - 200;
      else src = (L - 200 - 1) - (p - (200 + L));       // right pad (quirk): reversed x[L-400 .. L-200)
      v[q] = f0 + q < F ? x[src] * window[n] : 0.f;
    }
    fr[n] = make_float4(v[0], v[1], v[2], v[3]);
    tw[n] = make_float2(twid[2 * n], twid[2 * n + 1]);
  }
  __syncthreads();
  if (tid < 201) {
    float re[MEL_FPB] = {0.f, 0.f, 0.f, 0.f}, im[MEL_FPB] = {0.f, 0.f, 0.f, 0.f};
    int j = 0;  // (k*n) mod 400
    for (int n = 0; n < 400; ++n) {
      const float4 v4 = fr[n];
      const float2 t = tw[j];
      // (sin in a register of its own: clang packs the four frames' FMAs in pairs and, with (cos, sin) as one register pair, broadcasts
      // the pair's HIGH element into both halves -- v_pk_fma_f32 op_sel:[0,1,0], the form that reads the wrong element beside a kernel of
      // another stream on MI355X: tests/test_isa_cpu.py, profiles/r03_simd_coresidency.md)
      float tx = t.x, ty = t.y;
      asm volatile("" : "+v"(tx), "+v"(ty));
      const float v[MEL_FPB] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int q = 0; q < MEL_FPB; ++q) {
        re[q] = fmaf(v[q], tx, re[q]);
        im[q] = fmaf(v[q], ty, im[q]);
      }
      j += tid;
      if (j >= 400) j -= 400;
    }
    // norm_sqr (audio_utils.rs:1303-1311)
    pw[tid] = make_float4(re[0] * re[0] + im[0] * im[0], re[1] * re[1] + im[1] * im[1], re[2] * re[2] + im[2] * im[2], re[3] * re[3] + im[3] * im[3]);
  }
  __syncthreads();
  float lg[MEL_FPB] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  if (tid < 128) {
    float m[MEL_FPB] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 201; ++k) {
      const float w = melfb[k * 128 + tid];
      const float4 p4 = pw[k];
      m[0] = fmaf(w, p4.x, m[0]);
      m[1] = fmaf(w, p4.y, m[1]);
      m[2] = fmaf(w, p4.z, m[2]);
      m[3] = fmaf(w, p4.w, m[3]);
    }
#pragma unroll
    for (int q = 0; q < MEL_FPB; ++q) {
      if (f0 + q >= F) continue;
      const float mm = fmaxf(m[q], 1e-10f);
      lg[q] = logf(mm) * (float)(1.0 / 2.302585092994046);  // log10 = ln * (1/ln 10) (modules.rs:1256-1258)
      out[(int64_t)tid * F + f0 + q] = lg[q];
    }
  }
#pragma unroll
  for (int q = 0; q < MEL_FPB; ++q) {
    const float g = wave_max(lg[q]);
    if ((tid & 63) == 0) red[q][tid >> 6] = g;
  }
  __syncthreads();
  if (tid < MEL_FPB && f0 + tid < F) frame_max[f0 + tid] = fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3]));
}
// max over frames, then x = (max(x, gmax - 8) + 4) * 0.25 (feature_extraction_whisper.rs:110-113)
__global__ __launch_bounds__(256) void logmel_finalize_kernel(float* __restrict__ out, const float* __restrict__ frame_max, int F) {
  __shared__ float red[4];
  float g = -INFINITY;
  for (int i = threadIdx.x; i < F; i += 256) g = fmaxf(g, frame_max[i]);
  g = wave_max(g);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = g;
  __syncthreads();
  g = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - 8.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)128 * F; i += (int64_t)gridDim.x * 256)
    out[i] = (fmaxf(out[i], g) * 1.0f + 4.0f) * 0.25f;
}
void launch_logmel(const float* x, int64_t L, const float* window, const float* twid, const float* melfb, float* out,
                   float* frame_max, int F, hipStream_t st) {
  if (F <= 0) return;
  hipLaunchKernelGGL(logmel_power_kernel, dim3((F + MEL_FPB - 1) / MEL_FPB), dim3(256), 0, st, x, L, window, twid, melfb, out, frame_max, F);
  hipLaunchKernelGGL(logmel_finalize_kernel, dim3(64), dim3(256), 0, st, out, frame_max, F);
}

// ---- A1: conv2d 3x3, stride 2, pad 1 as im2col + GEMM (qwen3_asr/model.rs:196-204) ----------------------------------------
// conv1 (1 input channel): mel features (128, F) f32 -> bf16 (input_features.to_dtype(model dtype)), chunked along time
// into C windows of 100 frames (zero padded), rows = (chunk, ho, wo) over the (64, 50) output grid, 16 columns (9 used).
__global__ __launch_bounds__(256) void audio_im2col1_kernel(const float* __restrict__ feat, bf16_t* __restrict__ out, int F,
                                                            int64_t rows, int Hin, int Win, int Ho, int Wo) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho), ch = (int)(r / ((int64_t)Wo * Ho));
  uint32_t v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0u;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int h = 2 * ho - 1 + t / 3, w = 2 * wo - 1 + t % 3;
    float x = 0.f;
    if (h >= 0 && h < Hin && w >= 0 && w < Win) {
      const int64_t fw = (int64_t)ch * Win + w;
      if (fw < F) x = bf2f(f2bf(feat[(int64_t)h * F + fw]));
    }
    const uint32_t b = f2bf(x);
    v[t >> 1] |= (t & 1) ? (b << 16) : b;
  }
  u32x4_t* o = reinterpret_cast<u32x4_t*>(out + r * 16);
  o[0] = u32x4_t{v[0], v[1], v[2], v[3]};
  o[1] = u32x4_t{v[4], v[5], v[6], v[7]};
}
void launch_audio_im2col1(const float* feat, void* out, int F, int C, int Hin, int Win, hipStream_t st) {
  const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2;
  const int64_t rows = (int64_t)C * Ho * Wo;
  hipLaunchKernelGGL(audio_im2col1_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, feat, (bf16_t*)out, F, rows,
                     Hin, Win, Ho, Wo);
}
// NHWC (B, Hin, Win, Cin) bf16 -> rows (b, ho, wo) x columns (kh, kw, cin); one wave per (row, tap), 16-byte copies.
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int64_t rows,
                                                          int Hin, int Win, int Ho, int Wo, int Cin) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= rows * 9) return;
  const int t = (int)(wid % 9);
  const int64_t r = wid / 9;
  const int wo = (int)(r % Wo), ho = (int)((r / Wo) % Ho);
  const int64_t b = r / ((int64_t)Wo * Ho);
  const int h = 2 * ho - 1 + t / 3, w = 2 * wo - 1 + t % 3;
  const bool ok = h >= 0 && h < Hin && w >= 0 && w < Win;
  const bf16_t* src = in + ((b * Hin + (ok ? h : 0)) * Win + (ok ? w : 0)) * Cin;
  bf16_t* dst = out + r * (9 * (int64_t)Cin) + (int64_t)t * Cin;
  for (int v = lane; v < Cin / 8; v += 64)
    *reinterpret_cast<u32x4_t*>(dst + v * 8) = ok ? ld16(src + v * 8) : u32x4_t{0u, 0u, 0u, 0u};
}
void launch_im2col_nhwc(const void* in, void* out, int B, int Hin, int Win, int Cin, hipStream_t st) {
  const int Ho = (Hin + 1) / 2, Wo = (Win + 1) / 2;
  const int64_t rows = (int64_t)B * Ho * Wo;
  hipLaunchKernelGGL(im2col_nhwc_kernel, dim3((unsigned)((rows * 9 + 3) / 4)), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out,
                     rows, Hin, Win, Ho, Wo, Cin);
}
// conv3 output NHWC (B, Fq, T, Cc) -> (B*T, Cc*Fq) with feature index c*Fq + f  (permute(0,3,1,2).reshape, model.rs:206-211)
__global__ __launch_bounds__(256) void audio_tokens_gather_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B,
                                                                  int Fq, int T, int Cc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * T * Cc * Fq;
  if (i >= total) return;
  const int f = (int)(i % Fq), c = (int)((i / Fq) % Cc);
  const int64_t bt = i / ((int64_t)Fq * Cc);
  const int t = (int)(bt % T);
  const int64_t b = bt / T;
  out[i] = in[((b * Fq + f) * T + t) * Cc + c];
}
void launch_audio_tokens_gather(const void* in, void* out, int B, int Fq, int T, int Cc, hipStream_t st) {
  const int64_t total = (int64_t)B * T * Cc * Fq;
  hipLaunchKernelGGL(audio_tokens_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)in,
                     (bf16_t*)out, B, Fq, T, Cc);
}
// x[r, :] = bf16(x + bf16(pe[r % T])), pe[p] = cat(sin(p w), cos(p w)), w_i = 1/10000^(2i/d)  (sinusoidal_pe.rs:22-58)
__global__ __launch_bounds__(256) void sinus_pe_add_kernel(bf16_t* __restrict__ x, int64_t rows, int d, int T) {
  const int64_t r = blockIdx.x;
  const int p = (int)(r % T), half = d / 2;
  for (int j = threadIdx.x; j < d; j += 256) {
    const int i = j < half ? j : j - half;
    const float inv = 1.0f / powf(10000.0f, (float)(2 * i) / (float)d);
    const float ang = (float)p * inv;
    const float pe = rbf(j < half ? sinf(ang) : cosf(ang));
    x[r * d + j] = f2bf(bf2f(x[r * d + j]) + pe);
  }
}
void launch_sinus_pe_add(void* x, int64_t rows, int d, int T, hipStream_t st) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(sinus_pe_add_kernel, dim3((unsigned)rows), dim3(256), 0, st, (bf16_t*)x, rows, d, T);
}

// ---- A2 staging: K / V of a fused qkv activation -> attention pages (head_dim hd, multiple of 32) ----------------------
// Fragment-major page blocks (common.h kpage_elem / vpage_elem): K block [nh][64 * hd], V block [nh][hd * 64].
__global__ __launch_bounds__(256) void kv_pack_generic_kernel(const bf16_t* __restrict__ src, int64_t ld, int k_off, int v_off,
                                                              KvLayer kv, int N, int nh, int hd) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)N * nh) return;
  const int n = (int)(wid / nh), h = (int)(wid % nh);
  const int page = n / KV_PAGE_TOKENS, slot = n % KV_PAGE_TOKENS;
  bf16_t* base = reinterpret_cast<bf16_t*>(kv.page_ptrs[page] + kv.layer_off);
  bf16_t* kd = base + (int64_t)h * KV_PAGE_TOKENS * hd;
  bf16_t* vd = base + (int64_t)nh * KV_PAGE_TOKENS * hd + (int64_t)h * hd * KV_PAGE_TOKENS;
  const bf16_t* ks = src + (int64_t)n * ld + k_off + (int64_t)h * hd;
  const bf16_t* vs = src + (int64_t)n * ld + v_off + (int64_t)h * hd;
  for (int e = lane; e < hd; e += 64) {
    kd[kpage_elem(slot, e, hd / 32)] = ks[e];
    vd[vpage_elem(slot, e)] = vs[e];
  }
}
void launch_kv_pack_generic(const void* src, int64_t ld, int k_off, int v_off, KvLayer kv, int N, int nh, int hd, hipStream_t st) {
  const int64_t waves = (int64_t)N * nh;
  if (waves <= 0) return;
  hipLaunchKernelGGL(kv_pack_generic_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const bf16_t*)src, ld, k_off,
                     v_off, kv, N, nh, hd);
}

}  // namespace aha
