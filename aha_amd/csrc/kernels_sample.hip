// D11 on the device: the deterministic half of `sample_and_push` (reference src/models/common/generate.rs:70-86) --
// repeat penalty (src/models/common/sample.rs:41-60 -> candle_transformers::utils::apply_repeat_penalty) and the candidate
// set of candle's LogitsProcessor (Sampling::TopK / TopKThenTopP / TopP, sample.rs:7-38): the k largest penalised logits
// with their vocabulary indices, plus the max and the sum of exp((x - max) / T) over the WHOLE vocabulary, so that the host
// gets the same probabilities candle's full-vocabulary softmax would give those k tokens.  The random draw stays on the host
// (the caller's RNG): 8k + 8 bytes cross PCIe per token instead of the 608 KB logits vector.
//
// HBM-bound integer/compare work: one pass over V f32 logits (608 KB), no LDS staging needed.  Selection is k rounds of
// (wave max, lowest index among the maxima) over register-resident candidates: exact, ordered by (value desc, index asc).
#include "common.h"
#include "kernels.h"

namespace aha {
namespace {

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = min(r[0], r[1]);
  }
  {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = min(r[0], r[1]);
  }
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false));
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false));
  return v;
}

constexpr unsigned NO_IDX = 0xffffffffu;

// k rounds over the C candidates each lane holds in registers.  emit(r, value, index) is called by every lane with the
// wave-uniform winner of round r (index NO_IDX once the candidates are exhausted).
template <int C, typename Emit>
__device__ __forceinline__ void wave_topk_rounds(float (&v)[C], unsigned (&id)[C], int k, Emit emit) {
  for (int r = 0; r < k; ++r) {
    float lm = v[0];
    unsigned li = id[0];
#pragma unroll
    for (int j = 1; j < C; ++j) {
      const bool better = v[j] > lm || (v[j] == lm && id[j] < li);
      lm = better ? v[j] : lm;
      li = better ? id[j] : li;
    }
    const float wm = wave_max(lm);
    const unsigned wi = wave_min_u32(lm == wm ? li : NO_IDX);
    emit(r, wm, wi);
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const bool hit = id[j] == wi && wi != NO_IDX;
      v[j] = hit ? -INFINITY : v[j];
      id[j] = hit ? NO_IDX : id[j];
    }
  }
}

// work[t] = penalised logit of every context token t (reads the untouched logits, so duplicates in ctx write the same value:
// the reference applies the penalty once per distinct id through a HashSet).
__global__ void repeat_penalty_kernel(const float* logits, float* work, const uint32_t* ctx, int n, float penalty, int V) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = ctx[i];
  if (t >= (uint32_t)V) return;  // `logits.get_mut(token_id)` is None: ignored
  const float x = logits[t];
  work[t] = x >= 0.f ? x / penalty : x * penalty;
}

constexpr int S1_C = 8;                      // elements per lane in stage 1
constexpr int S1_WAVE_ELEMS = 64 * S1_C;     // 512 logits per wave
constexpr int S1_WAVES_PER_BLOCK = 4;

// stage 1: every wave reduces 512 consecutive logits to its k best and its softmax partial (max, sum exp((x - max) * inv_temp))
__global__ __launch_bounds__(64 * S1_WAVES_PER_BLOCK) void topk_stage1_kernel(const float* x, int V, int k, float inv_temp,
                                                                              float* cand_val, unsigned* cand_idx,
                                                                              float* part_m, float* part_s) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * S1_WAVES_PER_BLOCK + (threadIdx.x >> 6);
  const int base = w * S1_WAVE_ELEMS;
  if (base >= V) return;
  float v[S1_C];
  unsigned id[S1_C];
  float lm = -INFINITY;
#pragma unroll
  for (int j = 0; j < S1_C; ++j) {
    const int i = base + j * 64 + lane;
    const bool ok = i < V;
    v[j] = ok ? x[ok ? i : 0] : -INFINITY;
    id[j] = ok ? (unsigned)i : NO_IDX;
    lm = fmaxf(lm, v[j]);
  }
  const float wm = wave_max(lm);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < S1_C; ++j)
    if (id[j] != NO_IDX) s += __expf((v[j] - wm) * inv_temp);
  s = wave_sum(s);
  if (lane == 0) {
    part_m[w] = wm;
    part_s[w] = s;
  }
  wave_topk_rounds<S1_C>(v, id, k, [&](int r, float val, unsigned idx) {
    if (lane == 0) {
      cand_val[(size_t)w * k + r] = val;
      cand_idx[(size_t)w * k + r] = idx;
    }
  });
}

constexpr int S2_WAVES = 16;
constexpr int S2_C = 20;  // 16 waves x 64 lanes x 20 >= 297 stage-1 waves x 64 candidates

// stage 2a (16 one-wave blocks, one per CU -- the rounds are VALU-bound, so the waves must not share a SIMD): each wave
// reduces a contiguous sixteenth of the stage-1 candidates to its k best
__global__ __launch_bounds__(64) void topk_stage2a_kernel(const float* cand_val, const unsigned* cand_idx, int n_cand, int k,
                                                          float* mid_val, unsigned* mid_idx) {
  const int lane = threadIdx.x, wave = blockIdx.x;
  const int chunk = (n_cand + S2_WAVES - 1) / S2_WAVES;
  const int c0 = wave * chunk, c1 = min(c0 + chunk, n_cand);
  float v[S2_C];
  unsigned id[S2_C];
#pragma unroll
  for (int j = 0; j < S2_C; ++j) {
    const int i = c0 + j * 64 + lane;
    const bool ok = i < c1;
    v[j] = ok ? cand_val[ok ? i : 0] : -INFINITY;
    id[j] = ok ? cand_idx[ok ? i : 0] : NO_IDX;
    if (id[j] == NO_IDX) v[j] = -INFINITY;
  }
  wave_topk_rounds<S2_C>(v, id, k, [&](int r, float val, unsigned idx) {
    if (lane == 0) {
      mid_val[wave * k + r] = val;
      mid_idx[wave * k + r] = idx;
    }
  });
}

// stage 2b (one wave): 16 x k -> k in (value desc, index asc) order, and the softmax partials -> (max, sumexp)
__global__ __launch_bounds__(64) void topk_stage2b_kernel(const float* mid_val, const unsigned* mid_idx, const float* part_m,
                                                          const float* part_s, int n_part, int k, float inv_temp,
                                                          float* out_val, unsigned* out_idx, float* out_ms) {
  const int lane = threadIdx.x;
  {
    float m = -INFINITY;
    for (int i = lane; i < n_part; i += 64) m = fmaxf(m, part_m[i]);
    const float M = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < n_part; i += 64) s += part_s[i] * __expf((part_m[i] - M) * inv_temp);
    s = wave_sum(s);
    if (lane == 0) {
      out_ms[0] = M;
      out_ms[1] = s;
    }
  }
  float v[S2_WAVES];
  unsigned id[S2_WAVES];
#pragma unroll
  for (int j = 0; j < S2_WAVES; ++j) {
    const int i = j * 64 + lane;
    const bool ok = i < S2_WAVES * k;
    v[j] = ok ? mid_val[ok ? i : 0] : -INFINITY;
    id[j] = ok ? mid_idx[ok ? i : 0] : NO_IDX;
    if (id[j] == NO_IDX) v[j] = -INFINITY;
  }
  wave_topk_rounds<S2_WAVES>(v, id, k, [&](int r, float val, unsigned idx) {
    if (lane == 0) {
      out_val[r] = val;
      out_idx[r] = idx;
    }
  });
}

}  // namespace

int sample_stage1_waves(int V) { return (V + S1_WAVE_ELEMS - 1) / S1_WAVE_ELEMS; }
// k <= 64 and the stage-1 candidates within what stage 2a holds in registers (16 waves x 64 lanes x 20)
bool sample_shape_ok(int V, int k) { return V > 0 && k >= 1 && k <= 64 && (int64_t)sample_stage1_waves(V) * k <= S2_WAVES * 64 * S2_C; }

void launch_repeat_penalty(const float* logits, float* work, const uint32_t* ctx, int n, float penalty, int V, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(repeat_penalty_kernel, dim3((n + 255) / 256), dim3(256), 0, st, logits, work, ctx, n, penalty, V);
}

void launch_topk_candidates(const float* x, int V, int k, float inv_temp, float* cand_val, unsigned* cand_idx, float* part_m,
                            float* part_s, float* out_val, unsigned* out_idx, float* out_ms, hipStream_t st) {
  const int nw = sample_stage1_waves(V);
  hipLaunchKernelGGL(topk_stage1_kernel, dim3((nw + S1_WAVES_PER_BLOCK - 1) / S1_WAVES_PER_BLOCK), dim3(64 * S1_WAVES_PER_BLOCK), 0,
                     st, x, V, k, inv_temp, cand_val, cand_idx, part_m, part_s);
  // the 16 x k intermediates live right behind the stage-1 candidates (model.hip sizes cand_* as (nw + 16) * 64)
  float* mid_val = cand_val + (size_t)nw * 64;
  unsigned* mid_idx = cand_idx + (size_t)nw * 64;
  hipLaunchKernelGGL(topk_stage2a_kernel, dim3(S2_WAVES), dim3(64), 0, st, cand_val, cand_idx, nw * k, k, mid_val, mid_idx);
  hipLaunchKernelGGL(topk_stage2b_kernel, dim3(1), dim3(64), 0, st, mid_val, mid_idx, part_m, part_s, nw, k, inv_temp, out_val,
                     out_idx, out_ms);
}

}  // namespace aha
