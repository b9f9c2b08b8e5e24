// Qwen3-ASR host pieces (audio_tower.hip).  SURVEY.md section 8a A0-A3.
#pragma once
#include "model.h"

namespace aha {

int audio_create(aha_model* m, const aha_tensor_view* w, size_t nw);
void audio_destroy(aha_model* m);
// log-mel (if raw samples are given) -> conv stack -> encoder -> projector -> scatter into the <|audio_pad|> rows of x
int audio_forward_and_scatter(aha_model* m, const uint32_t* ids, size_t n, const aha_mm_input* mm, void* x);
int audio_debug_embeds(aha_model* m, float* out, size_t n);
int logmel_standalone(const float* d_samples, int64_t n_samples, float* d_out, hipStream_t st);
// audio_pre.hip: resample_audio_from_vec_f32 (audio_utils.rs:590-616)
int64_t debug_resample_taps(int64_t orig, int64_t new_f, float* taps, int64_t cap, int32_t* width, int32_t* klen);
int64_t resample_output_len(int64_t length, int64_t orig_sr, int64_t target_sr);
int64_t audio_resample(aha_ctx* ctx, const float* pcm, int64_t n_frames, int channels, int orig_sr, int target_sr, float* out,
                       int64_t out_cap);

}  // namespace aha
