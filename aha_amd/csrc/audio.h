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

}  // namespace aha
