// Qwen3-VL specific host pieces: M-RoPE position bookkeeping (M2), the vision tower (V1-V7) and the visual-token
// scatter / DeepStack adds (M3).  SURVEY.md section 8a.
#pragma once
#include "model.h"

namespace aha {

int vision_create(aha_model* m, const aha_tensor_view* w, size_t nw);
void vision_destroy(aha_model* m);
// get_rope_index (/root/reference/src/models/qwen3vl/model.rs:901-1133): fills pos (3, n) rows T,H,W and sets
// m->rope_delta.  mm == nullptr => text only: rows = arange(n) + offset, delta 0.
int rope_index_core(const aha_model_desc& c, const uint32_t* ids, size_t S, const uint32_t* grid_thw, int n_images,
                    const uint32_t* video_grid_thw, int n_videos, int32_t* pos, int64_t* rope_delta);
int vl_rope_index(aha_model* m, const uint32_t* ids, size_t n, size_t offset, const aha_mm_input* mm, int32_t* pos);
int vision_forward_and_scatter(aha_model* m, const uint32_t* ids, size_t n, const aha_mm_input* mm, void* x);
int vision_deepstack_add(aha_model* m, int layer, void* x);
bool vision_has_deepstack(aha_model* m, int layer);   // vision_deepstack_add(layer) would change rows
int vision_debug_embeds(aha_model* m, int which, float* out, size_t n);
int vision_encode(aha_model* m, const aha_mm_input* mm, void* out_dev, int64_t* n_tokens);

}  // namespace aha
