// Qwen3-VL vision tower (V1-V7) -- placeholder until the ViT kernels land: text-only Qwen3-VL works, images are refused.
#include "vision.h"

namespace aha {

struct VisionModel {};

int vision_create(aha_model* m, const aha_tensor_view* w, size_t nw) {
  (void)w;
  (void)nw;
  m->vision = nullptr;
  return AHA_OK;
}
void vision_destroy(aha_model* m) {
  delete m->vision;
  m->vision = nullptr;
}
int vision_forward_and_scatter(aha_model* m, const uint32_t*, size_t, const aha_mm_input*, void*) {
  (void)m;
  set_error("vision tower not built");
  return AHA_ERR_UNSUPPORTED;
}
int vision_deepstack_add(aha_model*, int, void*) { return AHA_OK; }

}  // namespace aha
