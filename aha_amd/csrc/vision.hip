// Qwen3-VL host pieces.  (Vision tower kernels: see vision_tower.hip once present.)
#include "vision.h"

#include <algorithm>

#include "common.h"

namespace aha {

// get_rope_index, /root/reference/src/models/qwen3vl/model.rs:901-1133 (B = 1, no attention mask).
// Reference call rule (model.rs:1229-1264): the FIRST forward after clear_cache (rope_deltas == None) computes
// positions from the ids alone (seqlen_offset is ignored) and stores rope_delta = max_pos + 1 - S; every later call
// uses arange(S) + seqlen_offset + rope_delta on all three rows.
int vl_rope_index(aha_model* m, const uint32_t* ids, size_t n, size_t offset, const aha_mm_input* mm, int32_t* pos) {
  const aha_model_desc& c = m->desc;
  const size_t S = n;
  if (m->rope_delta_valid) {
    for (int a = 0; a < 3; ++a)
      for (size_t i = 0; i < S; ++i) pos[a * S + i] = (int32_t)((int64_t)i + (int64_t)offset + m->rope_delta);
    return AHA_OK;
  }
  if (!mm || (mm->n_images <= 0 && mm->n_videos <= 0)) {
    for (int a = 0; a < 3; ++a)
      for (size_t i = 0; i < S; ++i) pos[a * S + i] = (int32_t)i;
    m->rope_delta = 0;
    m->rope_delta_valid = true;
    return AHA_OK;
  }
  int64_t delta = 0;
  const int rc = rope_index_core(c, ids, n, mm->image_grid_thw, mm->n_images, mm->video_grid_thw, mm->n_videos, pos, &delta);
  if (rc) return rc;
  m->rope_delta = delta;
  m->rope_delta_valid = true;
  return AHA_OK;
}

// The index arithmetic of get_rope_index alone (host only, no model state): positions (3, S) and rope_delta.
int rope_index_core(const aha_model_desc& c, const uint32_t* ids, size_t S, const uint32_t* grid_thw, int n_images,
                    const uint32_t* video_grid_thw, int n_videos, int32_t* pos, int64_t* rope_delta) {
  const int merge = c.vis_spatial_merge_size;
  size_t text_start = 0, out = 0;
  int64_t max_pos = -1;  // max over the previous block (llm_pos_ids_list.last().max_all())
  int image_index = 0;
  // model.rs:908-925: a (t, h, w) video grid becomes t rows (1, h, w), consumed one per <|vision_start|><|video_pad|> run
  int video_index = 0, video_frame = 0;
  bool any_block = false;
  for (size_t j = 0; j + 1 < S; ++j) {
    if (ids[j] != (uint32_t)c.vision_start_token_id) continue;
    const size_t e = j + 1;  // index of the first token after <|vision_start|>
    uint32_t frame_thw[3];
    const uint32_t* thw;
    if (ids[e] == (uint32_t)c.image_token_id) {
      if (image_index >= n_images) {
        set_error("get_rope_index: more <|vision_start|><|image_pad|> runs than images");
        return AHA_ERR_SHAPE;
      }
      thw = grid_thw + 3 * (size_t)image_index++;
    } else if (ids[e] == (uint32_t)c.video_token_id) {
      while (video_index < n_videos && (uint32_t)video_frame >= video_grid_thw[3 * (size_t)video_index]) {
        ++video_index;
        video_frame = 0;
      }
      if (video_index >= n_videos) {
        set_error("get_rope_index: more <|vision_start|><|video_pad|> runs than video frames");
        return AHA_ERR_SHAPE;
      }
      frame_thw[0] = 1;
      frame_thw[1] = video_grid_thw[3 * (size_t)video_index + 1];
      frame_thw[2] = video_grid_thw[3 * (size_t)video_index + 2];
      ++video_frame;
      thw = frame_thw;
    } else {
      // neither an image nor a video token after <|vision_start|>: the reference would reuse the previous grid (or index an
      // empty one and panic, model.rs:966-984); a prompt the processor wrote never has this, and the run is skipped here
      continue;
    }
    const int64_t t = thw[0], gh = thw[1] / merge, gw = thw[2] / merge;
    if (e < text_start) {
      set_error("get_rope_index: overlapping vision runs");
      return AHA_ERR_SHAPE;
    }
    const int64_t text_len = (int64_t)e - (int64_t)text_start;
    const int64_t start = any_block ? max_pos + 1 : 0;
    for (int64_t i = 0; i < text_len; ++i)
      for (int a = 0; a < 3; ++a) pos[a * S + out + i] = (int32_t)(start + i);
    out += text_len;
    if (text_len > 0) max_pos = start + text_len - 1;
    any_block = true;  // the (possibly empty) text block was pushed
    const int64_t base = start + text_len;
    if (out + (size_t)(t * gh * gw) > S) {
      set_error("get_rope_index: image grid longer than the remaining sequence");
      return AHA_ERR_SHAPE;
    }
    for (int64_t ti = 0; ti < t; ++ti)
      for (int64_t hi = 0; hi < gh; ++hi)
        for (int64_t wi = 0; wi < gw; ++wi) {
          const size_t o = out + (size_t)((ti * gh + hi) * gw + wi);
          pos[0 * S + o] = (int32_t)(base + ti);
          pos[1 * S + o] = (int32_t)(base + hi);
          pos[2 * S + o] = (int32_t)(base + wi);
        }
    out += (size_t)(t * gh * gw);
    max_pos = base + std::max(t, std::max(gh, gw)) - 1;  // max over the (3, t*gh*gw) vision block just pushed
    text_start = e + (size_t)(t * gh * gw);
  }
  if (text_start < S) {
    const int64_t start = any_block ? max_pos + 1 : 0;
    const int64_t text_len = (int64_t)S - (int64_t)text_start;
    for (int64_t i = 0; i < text_len; ++i)
      for (int a = 0; a < 3; ++a) pos[a * S + out + i] = (int32_t)(start + i);
    out += text_len;
  }
  if (out != S) {
    set_error("get_rope_index: position list length " + std::to_string(out) + " != sequence length " + std::to_string(S));
    return AHA_ERR_SHAPE;
  }
  int64_t mx = 0;
  for (size_t i = 0; i < 3 * S; ++i) mx = std::max<int64_t>(mx, pos[i]);
  *rope_delta = mx + 1 - (int64_t)S;
  return AHA_OK;
}

}  // namespace aha
