// Host-side model object behind the C ABI (include/aha_hip.h).  Mirrors the reference's Qwen3Model
// (/root/reference/src/models/qwen3/model.rs:94-214) and its InferenceModel impl, re-designed for one MI355X:
// fused weight layouts, a paged KV cache that grows without reallocating, device-resident step state.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/aha_hip.h"
#include "kernels.h"

struct aha_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
};

namespace aha {

void set_error(const std::string& msg);
#define AHA_HIP_CHECK(expr)                                                                          \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      ::aha::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));                    \
      return AHA_ERR_HIP;                                                                            \
    }                                                                                                \
  } while (0)

struct LayerWeights {
  void* wqkv = nullptr;     // ((nh+2kvh)*d, H): q rows, then k rows, then v rows
  void* wo = nullptr;       // (H, nh*d)
  void* wgu = nullptr;      // (2I, H): 16-row blocks alternating gate / up (see kernels_gemm.hip ACT_SILU_MUL_PAIRS)
  void* wdown = nullptr;    // (H, I)
  void* in_norm = nullptr;  // (H)
  void* post_norm = nullptr;
  void* q_norm = nullptr;   // (d)
  void* k_norm = nullptr;
};

// device-resident scalars every decode kernel reads (so a step can be enqueued / replayed without host values)
constexpr uint32_t RING_CAP = 256;   // token ring of the device-resident greedy loop (>> its run-ahead)

struct StepState {
  uint32_t token;
  int32_t pos[3];     // rope position rows T,H,W of the current token
  int32_t kv_start;   // cache slot the current token is written to
  int32_t kv_len;     // valid cache tokens after the append
  uint32_t next_token;  // argmax of the last logits
  int32_t step;       // index into the device token log
};

struct ProfRec { int cls; hipEvent_t e0, e1; double bytes, flops; };

struct VisionModel;  // vision.h
struct AudioModel;   // audio.h

}  // namespace aha

// mmapped checkpoint directory (loader.hip)
struct aha_weights {
  struct Mapping { void* base; size_t len; };
  std::vector<Mapping> maps;
  std::vector<std::string> names;       // stable storage for aha_tensor_view::name
  std::vector<aha_tensor_view> views;
  ~aha_weights();
};

struct aha_model {
  aha_ctx* ctx = nullptr;
  aha_model_desc desc{};
  hipStream_t stream = nullptr;
  // weights
  void* embed = nullptr;
  void* lm_head = nullptr;  // == embed when tied
  void* final_norm = nullptr;
  std::vector<aha::LayerWeights> layers;
  std::vector<void*> owned;  // every hipMalloc'd block (weights, scratch), freed on destroy
  // rope constants
  float* d_inv_freq = nullptr;
  int32_t* d_axis_map = nullptr;
  float* d_rope = nullptr;       // (128) f32: the current decode step's rope cos/sin table (embed_state_kernel)
  float attn_scale = 0.f;
  // paged KV cache
  std::vector<void*> slabs;
  std::vector<uint64_t> h_page_ptrs;  // logical page -> byte address of its layer-0 storage
  uint64_t* d_page_ptrs = nullptr;
  size_t page_table_cap = 0;
  size_t pages_per_slab = 0;
  uint64_t page_bytes = 0;      // bytes of one page of one layer (K block + V block)
  uint64_t layer_stride = 0;    // bytes between layers inside a slab
  size_t n_pages = 0;           // pages currently mapped
  size_t cache_len = 0;         // tokens in the cache (== reference kv_cache.dim(2))
  bool scramble_pages = false;
  std::vector<uint64_t> free_pages;
  // step state
  aha::StepState* d_state = nullptr;
  aha::StepState* h_state = nullptr;  // pinned
  uint32_t* d_token_log = nullptr;
  size_t token_log_cap = 0;
  // device-resident greedy loop (model_decode_greedy): the generated tokens and the count of finished steps, written by the last
  // kernel of every step into PINNED host memory (system-scope stores) so the host can watch for a stop token while later steps
  // are still queued, without a stream synchronisation
  uint32_t* h_ring = nullptr;        // [RING_CAP] tokens, slot = step % RING_CAP   (host pointer)
  uint32_t* h_ring_dev = nullptr;    // the same buffer as the device sees it
  uint32_t* h_done = nullptr;        // steps finished since the loop started (host pointer; h_ring + RING_CAP)
  uint32_t* h_done_dev = nullptr;
  int64_t steps_executed = 0;        // decode steps that ran on the device in the last model_decode_greedy call (tests)
  bool rope_delta_valid = false;  // Qwen3VLModel::rope_deltas.is_some() (qwen3vl/model.rs:1229-1236)
  int64_t rope_delta = 0;  // Qwen3-VL: decode position = seqlen_offset + rope_delta (qwen3vl/model.rs:1235-1264)
  // decode scratch
  void *d_x = nullptr, *d_qkv = nullptr, *d_q = nullptr, *d_attn = nullptr, *d_act = nullptr, *d_hlast = nullptr;
  float* d_logits = nullptr;
  float* d_blk_max = nullptr;
  uint32_t* d_blk_idx = nullptr;
  float* d_part_o = nullptr;
  float* d_part_ml = nullptr;
  int max_nsplit = 64;
  // tensor parallelism (desc.tp_rank / tp_size); desc.num_attention_heads etc. hold the LOCAL (per-rank) sizes
  int tp_rank = 0, tp_size = 1;
  aha_allreduce_fn allreduce_cb = nullptr;
  void* allreduce_user = nullptr;
  aha_reduce_scatter_fn reduce_scatter_cb = nullptr;   // sequence-parallel prefill (aha_hip_set_seq_parallel)
  aha_all_gather_fn all_gather_cb = nullptr;
  void* sp_user = nullptr;
  void* rccl_comm = nullptr;
  void* rccl_comm_side = nullptr;   // a second communicator of the same ranks (ncclCommSplit) for the collectives enqueued on comm_stream
  // sequence-parallel prefill with the collectives of a row-parallel projection overlapped with its GEMM (model.hip
  // gemm_row_parallel): RCCL runs on its own high-priority stream, ordered against the compute stream by events
  hipStream_t comm_stream = nullptr;
  bool reserved_cus_set = false;   // this model holds a share of the reference-counted CU reservation (released with the communicator)
  hipEvent_t ev_gemm[8] = {};
  hipEvent_t ev_ag[8] = {};   // chunk i of a chunked all-gather has landed (norm_gather_gemm)
  hipEvent_t ev_comm = nullptr;
  int async_rc = 0;             // first error of an all-reduce issued from inside an enqueue helper
  int lm_rows = 0, lm_row0 = 0; // lm_head rows this rank streams (vocab-parallel under TP) and the first of them
  float* d_partial = nullptr;   // decode: (hidden) f32 partial projection; also the 2T-float argmax pair exchange
  float* p_partial = nullptr;   // prefill: (S, hidden) f32
  // Context-parallel prefill (aha_hip_set_context_parallel; model.hip cp_make_plan): every rank holds the FULL weights (tp_size 1) and
  // owns two page-aligned row chunks of the prompt; the only exchange is one all-gather of the layer's K / V pages per layer
  int cp_rank = 0, cp_size = 1;
  int comm_rank = 0;              // this rank's index in rccl_comm (tp_rank or cp_rank)
  aha_all_gather_fn cp_all_gather_cb = nullptr;   // host-callback seam (tests); RCCL otherwise
  void* cp_user = nullptr;
  void* p_cp_stage = nullptr;     // [rank][page slot][kv heads][K | V] of ONE layer, bf16
  // context-parallel prefill in flight: prompt row -> row of this rank's (compacted) activation buffers, or the scratch row behind them
  // for rows another rank owns; empty otherwise.  The vision tower's scatter / DeepStack adds go through it (vision_tower.hip).
  std::vector<int32_t> cp_row_map;
  size_t cp_stage_bytes = 0;
  void* p_hstage = nullptr;     // tensor-parallel prefill: staging of the chunked all-gather, [chunk][rank][rows] bf16 (norm_gather_gemm)
  unsigned head_ctr_base = 0;       // value every kv head's split-arrival counter has reached after all launches so far
  unsigned* d_bar = nullptr;        // split-arrival counters of the fused decode attention (kernels.h DECODE_SYNC_BYTES)
  unsigned long long* d_gemv_trace = nullptr;  // AHA_GEMV_TRACE timeline
  unsigned long long* d_attn_trace = nullptr;  // AHA_ATTN_TRACE timeline
  bool decode_fused = true;  // attention block of a decode step in one launch (kernels_attn.hip attn_decode_fused_kernel)
  float* h_logits = nullptr;  // pinned
  bool have_logits = false;   // d_logits holds the logits of a completed forward call
  bool logits_assembled = false;  // vocab-parallel TP: d_logits already all-reduced into the full vector
  // D11 candidate extraction (kernels_sample.hip), allocated on first use
  float* d_samp_work = nullptr;   // (V) penalised copy of the logits
  float* d_samp_f = nullptr;      // cand_val | part_m | part_s | out block {vals[64], max, sumexp, idx[64]}
  unsigned* d_samp_u = nullptr;   // cand_idx
  uint32_t* d_samp_ctx = nullptr;
  uint32_t* h_samp = nullptr;     // pinned: context ids (samp_ctx_cap) followed by the 130-word out block
  size_t samp_ctx_cap = 0;
  // prefill scratch (grown on demand)
  size_t pf_cap = 0;
  uint32_t* p_ids = nullptr;
  int32_t* p_pos = nullptr;
  void *p_x = nullptr, *p_h = nullptr, *p_qkv = nullptr, *p_q = nullptr, *p_attn = nullptr, *p_act = nullptr;
  void* p_rope = nullptr;       // (S, 128) bf16 cos | sin table of the prefill's positions (kernels.h launch_rope_table)
  void* p_gemm_ws = nullptr;    // f32 split-K slabs (kernels_gemm.hip)
  size_t gemm_ws_bytes = 0;
  void* d_sk_ctrs = nullptr;    // SK_MAX_COUNTERS zeroed u32: per-tile arrival counters of the persistent GEMM kernel (kernels_gemm_sk.hip)
  std::vector<void*> pf_owned;
  // vision tower (Qwen3-VL)
  aha::VisionModel* vision = nullptr;
  // audio tower (Qwen3-ASR)
  aha::AudioModel* audio = nullptr;
  // profiling
  bool profiling = false;
  std::vector<aha::ProfRec> prof;
  std::map<std::string, int> prof_cls;
  std::vector<std::string> prof_names;
  struct ProfAcc { double ms = 0, bytes = 0, flops = 0; int64_t n = 0; };
  std::vector<ProfAcc> prof_acc;
};

namespace aha {

// model.hip
int model_create(aha_ctx* ctx, const aha_model_desc* desc, const aha_tensor_view* w, size_t nw, aha_model** out);
void model_destroy(aha_model* m);
int model_embed(aha_model* m, const uint32_t* ids, size_t n, float* out);
int model_forward_initial(aha_model* m, const uint32_t* ids, size_t n, size_t offset, const aha_mm_input* mm,
                          float* logits_out, uint32_t* argmax_out);
int model_forward_step(aha_model* m, uint32_t token, size_t offset, float* logits_out, uint32_t* argmax_out);
int model_decode_greedy(aha_model* m, uint32_t first_token, size_t offset, size_t max_new, uint32_t* out);
int model_clear_cache(aha_model* m);
int model_last_logits(aha_model* m, float* logits_out);
int model_sample_candidates(aha_model* m, const uint32_t* ctx, size_t n_ctx, float repeat_penalty, float temperature, int k,
                            float* vals_out, uint32_t* idx_out, float* max_out, float* sumexp_out);
int model_ensure_pages(aha_model* m, size_t tokens);
KvLayer model_kv_layer(aha_model* m, int layer);
int model_debug_graph_step(aha_model* m, int replays, double* us_launches, double* us_graph);
int model_kv_export(aha_model* m, void* out_dev, size_t out_bytes, size_t* bytes_needed, size_t* n_tokens, int64_t* rope_delta);
int model_kv_import(aha_model* m, const void* in_dev, size_t in_bytes, int src_heads, int src_head0, int dst_head0, int n_heads, size_t n_tokens,
                    int64_t rope_delta);
int prof_collect(aha_model* m);
int model_allreduce(aha_model* m, float* buf, size_t count);
// loader.hip
int config_parse(const char* dir, aha_model_desc* out);
int config_torch_dtype(const char* dir, std::string* out);
int weights_open(const char* dir, aha_weights** out);
int model_load(aha_ctx* ctx, const char* dir, size_t kv_reserve_tokens, aha_model** out);
int rccl_allreduce(aha_model* m, float* buf, size_t count);  // tp_rccl.hip
int rccl_reduce_scatter(aha_model* m, float* buf, size_t count_per_rank, hipStream_t st = nullptr);   // in place: rank r keeps slice r; st: stream (default: the model's)
int rccl_all_gather(aha_model* m, void* buf, size_t bytes_per_rank, hipStream_t st = nullptr);   // in place: slice r is rank r's contribution
int tp_unique_id(void* out128);
int tp_init_rccl(aha_model* m, const void* id128);
int cp_init_rccl(aha_model* m, const void* id128);
int debug_cp_plan(int S, int world, int rank, int* out5);
void tp_destroy(aha_model* m);

// helpers shared with vision.hip
const aha_tensor_view* find_tensor(const aha_tensor_view* w, size_t nw, const std::string& name);
int upload_tensor(aha_model* m, const aha_tensor_view* t, const std::vector<int64_t>& shape, void** out,
                  int64_t pad_rows_to = 0, int64_t pad_cols_to = 0);
int dev_alloc(aha_model* m, size_t bytes, void** out, bool zero = false);

struct ProfScope {
  aha_model* m;
  int idx = -1;
  ProfScope(aha_model* m, const char* cls, double bytes, double flops);
  ~ProfScope();
};

}  // namespace aha
