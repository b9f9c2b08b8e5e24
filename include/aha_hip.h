/* aha_hip.h -- C ABI of the MI355X-native backend for aha's Qwen3 / Qwen3-VL hot path.
 *
 * The reference (jhqxxx/aha v0.2.6) has no FFI or backend trait; its only seam for this path is the Rust trait
 *   InferenceModel { forward_initial, forward_step, clear_cache, stop_token_ids }
 *     -- /root/reference/src/models/common/mod.rs:25-45
 * driven by generate_generic -- /root/reference/src/models/common/generate.rs:115-159.
 * The model-level entry points below are exactly what a Rust `impl InferenceModel for HipQwen3` would bind
 * (INTEGRATION.md shows the shim).  The op-level entry points exist for unit parity tests of each kernel against
 * the oracle; they take DEVICE pointers and a hipStream_t (passed as void*).
 *
 * Conventions: every function returns 0 on success or a negative aha_status; the message of the last error on the
 * calling thread is available from aha_hip_last_error().  The library never aborts and never throws across the ABI.
 * Handles are NOT thread-safe (same contract as the reference: `&mut self`, server holds a write lock per request,
 * /root/reference/src/server/api.rs:117).  Host buffers passed in stay owned by the caller; weights are copied to HBM.
 */
#ifndef AHA_HIP_H
#define AHA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aha_ctx aha_ctx;
typedef struct aha_model aha_model;

enum aha_status {
  AHA_OK = 0,
  AHA_ERR_INVALID = -1,        /* bad argument / null handle */
  AHA_ERR_HIP = -2,            /* a HIP runtime call failed */
  AHA_ERR_OOM = -3,
  AHA_ERR_SHAPE = -4,          /* tensor shape does not match the model description */
  AHA_ERR_MISSING_WEIGHT = -5, /* a tensor name the reference looks up is absent */
  AHA_ERR_UNSUPPORTED = -6,
  AHA_ERR_STATE = -7           /* e.g. seqlen_offset does not equal the current cache length */
};

enum aha_dtype { AHA_BF16 = 0, AHA_F16 = 1, AHA_F32 = 2, AHA_U32 = 3, AHA_U8 = 4 };
enum aha_arch { AHA_ARCH_QWEN3 = 0, AHA_ARCH_QWEN3VL = 1, AHA_ARCH_QWEN3ASR = 2 };

/* Mirrors Qwen3Config (/root/reference/src/models/qwen3/config.rs:4-27), Qwen3VLTextConfig / Qwen3VLVisionConfig /
 * Qwen3VLConfig (/root/reference/src/models/qwen3vl/config.rs:59-133).  Vision fields are ignored for AHA_ARCH_QWEN3. */
typedef struct aha_model_desc {
  int32_t arch;
  int32_t hidden_size, intermediate_size, num_hidden_layers;
  int32_t num_attention_heads, num_key_value_heads, head_dim, vocab_size;
  float rms_norm_eps, rope_theta;
  int32_t tie_word_embeddings;
  int32_t mrope_section[3];        /* {0,0,0} => plain 1-D RoPE (rope.rs:583-612); else interleaved M-RoPE (rope.rs:454-476) */
  /* vision tower (Qwen3VLVisionConfig) */
  int32_t vis_depth, vis_hidden_size, vis_num_heads, vis_intermediate_size, vis_in_channels, vis_patch_size,
      vis_temporal_patch_size, vis_spatial_merge_size, vis_out_hidden_size, vis_num_position_embeddings;
  int32_t vis_deepstack_indexes[8];
  int32_t vis_num_deepstack;
  /* token ids (Qwen3VLConfig) */
  int32_t image_token_id, video_token_id, vision_start_token_id, vision_end_token_id;
  /* KV-cache sizing hint (tokens).  The cache is paged and grows on demand; this only pre-reserves pages. */
  int32_t kv_reserve_tokens;
  int32_t n_stop_tokens;
  uint32_t stop_tokens[8];         /* generation_config.json eos_token_id list (qwen3/generate.rs:36-43) */
  /* audio tower (Qwen3ASRAudioConfig, /root/reference/src/models/qwen3_asr/config.rs:24-96); AHA_ARCH_QWEN3ASR only */
  int32_t aud_d_model, aud_encoder_layers, aud_attention_heads, aud_ffn_dim, aud_num_mel_bins,
      aud_downsample_hidden_size, aud_output_dim, aud_n_window;
  int32_t audio_token_id;
  /* Tensor parallelism of the decoder stack (the reference has none): rank tp_rank of tp_size holds
   * num_attention_heads/tp_size q heads, num_key_value_heads/tp_size kv heads (+ their KV cache) and
   * intermediate_size/tp_size MLP columns; o_proj / down_proj produce f32 partial sums that are all-reduced (RCCL or the
   * host callback below) before the residual add.  tp_size 0 or 1 = off.  Every rank passes the FULL checkpoint tensors. */
  int32_t tp_rank, tp_size;
  /* Compute dtype the caller asks for (the `dtype: Option<DType>` of XxxGenerateModel::init resolved by get_dtype,
   * /root/reference/src/utils/mod.rs:77-115; see aha_hip_get_dtype).  AHA_BF16 (= 0, the zero-initialised default) is the
   * only dtype the kernels compute in: f16 / f32 CHECKPOINTS are accepted and cast to bf16 at load, an f16 / f32 COMPUTE
   * request is refused by aha_hip_model_create with AHA_ERR_UNSUPPORTED instead of silently running in bf16. */
  int32_t compute_dtype;
} aha_model_desc;

/* One checkpoint tensor: HF name, pointer (host memory, e.g. an mmapped safetensors file; or, when on_device != 0,
 * device memory of the same GPU -- bf16 only), dtype, shape. */
typedef struct aha_tensor_view {
  const char* name;
  const void* data;
  int32_t dtype;
  int32_t ndim;
  int64_t shape[5];
  int32_t on_device;
} aha_tensor_view;

/* MultiModalData for Qwen3-VL (/root/reference/src/models/qwen3vl/generate.rs:79-101: data_vec =
 * [pixel_values, image_grid_thw, None, None, cache_position]).  pixel_values is the processor output
 * (N_patches, C*T*P*P) in merge-window row order (/root/reference/src/models/qwen3vl/processor.rs:174-227). */
/* (Layout history: the four video fields at the end were added in library version 0.2 -- aha_hip_version(); a caller built
 * against the 0.1 header must be recompiled, the library reads all fields.  Zero-initialise the struct and set what applies.) */
typedef struct aha_mm_input {
  const void* pixel_values;     /* host, (n_patches, patch_dim) */
  int32_t pixel_dtype;          /* AHA_BF16 or AHA_F32 */
  int64_t n_patches;
  const uint32_t* image_grid_thw; /* host, (n_images, 3) */
  int32_t n_images;
  /* Qwen3-ASR (MultiModalData = [input_features], /root/reference/src/models/qwen3_asr/generate.rs:100-125): either the
   * Whisper log-mel features (num_mel_bins, n_frames) f32, or raw 16 kHz mono samples from which the library computes
   * them on the GPU (WhisperFeatureExtractor, feature_extraction_whisper.rs:93-115).  Host pointers. */
  const float* audio_features;
  int64_t n_frames;
  const float* audio_samples;
  int64_t n_samples;
  /* Qwen3-VL, image-parallel ViT: embeddings already computed (by aha_hip_vision_encode, possibly on
   * other GPUs and all-gathered): device pointer to (1 + n_deepstack, n_image_tokens, hidden) bf16.  When set,
   * pixel_values is ignored (image_grid_thw is still needed for the M-RoPE positions). */
  const void* image_embeds;
  int64_t n_image_tokens;
  /* Qwen3-VL video input (data_vec[2..3] = pixel_values_video, video_grid_thw, /root/reference/src/models/qwen3vl/model.rs:
   * 1169-1187,1299-1307): patch rows of the sampled frames in processor order (process_videos, processor.rs:253-281) and one
   * (t, h, w) grid per video, t = temporal patches (frame pairs).  Host pointers; same dtype as pixel_values.  Frame decoding
   * and the swscale resize (get_video_data, processor.rs:447-571, ffmpeg) stay on the caller's side.  With image_embeds set,
   * the precomputed rows are the images' tokens followed by the videos' tokens. */
  const void* pixel_values_video;
  int64_t n_patches_video;
  const uint32_t* video_grid_thw; /* host, (n_videos, 3) */
  int32_t n_videos;
} aha_mm_input;

/* ---- lifecycle ---------------------------------------------------------------------------------------------- */
int aha_hip_init(int device, aha_ctx** out);
void aha_hip_shutdown(aha_ctx* ctx);
const char* aha_hip_last_error(void);
const char* aha_hip_version(void);

/* get_dtype (/root/reference/src/utils/mod.rs:77-115) as a `hip` cargo feature would extend it: an explicit request wins
 * (requested = an aha_dtype, or -1 for None); otherwise the checkpoint's config.json "torch_dtype" string decides --
 * "float32"/"float" -> AHA_F32, "float16" -> AHA_F16, "bfloat16" -> AHA_BF16 (gfx950 has native bf16: the role the
 * SM >= 8.0 test plays on the reference's cuda branch; its cpu branch maps bfloat16 to F16, mod.rs:107), anything else ->
 * AHA_F32 like the reference's `_` arm.  Host-only.  aha_hip_check_dtype says whether this library can COMPUTE in a dtype:
 * AHA_OK for AHA_BF16, AHA_ERR_UNSUPPORTED (with a message) for the others. */
int aha_hip_get_dtype(int32_t requested, const char* cfg_dtype, int32_t* out);
int aha_hip_check_dtype(int32_t dtype);

/* Replaces XxxGenerateModel::init's VarBuilder::from_mmaped_safetensors + Qwen3Model::new
 * (/root/reference/src/models/qwen3/generate.rs:22-50, qwen3/model.rs:104-134). */
int aha_hip_model_create(aha_ctx* ctx, const aha_model_desc* desc, const aha_tensor_view* weights, size_t n_weights,
                         aha_model** out);
void aha_hip_model_destroy(aha_model* m);

/* Host-only: Qwen3VLModel::get_rope_index (/root/reference/src/models/qwen3vl/model.rs:901-1133) for images: the (3, n_ids)
 * M-RoPE position rows T, H, W and rope_delta = max(position) + 1 - n_ids.  Only the token ids and vis_spatial_merge_size
 * of `desc` are read.  forward_initial runs the same code on its own input; this entry point exists so the index
 * arithmetic can be tested (and reused by a host) without a GPU. */
int aha_hip_get_rope_index(const aha_model_desc* desc, const uint32_t* input_ids, size_t n_ids, const uint32_t* image_grid_thw,
                           int32_t n_images, int32_t* pos_out, int64_t* rope_delta_out);
/* The same with videos (model.rs:908-925,973-981): every (t, h, w) video grid stands for t frames of (1, h, w), one per
 * <|vision_start|><|video_pad|> run (the processor writes a timestamp and one such run per temporal patch, processor.rs:397-426). */
int aha_hip_get_rope_index_mm(const aha_model_desc* desc, const uint32_t* input_ids, size_t n_ids, const uint32_t* image_grid_thw,
                              int32_t n_images, const uint32_t* video_grid_thw, int32_t n_videos, int32_t* pos_out,
                              int64_t* rope_delta_out);

/* ---- Qwen3-Embedding / Qwen3-Reranker ------------------------------------------------
 * == Qwen3Embedding::embed_one after tokenisation (/root/reference/src/models/qwen3_embedding/mod.rs:50-64):
 * forward_hidden(input_ids, offset 0) -> last position after the final RMSNorm -> f32 -> l2_normalize
 * (common/modules.rs:1287-1294) -> out[hidden_size]; the KV cache is cleared before and after, as the reference does.
 * Qwen3Reranker::rerank (qwen3_reranker/mod.rs:23-31) is the dot product of two such vectors
 * (cosine_similarity_no_l2, modules.rs:1381-1389) -- host arithmetic on the caller's side. */
int aha_hip_embed(aha_model* m, const uint32_t* input_ids, size_t n_ids, float* out);

/* ---- checkpoint directory -> model (XxxGenerateModel::init minus tokenizer / chat template) --------------------------
 * aha_hip_config_parse: <dir>/config.json -> aha_model_desc, the same field mapping serde does into Qwen3Config
 *   (/root/reference/src/models/qwen3/config.rs:4-27), Qwen3VLConfig (qwen3vl/config.rs:51-133, text_config / vision_config,
 *   top-level tie_word_embeddings qwen3vl/model.rs:853) or Qwen3ASRConfig (qwen3_asr/config.rs:6-22, thinker_config.*);
 *   stop tokens from <dir>/generation_config.json eos_token_id (qwen3/generate.rs:33-36; a scalar or a list).  The
 *   architecture is taken from "model_type" / the presence of vision_config / thinker_config.  Host only (no GPU).
 * aha_hip_weights_open: mmaps every *.safetensors file in <dir> (find_type_files, utils/mod.rs:121-137 +
 *   VarBuilder::from_mmaped_safetensors, qwen3/generate.rs:30-31) and indexes the tensors; views point into the mappings
 *   and stay valid until aha_hip_weights_close.  Host only.
 * aha_hip_model_load = config_parse + weights_open + aha_hip_model_create + weights_close. */
typedef struct aha_weights aha_weights;
int aha_hip_config_parse(const char* model_dir, aha_model_desc* out);
/* The dtype string the reference's init passes to get_dtype for this checkpoint: Qwen3 config.json "torch_dtype"
 * (qwen3/config.rs:23, qwen3/generate.rs:28), Qwen3-VL "text_config.dtype" (qwen3vl/config.rs:100), Qwen3-ASR the constant
 * "bfloat16" (qwen3_asr/config.rs:186).  NUL-terminated into out (cap bytes).  A host calls
 * aha_hip_get_dtype(requested, <this>, &d) and aha_hip_check_dtype(d) before aha_hip_model_load, so that an f16 / f32
 * checkpoint without an explicit dtype is refused the way the header promises instead of silently computing in bf16. */
int aha_hip_config_torch_dtype(const char* model_dir, char* out, size_t cap);
int aha_hip_weights_open(const char* model_dir, aha_weights** out);
size_t aha_hip_weights_count(const aha_weights* w);
int aha_hip_weights_get(const aha_weights* w, size_t index, aha_tensor_view* out);
void aha_hip_weights_close(aha_weights* w);
int aha_hip_model_load(aha_ctx* ctx, const char* model_dir, size_t kv_reserve_tokens, aha_model** out);

/* ---- InferenceModel (common/mod.rs:25-45) ------------------------------------------------------------------- */
/* forward_initial(&mut self, input_ids, seqlen_offset, data) -> logits (1,1,V).
 * logits_out (V floats, host, may be NULL) receives the last position's logits as f32 exactly as the generic loop
 * reads them (generate.rs:75).  argmax_out (may be NULL) receives the first maximal index (Sampling::ArgMax). */
int aha_hip_forward_initial(aha_model* m, const uint32_t* input_ids, size_t n_ids, size_t seqlen_offset,
                            const aha_mm_input* mm, float* logits_out, uint32_t* argmax_out);
/* forward_step(&mut self, input_ids (1,1), seqlen_offset) -> logits (1,1,V). */
int aha_hip_forward_step(aha_model* m, uint32_t token, size_t seqlen_offset, float* logits_out, uint32_t* argmax_out);
/* clear_cache(&mut self) */
int aha_hip_clear_cache(aha_model* m);
/* stop_token_ids(&self) -> Vec<u32>: writes up to cap ids, returns the count (>=0) or a negative status. */
int aha_hip_stop_token_ids(const aha_model* m, uint32_t* out, size_t cap);

/* Extension (not in the reference): device-resident greedy loop = generate_generic with temperature 0
 * (generate.rs:115-159) without a host round trip per token.  Must follow a forward_initial/forward_step call;
 * first_token is the token sampled from that call.  Writes up to max_new tokens; stops after an eos id.
 * Returns the number of tokens written or a negative status. */
int aha_hip_decode_greedy(aha_model* m, uint32_t first_token, size_t seqlen_offset, size_t max_new, uint32_t* tokens_out);

/* D11, device half of sample_and_push (common/generate.rs:70-86) for the non-greedy samplers of get_logit_processor
 * (common/sample.rs:7-38: candle LogitsProcessor TopK / TopKThenTopP / TopP).  Works on the logits the last
 * forward_initial / forward_step / decode_greedy call left on the device:
 *   1. use_repeat_penalty (sample.rs:41-60 -> candle_transformers::utils::apply_repeat_penalty): every DISTINCT id of
 *      `context` (the caller passes the last repeat_last_n generated ids, sample.rs:49-53) has its logit divided by
 *      repeat_penalty if >= 0, multiplied otherwise; repeat_penalty == 1 or n_context == 0 leaves the logits unchanged;
 *   2. the k (1..64) largest penalised logits, ordered by (value descending, index ascending), into vals_out / idx_out;
 *   3. max_out / sumexp_out = max_i x_i and sum_i exp((x_i - max) / temperature) over the WHOLE vocabulary, so that
 *      exp((vals_out[j] - max) / temperature) / sumexp equals the probability candle's softmax(logits / temperature)
 *      assigns to candidate j (temperature <= 0 is treated as 1).
 * The host finishes with the top-p cut over the candidates and the weighted draw (aha_hip_rng_* below): 8k + 8 bytes leave the device per
 * token instead of the V-float logits vector.  Not in the reference as a function: it replaces the body of
 * LogitsProcessor::sample up to the random draw.  If k exceeds the vocabulary the surplus entries come back as value -inf /
 * index 0xFFFFFFFF.  Under tensor parallelism (vocab-parallel lm_head) the call is collective: every rank makes it, the
 * shards of the logits are all-reduced first. */
int aha_hip_sample_candidates(aha_model* m, const uint32_t* context, size_t n_context, float repeat_penalty, float temperature,
                              int32_t k, float* vals_out, uint32_t* idx_out, float* max_out, float* sumexp_out);

/* The V f32 logits of the last forward call, exactly what logits_out of that call would have received (fallback of the
 * candidate path: Sampling::All, or a TopP whose nucleus is wider than 64 tokens). */
int aha_hip_last_logits(aha_model* m, float* logits_out);

/* The random draw of candle's LogitsProcessor (the `rng` field and sample_multinomial of candle-transformers 0.9.2, built by
 * get_logit_processor, /root/reference/src/models/common/sample.rs:7-37 with seed 299792458, common/generate.rs:408,452, or
 * 34562, qwen3_asr/generate.rs:134), host code:
 *   aha_hip_rng_create(seed)        = rand 0.9.2 StdRng::seed_from_u64(seed)   (candle-transformers' own rand, Cargo.lock:590-606):
 *                                     PCG32 expansion of the u64 to 32 seed bytes, ChaCha12 stream, block counter 0, stream id 0
 *   aha_hip_rng_next_u32            = RngCore::next_u32 on it
 *   aha_hip_rng_weighted_index      = WeightedIndex::<f32>::new(weights)?.sample(&mut rng): ONE next_u32; index of the first running
 *                                     f32 sum greater than the uniform draw in [0, total).  AHA_ERR_INVALID for a negative / NaN
 *                                     weight or an all-zero vector (the crate's Err(..)), with the RNG left untouched.
 * [unverified] against the crates themselves (not on disk, no cargo): restated from their published algorithms, see
 * csrc/sampler_rng.hip.  For Sampling::TopK / TopKThenTopP candle draws over the k probabilities in the order
 * select_nth_unstable_by leaves them, which Rust does not specify: callers pass them ranked (probability desc, logit desc,
 * index asc), the order aha_hip_sample_candidates returns. */
typedef struct aha_rng aha_rng;
int aha_hip_rng_create(uint64_t seed, aha_rng** out);
void aha_hip_rng_destroy(aha_rng* rng);
uint32_t aha_hip_rng_next_u32(aha_rng* rng);
int aha_hip_rng_weighted_index(aha_rng* rng, const float* weights, size_t n, uint32_t* index_out);
/* Test hook: the ChaCha block function (even `rounds`) on a 16-word state -- RFC 7539 section 2.3.2 pins it at 20 rounds. */
int aha_hip_debug_chacha_block(const uint32_t* state16, int rounds, uint32_t* out16);

/* Extension for the image-parallel ViT (each GPU encodes its share of the images, embeddings are all-gathered over RCCL):
 * runs only the vision tower on `mm` (its images, then its videos) and writes (1 + n_deepstack, n_tokens, hidden) bf16 to out_dev
 * (device memory, may be NULL to query n_tokens); rows = the images' tokens followed by the videos' tokens. */
int aha_hip_vision_encode(aha_model* m, const aha_mm_input* mm, void* out_dev, int64_t* n_tokens);

/* Tensor-parallel seam.  Either (a) a host callback that must leave buf = sum over ranks of buf (count f32, device memory)
 * before it returns -- e.g. torch.distributed.all_reduce over RCCL on a tensor view of buf -- or (b) an RCCL communicator
 * owned by the library: every rank calls aha_hip_tp_unique_id on rank 0's bytes (128) and aha_hip_tp_init_rccl.  With (b)
 * the all-reduce is enqueued on the model's stream (no host synchronisation). */
typedef int (*aha_allreduce_fn)(void* buf_f32_dev, size_t count, void* user);
int aha_hip_set_allreduce(aha_model* m, aha_allreduce_fn fn, void* user);
int aha_hip_tp_unique_id(void* out128);
int aha_hip_tp_init_rccl(aha_model* m, const void* unique_id128);
/* Sequence-parallel prefill ("reduce-scatter + all-gather, sequence-parallel norms").  With it a
 * tensor-parallel prefill keeps the residual stream row-sharded (rank r owns rows [r*ceil(S/T), ...)): the f32 partial sums of
 * o_proj / down_proj are REDUCE-SCATTERED over rows (same f32 sums as the all-reduce: identical numerics), the residual add and
 * the next RMSNorm run on the owned rows only, and the normalised bf16 rows are ALL-GATHERED for the next column-parallel
 * GEMM -- (T-1)/T * (4 + 2) bytes per element and rank instead of 2 * (T-1)/T * 4 for the ring all-reduce.
 * Used automatically when the library owns an RCCL communicator (aha_hip_tp_init_rccl; the collectives it overlaps with GEMMs on its
 * communication stream run on a second communicator of the same ranks, split off at init -- AHA_TP_SIDE_COMM=0 shares the first one);
 * with the host-callback seam install:
 *   reduce_scatter(buf, count_per_rank): buf holds T * count_per_rank f32 on the device; on return rank r's slice
 *       buf[r*count_per_rank ..) must hold the sum over ranks of that slice (the other slices are undefined);
 *   all_gather(buf, bytes_per_rank): rank r's slice of buf (T * bytes_per_rank bytes) is its contribution; on return every
 *       slice must be filled on every rank.
 * Passing NULLs removes them (all-reduce path).  AHA_TP_SP=0 in the environment forces the all-reduce path. */
typedef int (*aha_reduce_scatter_fn)(void* buf_f32_dev, size_t count_per_rank, void* user);
typedef int (*aha_all_gather_fn)(void* buf_dev, size_t bytes_per_rank, void* user);
int aha_hip_set_seq_parallel(aha_model* m, aha_reduce_scatter_fn reduce_scatter, aha_all_gather_fn all_gather, void* user);
/* Context-parallel prefill (round 4; the 288-GB design of SURVEY.md section 8e row 3 "long-context text prefill": an 8B checkpoint is
 * 16 GB, so every GPU holds the FULL weights -- create the model with tp_size 1 on every rank -- and the PROMPT is sharded instead).
 * The prompt's 64-token KV pages are cut into 2 * world contiguous chunks; rank r owns chunks r and 2 * world - 1 - r (balanced causal
 * work) and runs every GEMM / norm / rope of the prefill on its own rows only; per layer the ranks all-gather that layer's K / V pages
 * (the one op that couples rows is attention).  After the call EVERY rank holds the complete KV cache and the last position's logits
 * (rank 0 owns the last row; it is broadcast), so decode continues on any rank -- rank 0 by convention -- with no hand-back.
 * Inbound bytes per rank and layer at 41 k tokens on 8 GPUs: 147 MB, against 1.76 GB for the tensor-parallel form above.
 * Every rank calls aha_hip_forward_initial with the same ids (offset 0: a fresh cache; other calls run unsharded on every rank).
 * Collective: an RCCL communicator owned by the library (aha_hip_tp_unique_id on rank 0, then aha_hip_cp_init_rccl on every rank), or
 * the host callback `all_gather` (same contract as aha_hip_set_seq_parallel's; tests).  world = 1 switches it off.
 * Prompts below AHA_CP_MIN_ROWS (default 2048) tokens or with fewer than 4 * world pages run unsharded. */
int aha_hip_set_context_parallel(aha_model* m, int32_t rank, int32_t world, aha_all_gather_fn all_gather, void* user);
int aha_hip_cp_init_rccl(aha_model* m, const void* unique_id128);
/* Host only (no GPU): the rows rank `rank` of `world` owns in a context-parallel prefill of n_tokens -- out5 = {first row and length of
 * its early chunk, first row and length of its late chunk, page slots per rank of the exchange's staging buffer}.  Returns 1 (and leaves
 * out5 alone) when such a prompt is not sharded (fewer than 4 * world pages, world outside 2..8). */
int aha_hip_debug_cp_plan(int32_t n_tokens, int32_t world, int32_t rank, int32_t* out5);
/* KV hand-back after a sharded prefill ("for single-GPU decode afterwards, all-gather KV to GPU 0";
 * north_star: decode stays single-GPU).  A tensor-parallel prefill leaves every rank with the K / V of ITS kv heads, in its own
 * pages.  aha_hip_kv_export packs them into a contiguous device buffer
 *     [layer][page][this rank's kv head][K block 16 KB | V block 16 KB]      (pages = ceil(cache_len / 64), fragment-major blocks)
 * = the byte image of the pages, so it can cross a collective (RCCL gather / all-gather, 738 MB per rank and 41 k tokens at 8B)
 * untouched; aha_hip_kv_import copies `n_heads` heads starting at head `src_head0` of such a buffer (which holds `src_heads` heads
 * per page) into heads [dst_head0, dst_head0 + n_heads) of THIS model's pages -- an un-sharded model on GPU 0 imports rank r's
 * buffer at dst_head0 = r * kv_heads / T --, maps pages for n_tokens, and sets the cache length and the Qwen3-VL rope_delta so
 * that forward_step / decode_greedy continue exactly as after a single-GPU prefill.  out_dev NULL: only the sizes are returned.
 * `in_bytes` is the size of the buffer behind in_dev: it must hold layers x ceil(n_tokens / 64) x src_heads x 32 KB, else
 * AHA_ERR_INVALID (nothing is read).  The destination must be un-sharded (tp_size 1); an import invalidates the logits an earlier
 * forward call left behind.
 * No reference counterpart (the reference has no collectives): the contract is "decode after export + import == decode after
 * the same prefill on one GPU" (tests/test_tp_gpu.py). */
int aha_hip_kv_export(aha_model* m, void* out_dev, size_t out_bytes, size_t* bytes_needed, size_t* n_tokens, int64_t* rope_delta);
int aha_hip_kv_import(aha_model* m, const void* in_dev, size_t in_bytes, int32_t src_heads, int32_t src_head0, int32_t dst_head0,
                      int32_t n_heads, size_t n_tokens, int64_t rope_delta);
/* Test hook: run the installed all-reduce (RCCL communicator or callback) once on a caller-owned f32 device buffer and
 * wait for it.  Lets a 1-GPU box exercise the RCCL wiring with a communicator of size 1. */
int aha_hip_debug_allreduce(aha_model* m, void* buf_f32_dev, size_t count);

/* ---- introspection used by bench.py / tests ------------------------------------------------------------------ */
size_t aha_hip_cache_len(const aha_model* m);
/* Decode steps the device ran in the last aha_hip_decode_greedy call, including the ones queued past a stop token (at most
 * AHA_DECODE_RUNAHEAD - 1 = 3 by default: the host watches the tokens in pinned memory while later steps are queued). */
int64_t aha_hip_debug_steps_executed(const aha_model* m);
/* Diagnostic: microseconds per decode step at the current cache length, (a) enqueued launch by launch, (b) replayed as one captured
 * hipGraph -- identical kernel arguments in both, results discarded, the cache is cleared afterwards (scripts/bench_graph_step.py). */
int aha_hip_debug_graph_step(aha_model* m, int32_t replays, double* us_launches, double* us_graph);
/* Per-kernel-class HIP-event timing of subsequent forward calls (adds one event pair per launch).  0 = off. */
int aha_hip_set_profiling(aha_model* m, int enable);
/* name: e.g. "gemv", "gemm", "attn_decode", "attn_prefill".  Returns accumulated ms and launch count since enable. */
int aha_hip_get_profile(aha_model* m, const char* kernel_class, double* total_ms, int64_t* launches, double* bytes,
                        double* flops);
/* Debug knob for the paged-KV property tests: 1 => hand out physical pages in a scrambled order. */
int aha_hip_debug_scramble_pages(aha_model* m, int enable);
/* Copies the last hidden state before lm_head (hidden_size floats) / the image embeddings of the last
 * forward_initial (rows x out_hidden floats) to the host, for parity tests of intermediate tensors. */
int aha_hip_debug_last_hidden(aha_model* m, float* out, size_t n);
/* Test tool: fill the LDS of every CU with seeded garbage (on `stream`).  A kernel that consumes LDS it did not stage itself gives
 * run-to-run identical results when its launch is simply repeated and different ones behind different poisons
 * (tests/test_ops_gpu.py::test_kernels_do_not_consume_unstaged_lds). */
int aha_hip_debug_poison_lds(uint32_t seed, void* stream);
/* Test entry of the row-grouped GEMM behind the tensor-parallel prefill's chunked all-gather (csrc/kernels.h launch_gemm_grouped): `groups`
 * row segments of M rows each, all x W^T (N, K): segment g reads A rows [g * a_gstride, + M) (row pitch K) and writes C rows
 * [c_row0 + g * c_gstride, + M) (row pitch ldc), clipped to rows < m_total.  act: plain (0) or gate+up pairs (4). */
int aha_hip_debug_gemm_grouped(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, int32_t ldc, int32_t act,
                               int32_t groups, int32_t a_gstride, int32_t c_gstride, int32_t c_row0, int32_t m_total, void* stream);
/* Test hook: the prefill attention's score chain.  3 (the default since round 5; env AHA_ATTN_SMX) = the scores stay the f32 QK^T
 * accumulators through scale, mask, maximum and exponential, P is rounded to bf16 once for the P.V product; 1 / 0 = the reference's
 * eager path, which materialises `q.k^T` and `* scaling` in bf16 (modules.rs:782-783), with the scale multiply on the matrix pipe
 * (csrc/attn_common.h mfma_diag) / in the vector ALU -- the same bits as each other, the bit-faithful forms.  -1 = back to the
 * default.  Both chains are held to the same parity bounds (profiles/r05_attn_prefill.md). */
int aha_hip_debug_attn_variant(int32_t smx);
/* Test hook: the prefill attention's kernel form (env AHA_ATTN_FORM).  16 = 16 q rows per wave, two 8-wave (or 4-wave) workgroups per CU
 * (csrc/kernels_attn.hip: every score chain, every head dim); 64 = one wave per SIMD, 64 q rows per wave on 32x32x16 MFMAs, 256-row
 * workgroups (csrc/kernels_attn64.hip: f32 score chain, head_dim 128 and the ViT's 72); 65 = the same, software-pipelined inside the wave;
 * -1 = automatic (the 64-row form once its 256-row blocks fill the chip).  Same rounding points in every form; the accumulation order of
 * the two MFMA shapes differs, so outputs agree to the parity bound, not bit for bit. */
int aha_hip_debug_attn_form(int32_t form);
/* Test hook: force the GEMM tile (128, 256, 192 = 256 x 192, 2128 = 256 x 128 on the eight-wave ring kernel) and split-K factor of every
 * following GEMM launch of the process;
 * (0, 0) restores the automatic choice (csrc/kernels_gemm.hip plan_gemm).  tile 1256 / 1192: the persistent kernel on 256- /
 * 192-column tiles wherever it has an instantiation and a workspace (128^2 kernel elsewhere). */
int aha_hip_debug_gemm_plan(int32_t tile, int32_t splitk);
/* Host only (no GPU): the plan the GEMM launcher would pick for a shape -- out3 = {tile (128 | 256), split-K factor, 1 if the
 * columns run as a multiple of 256 + a tail launch}; workspace_bytes = size of the caller's split-K scratch (0 = none); has_residual
 * bit 0 = a residual is added, bit 1 = an RMSNorm of the output rides on the call (o_proj / down_proj in the layer loop).  Lets the
 * CPU tier pin the plans of the BASELINE shapes (csrc/kernels_gemm.hip plan_gemm is a cost model fitted on MI355X). */
int aha_hip_debug_plan_gemm(int32_t M, int32_t N, int32_t K, int32_t act, int32_t has_bias, int32_t has_residual, size_t workspace_bytes,
                            int32_t* out3);
/* Host only (no GPU): the segment lists the persistent GEMM kernel (csrc/kernels_gemm_sk.hip) would walk for a shape on `workers`
 * workgroups (a multiple of 8) with `tile_n` (256 | 192) column tiles: out = 8 ints per segment {m0, n0, first K tile, end K tile,
 * pieces the tile is cut into, this piece's index in K order, first chunk of the tile, counter of the tile}, grouped by worker;
 * off_out[workers + 1] = where each worker's list starts; info7 = {workers, chunks, counters, tiles cut, cut style, cuts, k steps on
 * the slowest worker x 1000}.  Returns the number of segments (at most `cap` are written) or a negative error.  The CPU tier checks
 * that every (tile, K tile) is covered exactly once for the BASELINE shapes. */
int aha_hip_debug_streamk_plan(int32_t M, int32_t N, int32_t K, int32_t tile_n, int32_t workers, size_t workspace_bytes, int32_t* out,
                               int32_t cap, int32_t* off_out, int32_t* info7);
/* CUs the persistent GEMM kernel leaves free (rounded so that its workgroup count stays a multiple of 8; 0 = use every CU; < 0 = take
 * AHA_GEMM_RESERVE_CUS from the environment).  Process-wide.  For tensor-parallel prefill: RCCL's kernels on the communication stream
 * need CUs next to a GEMM whose workgroups each fill one (csrc/model.hip gemm_row_parallel).  Must be the same on every rank only
 * for speed, not for correctness (no collective depends on it). */
int aha_hip_set_gemm_reserved_cus(int32_t n);
int aha_hip_debug_image_embeds(aha_model* m, int which /*0=merged, 1..=deepstack k*/, float* out, size_t n);

/* ---- op-level entry points (device pointers; stream = hipStream_t as void*, NULL = default stream) ---------- */
/* D3: y = x / sqrt(mean(x^2)+eps) * w over the last dim (qwen3/model.rs:79,83,186; modules.rs:512-513). bf16. */
int aha_hip_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int32_t dim, float eps, void* stream);
/* D4/D8 decode: y[n] = sum_k x[k] W[n,k]  (candle_nn::Linear, batch 1).  W (N,K) bf16 row-major, x (K) bf16.
 * norm_w != NULL fuses y = Linear(RMSNorm(x; norm_w, eps)); residual != NULL fuses y = residual + Linear(..). */
int aha_hip_gemv(const void* W, const void* x, void* y, int32_t N, int32_t K, const void* norm_w, float eps,
                 const void* residual, void* stream);
/* D8 decode: y[j] = silu(gate_j . h) * (up_j . h), h = RMSNorm(x) if norm_w else x.  Wg, Wu (I,K) bf16. */
int aha_hip_gemv_gate_up(const void* Wg, const void* Wu, const void* x, void* y, int32_t I, int32_t K,
                         const void* norm_w, float eps, void* stream);
/* D4/D8/V1 prefill: C[M,N] = A[M,K] . W[N,K]^T (+bias[N]) (+residual[M,N]); bf16 in/out, f32 accumulate (MFMA).
 * act: 0 none, 1 gelu_pytorch_tanh, 2 gelu (erf), 3 silu. lda/ldw/ldc in elements. K % 32 == 0. */
int aha_hip_gemm(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                 int32_t ldc, const void* bias, const void* residual, int32_t act, void* stream);
/* D5/M1: q/k RMSNorm over head_dim + rotary embedding (rope.rs:96-132) on a fused qkv activation (S, (nh+2kvh)*d).
 * pos: int32 (3,S) rows T,H,W (all equal for 1-D RoPE).  axis_map: int32[d/2], frequency slot -> row of pos.
 * Writes q_out (S, nh*d) and k_out / v_out (S, kvh*d) contiguous (op-level variant without the paged cache). */
int aha_hip_qknorm_rope(const void* qkv, const void* q_norm_w, const void* k_norm_w, const int32_t* pos,
                        const int32_t* axis_map, void* q_out, void* k_out, void* v_out, int32_t S, int32_t nh,
                        int32_t kvh, int32_t d, float eps, float theta, void* stream);
/* D6/D7 decode attention over a contiguous (kvh, L, d) K/V (op-level variant): o (nh*d) bf16. */
int aha_hip_attn_decode(const void* q, const void* k, const void* v, void* o, int32_t nh, int32_t kvh, int32_t d,
                        int32_t L, float scale, void* stream);
/* D7 prefill attention, causal with q position i attending to k positions <= kv_offset + i; q (S, nh*d),
 * k/v (L, kvh*d) token-major, L = kv_offset + S.  causal = 0 gives full (ViT / audio encoder) attention.  d = 128, or 64 with
 * nh == kvh (the Qwen3-ASR audio encoder's geometry). */
int aha_hip_attn_prefill(const void* q, const void* k, const void* v, void* o, int32_t S, int32_t L, int32_t nh,
                         int32_t kvh, int32_t d, int32_t kv_offset, int32_t causal, float scale, void* stream);
/* V0-pre, host arithmetic: img_smart_resize (src/utils/img_utils.rs:294-331) -- the size Qwen3VLProcessor::process_img
 * (qwen3vl/processor.rs:159-165) resizes an image to: multiples of `factor` (patch * merge = 32), area within
 * [min_pixels, max_pixels] (shortest_edge / longest_edge of the preprocessor config).  AHA_ERR_INVALID when the aspect ratio
 * exceeds 200, as the reference. */
int aha_hip_img_smart_resize(uint32_t h, uint32_t w, uint32_t factor, uint32_t min_pixels, uint32_t max_pixels, uint32_t* h_out,
                             uint32_t* w_out);
/* The video path's host arithmetic, host only (decoding and the swscale resize stay with the caller):
 *  - video_smart_resize (/root/reference/src/utils/video_utils.rs:9-59): the size get_video_data scales the frames to; factor =
 *    patch_size * merge_size, video_ratio = 16 (0 = none), min / max_pixels = the video preprocessor's shortest / longest edge
 *    (qwen3vl/processor.rs:496-505).  Errors carry the reference's messages.
 *  - the frame sampling of get_video_data (processor.rs:481-489,518-535): nframes (sizes the resize) and the sample interval;
 *    the kept frames are the decoded frames whose index is a multiple of the interval.
 *  - calculate_timestamps (processor.rs:283-307): one f32 second value per temporal patch; returns the count. */
int aha_hip_video_smart_resize(uint32_t num_frames, uint32_t h, uint32_t w, uint32_t temporal_factor, uint32_t factor, uint32_t min_pixels,
                               uint32_t max_pixels, uint32_t video_ratio, uint32_t* h_out, uint32_t* w_out);
int aha_hip_video_sample_frames(uint32_t total_frames, float rate, uint32_t fps, uint32_t min_frames, uint32_t max_frames,
                                uint32_t* nframes_out, uint32_t* interval_out);
int64_t aha_hip_video_timestamps(const uint32_t* frame_indices, size_t n, float fps, uint32_t t_merge_size, float* out, size_t cap);
/* V0-pre: DynamicImage::resize_exact(new_w, new_h, FilterType::CatmullRom) (qwen3vl/processor.rs:166) of an RGB8 image
 * (H, W, 3) in device memory into dst (new_h, new_w, 3), device.  Algorithm of crate image 0.25.10 imageops::resize as
 * restated in oracle/image_pre.py ([unverified] against the crate itself): vertical pass into f32, horizontal pass,
 * CatmullRom taps scaled by max(ratio, 1) and normalised, clamp, round half away from zero.  Synchronises the stream. */
int aha_hip_image_resize(const uint8_t* src_hwc, int32_t H, int32_t W, uint8_t* dst_hwc, int32_t new_h, int32_t new_w, void* stream);
/* Host-only debug views (no GPU work) of the tap tables the two pre-processing kernels use, so that the CPU test tier can
 * compare them bit for bit with the restatements: resize taps of one axis (left[n_out], count[n_out], weights concatenated,
 * returns their number) and the polyphase resampling taps (new_f x klen floats for rates already divided by their gcd). */
int aha_hip_debug_resize_taps(int32_t n_in, int32_t n_out, int32_t* left, int32_t* count, float* weights, int64_t weights_cap);
int64_t aha_hip_debug_resample_taps(int32_t orig, int32_t new_f, float* taps, int64_t cap, int32_t* width, int32_t* klen);
/* V0: one RGB8 image (H, W, 3) in device memory, H and W multiples of patch*merge -> the processor's pixel_values rows
 * ((H/patch)*(W/patch), 3*2*patch*patch) bf16 in merge-window order with the frame duplicated to T = 2
 * (/root/reference/src/models/qwen3vl/processor.rs:174-251; img_transform, /root/reference/src/utils/img_utils.rs:272-293). */
int aha_hip_image_to_patches(const uint8_t* img_hwc, void* out, int32_t H, int32_t W, int32_t patch, int32_t merge,
                             const float mean[3], const float std[3], void* stream);
/* The video counterpart (process_videos, /root/reference/src/models/qwen3vl/processor.rs:253-281): T RGB8 frames (T, H, W, 3) in
 * device memory, already sampled and resized (get_video_data's output) -> (ceil(T/2)*(H/patch)*(W/patch), 3*2*patch*patch) bf16
 * rows, a temporal patch = two consecutive frames (an odd last frame is repeated, processor.rs:176-186).  The normalisation
 * runs in bf16 op by op as the reference's does for videos (to_dtype, affine(1/255, 0), broadcast_sub, broadcast_div). */
int aha_hip_video_to_patches(const uint8_t* frames_thwc, void* out, int32_t T, int32_t H, int32_t W, int32_t patch, int32_t merge,
                             const float mean[3], const float std[3], void* stream);
/* A0: Whisper log-mel frontend on the GPU (extract_fbank_features, feature_extraction_whisper.rs:93-115): n_samples f32
 * device samples -> out (128, n_samples/160) f32 device (n_fft 400, hop 160, symmetric Hann, Slaney mel, log10, max-8 clamp,
 * (x+4)/4).  n_samples must be >= 401. */
int aha_hip_logmel(const float* samples, int64_t n_samples, float* out, void* stream);
/* A0-pre: resample_audio_from_vec_f32 (src/utils/audio_utils.rs:590-616), the step between the audio decoder and the
 * feature extractor on the Qwen3-ASR request path (qwen3_asr/processor.rs:76,85: 16 kHz, 1 channel): interleaved PCM f32
 * (n_frames x channels, host) -> mean over channels -> resample_simple (audio_utils.rs:247-255: sinc interpolation, Hann
 * window, lowpass_filter_width 6, rolloff 0.99; kernel :66-151, strided convolution :154-214) -> mono f32 at target_sr
 * (host).  Returns the number of output samples, min(ceil(new * n / orig), (n / orig + 1) * new) with orig / new the rates
 * divided by their gcd (n_frames itself when the rates are equal), or a negative status; out == NULL only queries it. */
int64_t aha_hip_audio_resample(aha_ctx* ctx, const float* pcm, int64_t n_frames, int32_t channels, int32_t orig_sr,
                               int32_t target_sr, float* out, int64_t out_cap);
/* Debug: audio embeddings of the last forward_initial (rows x output_dim floats). */
int aha_hip_debug_audio_embeds(aha_model* m, float* out, size_t n);
/* D11 greedy: first maximal index of an f32 vector. */
int aha_hip_argmax(const float* x, int64_t n, uint32_t* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AHA_HIP_H */
