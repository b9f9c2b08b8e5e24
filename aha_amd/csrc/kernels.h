// Host-callable launchers for the gfx950 kernels.  All pointers are device pointers; all launches are asynchronous
// on `st`.  Shapes/strides are in elements unless a name says bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aha {

// Paged KV cache view of ONE layer.  Page p of this layer lives at page_ptrs[p] + layer_off (bytes):
//   K block  [kvh][KV_PAGE_TOKENS * d]   fragment-major (common.h kpage_elem): the QK^T MFMA operand fragments, 1 KB each
//   V block  [kvh][d * KV_PAGE_TOKENS]   fragment-major (common.h vpage_elem): the P.V operand fragments, slot-permuted
struct KvLayer {
  const uint64_t* page_ptrs;  // device array, byte addresses of each logical page's layer-0 storage
  uint64_t layer_off;         // byte offset of this layer inside a page's slab slot
  int kvh, d;
};

// GEMV_PARTIAL_F32: tensor-parallel row-split projection -- y_f32[n] = un-rounded f32 partial sum over this rank's K slice
enum GemvEpi { GEMV_STORE = 0, GEMV_RESIDUAL = 1, GEMV_SILU_MUL = 2, GEMV_LOGITS = 3, GEMV_PARTIAL_F32 = 4 };

struct GemvArgs {
  const void* W;         // (N,K) bf16 row-major; for GEMV_SILU_MUL the gate matrix (I,K)
  const void* W2;        // GEMV_SILU_MUL: the up matrix (I,K)
  const void* x;         // (K) bf16
  const void* norm_w;    // optional (K) bf16: fuse h = RMSNorm(x; norm_w, eps)
  const void* residual;  // GEMV_RESIDUAL: (N) bf16, may alias y
  void* y;               // (N) bf16   [GEMV_LOGITS: unused]
  float* y_f32;          // GEMV_LOGITS: (N) f32 logits
  float* blk_max;        // GEMV_LOGITS: per-tile (max value, index) partials for the on-device argmax
  uint32_t* blk_idx;
  void* h_out;           // optional (K) bf16: block 0 also writes the normalised input (debug / last hidden)
  int N, K;
  float eps;
  int cached;            // 0 (default): non-temporal weight loads (streamed once); 1: default cache policy
  unsigned long long* trace;  // optional (AHA_GEMV_TRACE): 100 MHz stamps of blocks 0 / 255 / last: start, issued, prologue done,
                              // first group consumed, last group consumed, end
};
void launch_gemv(const GemvArgs& a, GemvEpi epi, hipStream_t st);
int gemv_num_tiles(int N, int K);  // number of (max,idx) partials GEMV_LOGITS writes

void launch_argmax_partials(const float* blk_max, const uint32_t* blk_idx, int n, uint32_t* out, hipStream_t st);
// vocab-parallel lm_head (tensor parallelism): per-rank (max, global index) pair into a zeroed 2T vector / pick after the all-reduce
void launch_argmax_pair(const float* blk_max, const uint32_t* blk_idx, int n, int row0, float* pairs, int rank, int T, hipStream_t st);
void launch_argmax_pick(const float* pairs, int T, uint32_t* out, hipStream_t st);
void launch_argmax_f32(const float* x, int64_t n, float* ws_max, uint32_t* ws_idx, uint32_t* out, hipStream_t st);

// D11 candidates (kernels_sample.hip): repeat penalty into a working copy of the logits; k largest + full-vocabulary softmax
// normaliser.  cand_* hold (sample_stage1_waves(V) + 16) * 64 entries (stage-1 candidates, then the 16 x k intermediates),
// part_* one entry per stage-1 wave, out_ms = {max, sumexp}.
int sample_stage1_waves(int V);
bool sample_shape_ok(int V, int k);
void launch_repeat_penalty(const float* logits, float* work, const uint32_t* ctx, int n, float penalty, int V, hipStream_t st);
void launch_topk_candidates(const float* x, int V, int k, float inv_temp, float* cand_val, unsigned* cand_idx, float* part_m,
                            float* part_s, float* out_val, unsigned* out_idx, float* out_ms, hipStream_t st);

void launch_embed_gather(const void* table, const uint32_t* ids, void* out, int S, int H, hipStream_t st);
void launch_rmsnorm_rows(const void* x, const void* w, void* y, int64_t rows, int dim, int64_t ldx, int64_t ldy,
                         float eps, hipStream_t st);

struct RopeArgs {
  const void* qkv;      // (S, ld) bf16: [q heads | k heads | v heads] per token
  int64_t ld;
  const void* q_norm_w; // (d) bf16
  const void* k_norm_w;
  const int32_t* pos;   // (3, pos_ld) rows T,H,W
  int64_t pos_ld;
  const float* inv_freq;    // (d/2)
  const int32_t* axis_map;  // (d/2) frequency slot -> row of pos
  void* q_out;          // (S, nh*d) bf16
  // paged destination (kv.page_ptrs != nullptr) or contiguous token-major k_out/v_out (S, kvh*d)
  KvLayer kv;
  const int32_t* kv_start;  // device scalar: cache position of token 0 of this call
  void* k_out;
  void* v_out;
  int S, nh, kvh, d;
  float eps;
  int kv_start_host = -1;   // the same value as *kv_start when the caller knows it on the host (prefill): selects the row-vectorised kernel
  const void* rope_tab = nullptr;  // optional (S, 128) bf16: cos[64] | sin[64] of every token's angles (launch_rope_table), else computed in place
  int skip_q = 0;           // row-vectorised prefill kernel only: the q heads are normed and rotated by the attention kernel's Q load (AttnPrefillArgs::q_norm_w); K and V only here
};
void launch_qknorm_rope(const RopeArgs& a, hipStream_t st);
// cos / sin of pos[axis(i)] * inv_freq[i], rounded to bf16 as apply_rotary_pos_emb casts them (rope.rs:96-132): computed ONCE per
// prefill for all layers (precise cosf / sinf of angles up to 1e5 rad cost more than the rest of the rope kernel)
void launch_rope_table(const int32_t* pos, int64_t pos_ld, const float* inv_freq, const int32_t* axis_map, int S, void* tab, hipStream_t st);

// contiguous token-major (L, kvh*d) K,V -> pages (op-level tests and TP KV gather)
void launch_kv_pack_pages(const void* k, const void* v, KvLayer kv, int L, hipStream_t st);

struct AttnDecodeArgs {
  const void* q;           // (nh*d) bf16, normed + roped
  KvLayer kv;
  const int32_t* kv_len;   // device scalar: number of valid tokens (including the one just appended)
  float* part_o;           // workspace (nsplit, nh, d) f32 un-normalised
  float* part_ml;          // workspace (nsplit, nh, 2) running max / sum
  void* o;                 // (nh*d) bf16
  int nh, kvh, d, nsplit;
  float scale;
};
void launch_attn_decode(const AttnDecodeArgs& a, hipStream_t st);

// Fused decode step of the attention block: q/k norm + rope + KV append + split-KV attention (kernels_attn.hip).
// The last split block of a kv head to finish merges the nsplit partials of the head's query heads and writes the bf16
// attention output (o); head_ctr / ctr_target decide who is last (monotonic counters, one 128-byte line per kv head).
struct AttnDecodeFusedArgs {
  const void* qkv;          // ((nh+2kvh)*128) bf16: output of the fused QKV matvec
  const void* q_norm_w;     // (128) bf16
  const void* k_norm_w;
  const float* rope;        // (128) f32: cos[64], sin[64] of the step's rope angles, already rounded to bf16 values
                            // (rope_step_kernel: once per step instead of once per layer and block)
  KvLayer kv;
  int kv_start_v;           // cache slot of the token (host value: the step is enqueued with its lengths known)
  int kv_len_v;             // cache length after the append (= kv_start_v + 1)
  float* part_o;            // (nsplit, nh, 128) f32
  float* part_ml;           // (nsplit, nh, 2) f32
  void* o;                  // (nh*128) bf16 attention output
  unsigned* head_ctr;       // [32 * kvhd]: split blocks of the head that have published their partial (all launches)
  unsigned ctr_target;      // value head_ctr reaches when this launch's nsplit blocks have all arrived
  unsigned long long* trace;  // optional timeline (AHA_ATTN_TRACE)
  int nh, kvh, nsplit;
  float eps, scale;
};
void launch_attn_decode_fused(const AttnDecodeFusedArgs& a, hipStream_t st);

struct AttnPrefillArgs {
  const void* q;           // (S, nh*d) bf16
  KvLayer kv;
  void* o;                 // (S, nh*d) bf16
  int S, nh, kvh, d;
  int kv_offset;           // q row i sees cache positions <= kv_offset + i (causal) or < kv_total (full)
  int kv_total;            // number of valid cache tokens
  int causal;
  float scale;
  int64_t q_ld;            // elements between consecutive q rows; 0 => nh * padded head dim
  int nqb;                 // set by the launcher: > 0 selects the XCD-aware 1-D block order over nqb q blocks x nh heads
  // Optional second causal segment in the same launch (context-parallel rank: its late chunk): q / o rows [S, S + S2) of the same
  // buffers, row S + i sees cache positions <= kv_offset2 + i, < kv_total2.  Needs causal; one launch where the XCD-aware order
  // applies (the late segment's blocks first, the early one's fill its last round), else two launches.
  int S2 = 0, kv_offset2 = 0, kv_total2 = 0;
  int epi_rows = 0;        // set by the launcher: output rows stored in row order through LDS (16 B per lane)
  // head_dim 72 only: every real token's V^T pad row 72 holds 1.0 (kernels_vit.hip vit_rope_pack_kernel), so output row 72 of V^T . P^T is
  // the softmax row sum and the f32-chain kernels may drop their own.  Only the packer that wrote the pages can promise it: unset, the
  // kernels keep the f32 sum of the probabilities.
  int v_ones_row = 0;
  // Rows of the WHOLE request this launch is a part of (0 = S + S2).  The automatic choice of the kernel form goes by it, so that a
  // context-parallel rank's share of a prompt runs the form the un-sharded prompt would: the forms agree to the parity bound, not bit for bit
  // (another MFMA shape = another accumulation order), and tests/test_cp_gpu.py demands bit-identical rows.
  int rows_hint = 0;
  // q-norm + RoPE of Q inside the kernel's Q load (round 6; head_dim 128, the 16-rows-per-wave kernel): `q` then points at the RAW q heads of
  // the qkv GEMM's output (q_ld = its row stride), q_norm_w = the (128) bf16 RMSNorm weight, q_rope_tab = the (rows, 128) bf16 cos | sin
  // table of launch_rope_table for the same rows as q, q_eps the norm's epsilon.  Same operations in the same order as qknorm_rope_rows_kernel
  // (bit-identical q); ask attn_prefill_takes_qfuse first -- the 64-row form does not take it.
  const void* q_norm_w = nullptr;
  const void* q_rope_tab = nullptr;
  float q_eps = 0.f;
};
void launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t st);
bool attn_prefill_takes_qfuse(const AttnPrefillArgs& a);   // would launch_attn_prefill run a kernel that norms + rotates Q itself for these arguments?
// the one-wave-per-SIMD, 64-q-rows-per-wave form (kernels_attn64.hip); false = not a shape of that kernel, nothing launched
bool launch_attn_prefill64(const AttnPrefillArgs& a, hipStream_t st, int pipe);
void set_attn_form_override(int form);     // test hook: -1 = automatic, 16 = the 16-rows-per-wave kernel, 64 / 65 = the 64-row kernel (plain / pipelined)
void set_attn_variant_override(int smx);   // test hook: -1 = the environment's / default choice

// ACT_PARTIAL_F32: tensor-parallel row-split projection -- C is (M,N) f32, the un-rounded partial sums over this rank's K slice
enum GemmAct { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_SILU = 3, ACT_SILU_MUL_PAIRS = 4, ACT_PARTIAL_F32 = 5 };
struct GemmArgs {
  const void* A;  // (M,K) bf16, row stride lda
  const void* W;  // (N,K) bf16, row stride ldw
  void* C;        // (M,N) bf16, row stride ldc   [ACT_SILU_MUL_PAIRS: (M,N/2)]
  const void* bias;      // optional (N) bf16
  const void* residual;  // optional (M,N) bf16 with stride ldc, may alias C
  int M, N, K;
  int64_t lda, ldw, ldc;
  int act;
  // RMSNorm of the finished output rows as part of the call (candle_nn::RmsNorm over N, weight norm_w (N) bf16): norm_out (M, N)
  // bf16, row pitch N.  Where the plan ends in a split-K reduce pass the norm runs inside it (the pass holds the finished row in
  // registers); otherwise launch_gemm appends launch_rmsnorm_rows -- the same values either way.
  const void* norm_w = nullptr;
  void* norm_out = nullptr;
  float norm_eps = 0.f;
  // norm_b != nullptr: the riding norm is a LayerNorm (candle_nn::LayerNorm, weight norm_w + bias norm_b over N: the ViT block's norm1 /
  // norm2, /root/reference/src/models/qwen3vl/model.rs:346-370) instead of an RMSNorm.  Folded into a split-K reduce pass where the plan has
  // one (gemm_splitk_reduce_layernorm_kernel), else launch_gemm appends launch_layernorm_rows: the same bits either way.
  const void* norm_b = nullptr;
  void* workspace = nullptr;     // optional f32 scratch for split-K slabs (splitk * M * N * 4 bytes) / the persistent kernel's chunks
  size_t workspace_bytes = 0;
  void* sk_counters = nullptr;   // optional SK_MAX_COUNTERS zeroed u32 words that belong to `workspace` (one per tile cut along K by the
                                 // persistent kernel, kernels_gemm_sk.hip; every launch leaves them zero again)
  int tile_group = 8;            // band width of the grouped tile order inside an XCD's run (kernels_gemm.hip tile_of_block); 0 = plain
  int partial_rows = 1;          // ACT_PARTIAL_F32 on the four-wave kernel: f32 sums stored in row order through LDS (0: fragment order; A/B)
  // Row groups (launch_gemm_grouped; tensor-parallel prefill: the staging layout of the chunked all-gather, csrc/model.hip
  // norm_gather_gemm): `groups` row segments of M rows each share W.  Segment g reads A rows [g * a_gstride, + M) and writes C rows
  // [c_row0 + g * c_gstride, + M), clipped to rows < m_total.
  int groups = 1, a_gstride = 0, c_gstride = 0, c_row0 = 0, m_total = 0;
};
void launch_gemm(const GemmArgs& a, hipStream_t st);
// One launch over all row segments on the four-wave 256 x 256 / 256 x 192 kernels (plain and gate+up epilogues, whole K tiles, M >= 256);
// any other shape: one launch_gemm per segment.  Every output element is the same K-ordered sum as in an ungrouped GEMM over its row.
void launch_gemm_grouped(const GemmArgs& a, hipStream_t st);
// split-K scratch used by launch_gemm calls of this THREAD whose GemmArgs carry none (the model sets it per forward)
void set_gemm_workspace(void* ws, size_t bytes, void* sk_counters = nullptr);
// ---- the persistent, segment-table-driven kernel (kernels_gemm_sk.hip) ----
constexpr int SK_MAX_COUNTERS = 4096;   // u32 words of GemmArgs::sk_counters
bool streamk_has_kernel(int act, bool has_bias, bool has_res, bool n192);
double streamk_estimate(const GemmArgs& a, int tile_n, int* n_chunks, int* n_split);   // k steps on the slowest worker; < 0: cannot plan
bool launch_gemm_streamk(const GemmArgs& a, int tile_n, hipStream_t st);               // false: nothing was launched
void set_streamk_forced_cut(int code);  // tests: style * 10 + cuts of the last round's tiles (style 0 = equal pieces, 1 = big + remainder); 0 = automatic
int gemm_streamk_workers();
int gemm_streamk_cus();                // CUs of the device, rounded down to a multiple of 8            // workgroups of the persistent kernel = CUs - reserved, a multiple of 8
int get_gemm_reserved_cus();           // the raw setting (-1 = unset)
void set_gemm_reserved_cus(int n);     // CUs the persistent GEMM leaves free (RCCL beside the GEMMs under TP); < 0: AHA_GEMM_RESERVE_CUS
bool acquire_gemm_cu_reservation(int cus);   // the automatic, reference-counted reservation of a communication stream (kernels_gemm_sk.hip)
void release_gemm_cu_reservation();
int debug_streamk_plan(int M, int N, int K, int tile_n, int workers, int group, size_t ws_bytes, int* out, int cap, int* off_out, int* info);
void debug_plan_gemm(int M, int N, int K, int act, bool has_bias, bool has_res, size_t ws_bytes, int* out, bool has_norm = false);   // host only: {tile, splitk, n_split}
void set_gemm_plan_override(int tile, int splitk);  // tests: force the 128 / 256 tile kernel and a split-K factor; 0 = automatic
void get_gemm_workspace(void** ws, size_t* bytes, void** sk_counters);
struct GemmWorkspaceScope {   // nests: the vision tower's own workspace inside the prefill's
  GemmWorkspaceScope(void* ws, size_t bytes, void* sk_counters = nullptr) {
    get_gemm_workspace(&prev_ws, &prev_bytes, &prev_ctrs);
    set_gemm_workspace(ws, bytes, sk_counters);
  }
  ~GemmWorkspaceScope() { set_gemm_workspace(prev_ws, prev_bytes, prev_ctrs); }
  GemmWorkspaceScope(const GemmWorkspaceScope&) = delete;
  GemmWorkspaceScope& operator=(const GemmWorkspaceScope&) = delete;
  void* prev_ws = nullptr;
  size_t prev_bytes = 0;
  void* prev_ctrs = nullptr;
};
// x[i] = bf16(x[i] + bf16(sum[i])): Linear output tensor (all-reduced f32 partials) -> bf16, then the residual add -> bf16
void launch_residual_add_f32(void* x, const float* sum, int64_t n, hipStream_t st);
void launch_poison_lds(uint32_t seed, hipStream_t st);
void launch_residual_add_f32_cols(void* x, int64_t ldx, const float* sum, int64_t lds, int64_t rows, int cols, hipStream_t st);
void launch_row_to_f32(const void* x_bf16_or_null, float* out, int n, hipStream_t st);   // nullptr: zeros
void launch_f32_to_row(const float* x, void* out_bf16, int n, hipStream_t st);

struct StepState;  // model.h
// words the fused decode attention synchronises through (zeroed once at model creation, monotonic afterwards)
constexpr size_t DECODE_SYNC_BYTES = 32768;
constexpr int DECODE_HEAD_CTR_WORD = 6144;  // + 32 * kv head: split-arrival counters of the fused decode attention

}  // namespace aha

// ---- Qwen3-VL vision tower (kernels_vit.hip) -----------------------------------------------------------------------
namespace aha {
constexpr int VIT_DQK = 96;  // ViT head_dim 72: Q/K rows zero-padded to 3 MFMA k-steps
constexpr int VIT_DV = 80;   //                   V block zero-padded to 5 16-row sub-tiles
void launch_layernorm_rows(const void* x, const void* w, const void* b, void* y, int64_t rows, int dim, float eps,
                           hipStream_t st);
// x[n,:] += bilinear pos-embed: sum_i table[idx[i][n],:] * wt[i][n]  (4 corners), every op rounded as the reference
void launch_pos_embed_add(void* x, const void* table, const int32_t* idx, const float* wt, int64_t N, int D,
                          hipStream_t st);
struct VitRopeArgs {
  const void* qkv;          // (N, 3*nh*hd) bf16: [q heads | k heads | v heads]
  const int32_t* rowcol;    // (N, 2) patch (row, col)
  const float* inv_freq;    // (hd/4)
  const int32_t* page_of;   // (N) page index of token n
  const int32_t* slot_of;   // (N) slot (0..63) of token n inside its page
  void* q_out;              // (N, nh, VIT_DQK) bf16, zero padded
  KvLayer kv;               // pages: K block [nh][64][VIT_DQK], V block [nh][VIT_DV][64]
  int N, nh, hd;
  const float* cs_tab = nullptr;   // (N, hd/2, 2) f32: bf16-rounded (cos, sin) per patch and rotary lane (launch_vit_rope_table)
  const int32_t* page_first = nullptr;  // (n_pages) first token of each page (a page holds consecutive tokens of one segment)
  const int32_t* page_cnt = nullptr;    // (n_pages) tokens in the page (64 except a segment's last page)
  int n_pages = 0;
};
void launch_vit_rope_pack(const VitRopeArgs& a, hipStream_t st);
void launch_vit_rope_table(const int32_t* rowcol, const float* inv_freq, int N, int hd, float* tab, hipStream_t st);
void launch_scatter_rows(void* dst, const void* src, const int32_t* rows, int64_t n, int D, int add, hipStream_t st);
// image_pre.hip (V0-pre): img_smart_resize (img_utils.rs:294-331) and resize_exact(.., CatmullRom) of an RGB8 image on the device
int img_smart_resize(uint32_t h, uint32_t w, uint32_t factor, uint32_t min_pixels, uint32_t max_pixels, uint32_t* h_out, uint32_t* w_out);
// the video path's host arithmetic (video_utils.rs:9-59, qwen3vl/processor.rs:283-307,481-535); host only
int video_smart_resize(uint32_t num_frames, uint32_t h, uint32_t w, uint32_t temporal_factor, uint32_t factor, uint32_t min_pixels,
                       uint32_t max_pixels, uint32_t video_ratio, uint32_t* h_out, uint32_t* w_out);
int video_sample_frames(uint32_t total_frames, float rate, uint32_t fps, uint32_t min_frames, uint32_t max_frames, uint32_t* nframes,
                        uint32_t* interval);
int64_t video_timestamps(const uint32_t* frame_indices, size_t n, float fps, uint32_t t_merge, float* out, size_t cap);
int image_resize(const uint8_t* src, int H, int W, uint8_t* dst, int new_h, int new_w, hipStream_t st);
int debug_resize_taps(int n_in, int n_out, int32_t* left, int32_t* count, float* weights, int64_t weights_cap);
void launch_video_to_patches(const uint8_t* frames, void* out, int T, int H, int W, int patch, int merge, const float* mean,
                             const float* stdv, hipStream_t st);
void launch_image_to_patches(const uint8_t* img, void* out, int H, int W, int patch, int merge, const float* mean,
                             const float* stdv, hipStream_t st);
}  // namespace aha

// ---- Qwen3-ASR audio path (kernels_audio.hip) ----------------------------------------------------------------------
namespace aha {
void launch_logmel(const float* x, int64_t L, const float* window, const float* twid, const float* melfb, float* out,
                   float* frame_max, int F, hipStream_t st);
void launch_audio_im2col1(const float* feat, void* out, int F, int C, int Hin, int Win, hipStream_t st);
void launch_im2col_nhwc(const void* in, void* out, int B, int Hin, int Win, int Cin, hipStream_t st);
void launch_audio_tokens_gather(const void* in, void* out, int B, int Fq, int T, int Cc, hipStream_t st);
void launch_sinus_pe_add(void* x, int64_t rows, int d, int T, hipStream_t st);
void launch_kv_pack_generic(const void* src, int64_t ld, int k_off, int v_off, KvLayer kv, int N, int nh, int hd, hipStream_t st);
}  // namespace aha
