// extern "C" surface of libaha_hip.so -- see include/aha_hip.h for the contract of every entry point.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "model.h"
#include "audio.h"
#include "vision.h"

namespace aha {
const char* last_error_cstr();
}
using namespace aha;

#define API_GUARD_BEGIN try {
#define API_GUARD_END                                   \
  }                                                     \
  catch (const std::exception& e) {                     \
    set_error(std::string("exception: ") + e.what());   \
    return AHA_ERR_INVALID;                             \
  }                                                     \
  catch (...) {                                         \
    set_error("unknown exception");                     \
    return AHA_ERR_INVALID;                             \
  }

extern "C" {

const char* aha_hip_last_error(void) { return last_error_cstr(); }
const char* aha_hip_version(void) { return "aha-hip 0.2 (gfx950)"; }   // 0.2: aha_mm_input grew (video fields)

int aha_hip_get_dtype(int32_t requested, const char* cfg_dtype, int32_t* out) {
  API_GUARD_BEGIN
  if (!out) {
    set_error("aha_hip_get_dtype: out is null");
    return AHA_ERR_INVALID;
  }
  if (requested >= 0) {   // Some(d) => d
    if (requested != AHA_BF16 && requested != AHA_F16 && requested != AHA_F32) {
      set_error("aha_hip_get_dtype: not a floating-point model dtype");
      return AHA_ERR_INVALID;
    }
    *out = requested;
    return AHA_OK;
  }
  const std::string c = cfg_dtype ? cfg_dtype : "";
  if (c == "float32" || c == "float") *out = AHA_F32;
  else if (c == "float16") *out = AHA_F16;
  else if (c == "bfloat16") *out = AHA_BF16;
  else *out = AHA_F32;
  return AHA_OK;
  API_GUARD_END
}

int aha_hip_check_dtype(int32_t dtype) {
  if (dtype == AHA_BF16) return AHA_OK;
  set_error(std::string("compute dtype ") + (dtype == AHA_F16 ? "f16" : dtype == AHA_F32 ? "f32" : "?") +
            " is not supported: the gfx950 kernels compute in bf16 (f16 / f32 checkpoints are cast to bf16 at load); pass "
            "Some(DType::BF16) or leave the dtype to a bfloat16 checkpoint's config");
  return AHA_ERR_UNSUPPORTED;
}

int aha_hip_init(int device, aha_ctx** out) {
  API_GUARD_BEGIN
  if (!out) {
    set_error("aha_hip_init: out is null");
    return AHA_ERR_INVALID;
  }
  int n = 0;
  AHA_HIP_CHECK(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) {
    set_error("aha_hip_init: device " + std::to_string(device) + " out of range (" + std::to_string(n) + " visible)");
    return AHA_ERR_INVALID;
  }
  AHA_HIP_CHECK(hipSetDevice(device));
  aha_ctx* c = new aha_ctx();
  c->device = device;
  AHA_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return AHA_OK;
  API_GUARD_END
}

void aha_hip_shutdown(aha_ctx* ctx) {
  if (!ctx) return;
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

int aha_hip_model_create(aha_ctx* ctx, const aha_model_desc* desc, const aha_tensor_view* weights, size_t n_weights,
                         aha_model** out) {
  API_GUARD_BEGIN
  if (desc) {
    const int rc = aha_hip_check_dtype(desc->compute_dtype);
    if (rc) return rc;
  }
  return model_create(ctx, desc, weights, n_weights, out);
  API_GUARD_END
}
void aha_hip_model_destroy(aha_model* m) { model_destroy(m); }

int aha_hip_forward_initial(aha_model* m, const uint32_t* input_ids, size_t n_ids, size_t seqlen_offset,
                            const aha_mm_input* mm, float* logits_out, uint32_t* argmax_out) {
  API_GUARD_BEGIN
  if (!m) {
    set_error("null model");
    return AHA_ERR_INVALID;
  }
  return model_forward_initial(m, input_ids, n_ids, seqlen_offset, mm, logits_out, argmax_out);
  API_GUARD_END
}
int aha_hip_forward_step(aha_model* m, uint32_t token, size_t seqlen_offset, float* logits_out, uint32_t* argmax_out) {
  API_GUARD_BEGIN
  if (!m) {
    set_error("null model");
    return AHA_ERR_INVALID;
  }
  return model_forward_step(m, token, seqlen_offset, logits_out, argmax_out);
  API_GUARD_END
}
int aha_hip_clear_cache(aha_model* m) {
  if (!m) {
    set_error("null model");
    return AHA_ERR_INVALID;
  }
  return model_clear_cache(m);
}
int aha_hip_stop_token_ids(const aha_model* m, uint32_t* out, size_t cap) {
  if (!m) {
    set_error("null model");
    return AHA_ERR_INVALID;
  }
  const int n = m->desc.n_stop_tokens;
  for (int i = 0; i < n && (size_t)i < cap; ++i) out[i] = m->desc.stop_tokens[i];
  return n;
}
int aha_hip_decode_greedy(aha_model* m, uint32_t first_token, size_t seqlen_offset, size_t max_new, uint32_t* tokens_out) {
  API_GUARD_BEGIN
  if (!m || !tokens_out) {
    set_error("null argument");
    return AHA_ERR_INVALID;
  }
  return model_decode_greedy(m, first_token, seqlen_offset, max_new, tokens_out);
  API_GUARD_END
}

int aha_hip_sample_candidates(aha_model* m, const uint32_t* context, size_t n_context, float repeat_penalty, float temperature,
                              int32_t k, float* vals_out, uint32_t* idx_out, float* max_out, float* sumexp_out) {
  API_GUARD_BEGIN
  if (!m) {
    set_error("null handle");
    return AHA_ERR_INVALID;
  }
  return model_sample_candidates(m, context, n_context, repeat_penalty, temperature, k, vals_out, idx_out, max_out, sumexp_out);
  API_GUARD_END
}

int aha_hip_last_logits(aha_model* m, float* logits_out) {
  API_GUARD_BEGIN
  if (!m || !logits_out) {
    set_error("null argument");
    return AHA_ERR_INVALID;
  }
  return model_last_logits(m, logits_out);
  API_GUARD_END
}

size_t aha_hip_cache_len(const aha_model* m) { return m ? m->cache_len : 0; }
int64_t aha_hip_debug_steps_executed(const aha_model* m) { return m ? m->steps_executed : 0; }
int aha_hip_debug_graph_step(aha_model* m, int32_t replays, double* us_launches, double* us_graph) {
  if (!m) return AHA_ERR_INVALID;
  API_GUARD_BEGIN
  return model_debug_graph_step(m, replays, us_launches, us_graph);
  API_GUARD_END
}
int aha_hip_kv_export(aha_model* m, void* out_dev, size_t out_bytes, size_t* bytes_needed, size_t* n_tokens, int64_t* rope_delta) {
  if (!m) return AHA_ERR_INVALID;
  API_GUARD_BEGIN
  return model_kv_export(m, out_dev, out_bytes, bytes_needed, n_tokens, rope_delta);
  API_GUARD_END
}
int aha_hip_kv_import(aha_model* m, const void* in_dev, size_t in_bytes, int32_t src_heads, int32_t src_head0, int32_t dst_head0, int32_t n_heads,
                      size_t n_tokens, int64_t rope_delta) {
  if (!m) return AHA_ERR_INVALID;
  API_GUARD_BEGIN
  return model_kv_import(m, in_dev, in_bytes, src_heads, src_head0, dst_head0, n_heads, n_tokens, rope_delta);
  API_GUARD_END
}

int aha_hip_set_profiling(aha_model* m, int enable) {
  if (!m) return AHA_ERR_INVALID;
  int rc = prof_collect(m);
  if (rc) return rc;
  m->profiling = enable != 0;
  if (enable)
    for (auto& a : m->prof_acc) a = aha_model::ProfAcc();
  return AHA_OK;
}
int aha_hip_get_profile(aha_model* m, const char* kernel_class, double* total_ms, int64_t* launches, double* bytes,
                        double* flops) {
  if (!m || !kernel_class) return AHA_ERR_INVALID;
  int rc = prof_collect(m);
  if (rc) return rc;
  auto it = m->prof_cls.find(kernel_class);
  aha_model::ProfAcc a;
  if (it != m->prof_cls.end()) a = m->prof_acc[it->second];
  if (total_ms) *total_ms = a.ms;
  if (launches) *launches = a.n;
  if (bytes) *bytes = a.bytes;
  if (flops) *flops = a.flops;
  return AHA_OK;
}
int aha_hip_debug_scramble_pages(aha_model* m, int enable) {
  if (!m) return AHA_ERR_INVALID;
  m->scramble_pages = enable != 0;
  if (enable && !m->free_pages.empty()) {
    uint64_t s = 0x2545F4914F6CDD1Dull;
    for (size_t i = m->free_pages.size(); i > 1; --i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      std::swap(m->free_pages[i - 1], m->free_pages[(s >> 33) % i]);
    }
  }
  return AHA_OK;
}
int aha_hip_debug_poison_lds(uint32_t seed, void* stream) {
  API_GUARD_BEGIN
  launch_poison_lds(seed, (hipStream_t)stream);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
  API_GUARD_END
}
int aha_hip_debug_attn_variant(int32_t smx) {
  if (smx < -1 || smx > 3 || smx == 2) {
    set_error("debug_attn_variant: smx -1 (default), 0, 1 or 3");
    return AHA_ERR_INVALID;
  }
  set_attn_variant_override(smx);
  return AHA_OK;
}
int aha_hip_debug_attn_form(int32_t form) {
  if (form != -1 && form != 16 && form != 64 && form != 65) {
    set_error("debug_attn_form: -1 (automatic), 16, 64 or 65");
    return AHA_ERR_INVALID;
  }
  set_attn_form_override(form);
  return AHA_OK;
}
int aha_hip_debug_gemm_plan(int32_t tile, int32_t splitk) {
  const bool sk = tile == 1256 || tile == 1192;   // the persistent kernel; splitk = style * 10 + cuts of the last round (0 = the planner's)
  if ((tile != 0 && tile != 128 && tile != 2128 && tile != 256 && tile != 192 && !sk) || splitk < 0 || (!sk && splitk > 8) ||
      (sk && !(splitk <= 4 || (splitk >= 11 && splitk <= 13)))) {
    set_error("debug_gemm_plan: tile must be 0, 128, 2128 (256 x 128), 192 (256 x 192, where instantiated), 256 or 1256 / 1192 (persistent kernel); splitk 0..8 "
              "(persistent: 0..4 equal pieces, 11..13 big pieces + remainder)");
    return AHA_ERR_INVALID;
  }
  set_gemm_plan_override(tile, splitk);
  return AHA_OK;
}
int aha_hip_debug_plan_gemm(int32_t M, int32_t N, int32_t K, int32_t act, int32_t has_bias, int32_t has_residual, size_t workspace_bytes,
                            int32_t* out3) {
  if (!out3 || M <= 0 || N <= 0 || K <= 0) {
    set_error("debug_plan_gemm: bad argument");
    return AHA_ERR_INVALID;
  }
  int o[3];
  debug_plan_gemm(M, N, K, act, has_bias != 0, (has_residual & 1) != 0, workspace_bytes, o, (has_residual & 2) != 0);
  out3[0] = o[0]; out3[1] = o[1]; out3[2] = o[2];
  return AHA_OK;
}
int aha_hip_debug_streamk_plan(int32_t M, int32_t N, int32_t K, int32_t tile_n, int32_t workers, size_t workspace_bytes, int32_t* out,
                               int32_t cap, int32_t* off_out, int32_t* info7) {
  if (!out || cap < 0 || M <= 0 || N <= 0 || K <= 0) {
    set_error("debug_streamk_plan: bad argument");
    return AHA_ERR_INVALID;
  }
  const int n = debug_streamk_plan(M, N, K, tile_n, workers, 8, workspace_bytes, out, cap, off_out, info7);
  if (n < 0) {
    set_error("debug_streamk_plan: K must be a multiple of 64, workers a multiple of 8, tile_n 256 or 192, and the workspace must hold the chunks");
    return AHA_ERR_INVALID;
  }
  return n;
}
int aha_hip_set_gemm_reserved_cus(int32_t n) {
  set_gemm_reserved_cus(n);
  return AHA_OK;
}
int aha_hip_debug_last_hidden(aha_model* m, float* out, size_t n) {
  API_GUARD_BEGIN
  if (!m || !out || n != (size_t)m->desc.hidden_size) {
    set_error("debug_last_hidden: bad size");
    return AHA_ERR_INVALID;
  }
  std::vector<uint16_t> tmp(n);
  AHA_HIP_CHECK(hipMemcpy(tmp.data(), m->d_hlast, n * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {
    uint32_t u = (uint32_t)tmp[i] << 16;
    memcpy(&out[i], &u, 4);
  }
  return AHA_OK;
  API_GUARD_END
}

// ---- op-level entry points --------------------------------------------------------------------------------------
int aha_hip_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int32_t dim, float eps, void* stream) {
  if (!x || !w || !y || dim % 8 || dim > 8192) {
    set_error("rmsnorm: dim must be a multiple of 8 and <= 8192");
    return AHA_ERR_INVALID;
  }
  launch_rmsnorm_rows(x, w, y, rows, dim, dim, dim, eps, (hipStream_t)stream);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}

int aha_hip_gemv(const void* W, const void* x, void* y, int32_t N, int32_t K, const void* norm_w, float eps,
                 const void* residual, void* stream) {
  if (!W || !x || !y || K % 8 || K > 32768) {
    set_error("gemv: K must be a multiple of 8 and <= 32768");
    return AHA_ERR_INVALID;
  }
  GemvArgs g{};
  g.W = W; g.x = x; g.y = y; g.N = N; g.K = K; g.norm_w = norm_w; g.eps = eps; g.residual = residual;
  launch_gemv(g, residual ? GEMV_RESIDUAL : GEMV_STORE, (hipStream_t)stream);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}

int aha_hip_gemv_gate_up(const void* Wg, const void* Wu, const void* x, void* y, int32_t I, int32_t K,
                         const void* norm_w, float eps, void* stream) {
  if (!Wg || !Wu || !x || !y || K % 8 || K > 32768) {
    set_error("gemv_gate_up: bad arguments");
    return AHA_ERR_INVALID;
  }
  GemvArgs g{};
  g.W = Wg; g.W2 = Wu; g.x = x; g.y = y; g.N = I; g.K = K; g.norm_w = norm_w; g.eps = eps;
  launch_gemv(g, GEMV_SILU_MUL, (hipStream_t)stream);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}

int aha_hip_gemm(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw,
                 int32_t ldc, const void* bias, const void* residual, int32_t act, void* stream) {
  if (!A || !W || !C || K % 8 || N % 8 || lda % 8 || ldw % 8 || ldc % 4 || act < 0 || act > ACT_SILU_MUL_PAIRS) {
    set_error("gemm: K, N, lda, ldw must be multiples of 8 (16-byte rows), ldc of 4");
    return AHA_ERR_INVALID;
  }
  if (act == ACT_SILU_MUL_PAIRS && (N % 32 || bias || residual)) {
    set_error("gemm: ACT_SILU_MUL_PAIRS needs N % 32 == 0 and no bias/residual");
    return AHA_ERR_INVALID;
  }
  GemmArgs g{};
  g.A = A; g.W = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.bias = bias; g.residual = residual; g.act = act;
  // op-level entry (tests, scripts): one split-K scratch per DEVICE, grown on demand; not for concurrent callers
  struct OpScratch { void* ws = nullptr; size_t ws_bytes = 0; void* ctrs = nullptr; };
  static std::map<int, OpScratch> scratch_by_dev;
  static std::mutex scratch_mu;   // the map is reachable from several devices / threads; a device's scratch itself is still one caller at a time
  int cur_dev = 0;
  if (hipGetDevice(&cur_dev) != hipSuccess) (void)hipGetLastError();
  std::unique_lock<std::mutex> scratch_lk(scratch_mu);
  OpScratch& sc = scratch_by_dev[cur_dev];
  const size_t want = std::max((size_t)4 * M * N * 4, (size_t)96 << 20);   // split-K slabs / >= 384 chunks of the persistent kernel
  if (want > sc.ws_bytes && want <= ((size_t)1 << 30)) {
    if (sc.ws) hipFree(sc.ws);
    sc.ws = nullptr;
    sc.ws_bytes = 0;
    if (hipMalloc(&sc.ws, want) == hipSuccess) sc.ws_bytes = want;
  }
  // the persistent kernel's per-tile counters (zero between launches)
  if (!sc.ctrs && (hipMalloc(&sc.ctrs, SK_MAX_COUNTERS * 4) != hipSuccess || hipMemset(sc.ctrs, 0, SK_MAX_COUNTERS * 4) != hipSuccess)) sc.ctrs = nullptr;
  g.workspace = sc.ws;
  g.workspace_bytes = sc.ws_bytes;
  g.sk_counters = sc.ws ? sc.ctrs : nullptr;
  scratch_lk.unlock();   // (std::map nodes are stable: `sc` stays valid)
  launch_gemm(g, (hipStream_t)stream);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}

int aha_hip_debug_gemm_grouped(const void* A, const void* W, void* C, int32_t M, int32_t N, int32_t K, int32_t ldc, int32_t act,
                               int32_t groups, int32_t a_gstride, int32_t c_gstride, int32_t c_row0, int32_t m_total, void* stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K % 8 || N % 8 || ldc % 4 || (act != ACT_NONE && act != ACT_SILU_MUL_PAIRS) || groups < 1 ||
      a_gstride < M || c_gstride < 0 || c_row0 < 0 || m_total < 0 || (act == ACT_SILU_MUL_PAIRS && N % 32)) {
    set_error("debug_gemm_grouped: bad argument");
    return AHA_ERR_INVALID;
  }
  GemmArgs g{};
  g.A = A; g.W = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldw = K; g.ldc = ldc; g.act = act;
  g.groups = groups; g.a_gstride = a_gstride; g.c_gstride = c_gstride; g.c_row0 = c_row0; g.m_total = m_total;
  launch_gemm_grouped(g, (hipStream_t)stream);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}

int aha_hip_qknorm_rope(const void* qkv, const void* q_norm_w, const void* k_norm_w, const int32_t* pos,
                        const int32_t* axis_map, void* q_out, void* k_out, void* v_out, int32_t S, int32_t nh,
                        int32_t kvh, int32_t d, float eps, float theta, void* stream) {
  API_GUARD_BEGIN
  if (d != 128) {
    set_error("qknorm_rope: head_dim must be 128");
    return AHA_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  std::vector<float> inv(d / 2);
  for (int i = 0; i < d / 2; ++i) inv[i] = 1.0f / powf(theta, (float)(2 * i) / (float)d);
  float* d_inv = nullptr;
  AHA_HIP_CHECK(hipMalloc((void**)&d_inv, inv.size() * 4));
  AHA_HIP_CHECK(hipMemcpy(d_inv, inv.data(), inv.size() * 4, hipMemcpyHostToDevice));
  RopeArgs r{};
  r.qkv = qkv; r.ld = (int64_t)(nh + 2 * kvh) * d; r.q_norm_w = q_norm_w; r.k_norm_w = k_norm_w; r.pos = pos; r.pos_ld = S;
  r.inv_freq = d_inv; r.axis_map = axis_map; r.q_out = q_out; r.k_out = k_out; r.v_out = v_out;
  r.kv.page_ptrs = nullptr; r.S = S; r.nh = nh; r.kvh = kvh; r.d = d; r.eps = eps;
  launch_qknorm_rope(r, st);
  AHA_HIP_CHECK(hipGetLastError());
  AHA_HIP_CHECK(hipStreamSynchronize(st));
  hipFree(d_inv);
  return AHA_OK;
  API_GUARD_END
}

namespace {
struct TmpPages {
  void* store = nullptr;
  uint64_t* d_ptrs = nullptr;
  int32_t* d_len = nullptr;
  void* fused = nullptr;   // head_dim 64 entry: the [K | V] rows the audio tower's packer reads (freed here: every early return is covered)
  KvLayer kv{};
  ~TmpPages() {
    if (fused) hipFree(fused);
    if (store) hipFree(store);
    if (d_ptrs) hipFree(d_ptrs);
    if (d_len) hipFree(d_len);
  }
};
int build_tmp_pages(TmpPages& t, const void* k, const void* v, int L, int kvh, int d, hipStream_t st) {
  const int npages = (L + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;
  const size_t page_bytes = (size_t)2 * kvh * KV_PAGE_TOKENS * d * 2;
  AHA_HIP_CHECK(hipMalloc(&t.store, page_bytes * npages));
  AHA_HIP_CHECK(hipMemsetAsync(t.store, 0, page_bytes * npages, st));
  std::vector<uint64_t> ptrs(npages);
  // hand the pages out back to front so the test exercises the indirection
  for (int i = 0; i < npages; ++i) ptrs[i] = (uint64_t)(uintptr_t)t.store + (size_t)(npages - 1 - i) * page_bytes;
  AHA_HIP_CHECK(hipMalloc((void**)&t.d_ptrs, npages * 8));
  AHA_HIP_CHECK(hipMemcpy(t.d_ptrs, ptrs.data(), npages * 8, hipMemcpyHostToDevice));
  AHA_HIP_CHECK(hipMalloc((void**)&t.d_len, 4));
  AHA_HIP_CHECK(hipMemcpy(t.d_len, &L, 4, hipMemcpyHostToDevice));
  t.kv.page_ptrs = t.d_ptrs;
  t.kv.layer_off = 0;
  t.kv.kvh = kvh;
  t.kv.d = d;
  launch_kv_pack_pages(k, v, t.kv, L, st);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}
}  // namespace

int aha_hip_attn_decode(const void* q, const void* k, const void* v, void* o, int32_t nh, int32_t kvh, int32_t d,
                        int32_t L, float scale, void* stream) {
  API_GUARD_BEGIN
  if (d != 128 || L <= 0 || nh % kvh || nh / kvh > 16) {
    set_error("attn_decode: head_dim must be 128, L > 0, group size <= 16");
    return AHA_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  TmpPages t;
  int rc = build_tmp_pages(t, k, v, L, kvh, d, st);
  if (rc) return rc;
  const int npages = (L + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;
  const int nsplit = std::max(1, std::min((npages + 3) / 4, 64));
  float *po = nullptr, *pml = nullptr;
  AHA_HIP_CHECK(hipMalloc((void**)&po, (size_t)nsplit * 4 * nh * d * 4));
  AHA_HIP_CHECK(hipMalloc((void**)&pml, (size_t)nsplit * 4 * nh * 2 * 4));
  AttnDecodeArgs a{};
  a.q = q; a.kv = t.kv; a.kv_len = t.d_len; a.part_o = po; a.part_ml = pml; a.o = o; a.nh = nh; a.kvh = kvh; a.d = d;
  a.nsplit = nsplit; a.scale = scale;
  launch_attn_decode(a, st);
  hipError_t e = hipGetLastError();
  hipStreamSynchronize(st);
  hipFree(po);
  hipFree(pml);
  AHA_HIP_CHECK(e);
  return AHA_OK;
  API_GUARD_END
}

int aha_hip_attn_prefill(const void* q, const void* k, const void* v, void* o, int32_t S, int32_t L, int32_t nh,
                         int32_t kvh, int32_t d, int32_t kv_offset, int32_t causal, float scale, void* stream) {
  API_GUARD_BEGIN
  if ((d != 128 && d != 64) || L <= 0 || S <= 0 || nh % kvh || (d == 64 && nh != kvh)) {
    set_error("attn_prefill: head_dim must be 128, or 64 with nh == kvh (the audio encoder's geometry)");
    return AHA_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  TmpPages t;
  if (d == 128) {
    int rc = build_tmp_pages(t, k, v, L, kvh, d, st);
    if (rc) return rc;
  } else {   // head_dim 64: pages through the audio tower's own packer (K | V of a fused row)
    const int npages = (L + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;
    const size_t page_bytes = (size_t)2 * kvh * KV_PAGE_TOKENS * d * 2, row = (size_t)kvh * d * 2;
    AHA_HIP_CHECK(hipMalloc(&t.store, page_bytes * npages));
    AHA_HIP_CHECK(hipMemsetAsync(t.store, 0, page_bytes * npages, st));
    std::vector<uint64_t> ptrs(npages);
    for (int i = 0; i < npages; ++i) ptrs[i] = (uint64_t)(uintptr_t)t.store + (size_t)(npages - 1 - i) * page_bytes;
    AHA_HIP_CHECK(hipMalloc((void**)&t.d_ptrs, npages * 8));
    AHA_HIP_CHECK(hipMemcpy(t.d_ptrs, ptrs.data(), npages * 8, hipMemcpyHostToDevice));
    t.kv.page_ptrs = t.d_ptrs; t.kv.layer_off = 0; t.kv.kvh = kvh; t.kv.d = d;
    AHA_HIP_CHECK(hipMalloc(&t.fused, row * 2 * L));
    AHA_HIP_CHECK(hipMemcpy2DAsync(t.fused, row * 2, k, row, row, L, hipMemcpyDeviceToDevice, st));
    AHA_HIP_CHECK(hipMemcpy2DAsync((char*)t.fused + row, row * 2, v, row, row, L, hipMemcpyDeviceToDevice, st));
    launch_kv_pack_generic(t.fused, (int64_t)2 * kvh * d, 0, kvh * d, t.kv, L, kvh, d, st);
    AHA_HIP_CHECK(hipGetLastError());
  }
  AttnPrefillArgs a{};
  a.q = q; a.kv = t.kv; a.o = o; a.S = S; a.nh = nh; a.kvh = kvh; a.d = d; a.kv_offset = kv_offset; a.kv_total = L;
  a.causal = causal; a.scale = scale;
  launch_attn_prefill(a, st);
  if (const char* reps_env = getenv("AHA_ATTN_TIME")) {  // microbenchmark hook (scripts/bench_attn.py): kernel-only time
    const int reps = atoi(reps_env) > 0 ? atoi(reps_env) : 5;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch_attn_prefill(a, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    fprintf(stderr, "[attn_prefill] S=%d L=%d nh=%d causal=%d: %.3f ms/launch\n", S, L, nh, causal, ms / reps);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  hipError_t e = hipGetLastError();
  hipStreamSynchronize(st);   // (t's destructor frees the pages and the fused rows: after the stream has drained)
  AHA_HIP_CHECK(e);
  return AHA_OK;
  API_GUARD_END
}

int aha_hip_image_to_patches(const uint8_t* img_hwc, void* out, int32_t H, int32_t W, int32_t patch, int32_t merge,
                             const float mean[3], const float std[3], void* stream) {
  if (!img_hwc || !out || !mean || !std || patch <= 0 || merge <= 0 || H % (patch * merge) || W % (patch * merge)) {
    set_error("image_to_patches: H and W must be multiples of patch*merge");
    return AHA_ERR_INVALID;
  }
  launch_image_to_patches(img_hwc, out, H, W, patch, merge, mean, std, (hipStream_t)stream);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}

int aha_hip_video_to_patches(const uint8_t* frames_thwc, void* out, int32_t T, int32_t H, int32_t W, int32_t patch, int32_t merge,
                             const float mean[3], const float std[3], void* stream) {
  if (!frames_thwc || !out || !mean || !std || T <= 0 || patch <= 0 || merge <= 0 || H % (patch * merge) || W % (patch * merge)) {
    set_error("video_to_patches: T > 0 frames, H and W multiples of patch*merge");
    return AHA_ERR_INVALID;
  }
  launch_video_to_patches(frames_thwc, out, T, H, W, patch, merge, mean, std, (hipStream_t)stream);
  AHA_HIP_CHECK(hipGetLastError());
  return AHA_OK;
}

int aha_hip_img_smart_resize(uint32_t h, uint32_t w, uint32_t factor, uint32_t min_pixels, uint32_t max_pixels, uint32_t* h_out,
                             uint32_t* w_out) {
  API_GUARD_BEGIN
  if (!h_out || !w_out) {
    set_error("null argument");
    return AHA_ERR_INVALID;
  }
  return img_smart_resize(h, w, factor, min_pixels, max_pixels, h_out, w_out);
  API_GUARD_END
}

int aha_hip_video_smart_resize(uint32_t num_frames, uint32_t h, uint32_t w, uint32_t temporal_factor, uint32_t factor, uint32_t min_pixels,
                               uint32_t max_pixels, uint32_t video_ratio, uint32_t* h_out, uint32_t* w_out) {
  API_GUARD_BEGIN
  if (!h_out || !w_out) {
    set_error("null argument");
    return AHA_ERR_INVALID;
  }
  return video_smart_resize(num_frames, h, w, temporal_factor, factor, min_pixels, max_pixels, video_ratio, h_out, w_out);
  API_GUARD_END
}
int aha_hip_video_sample_frames(uint32_t total_frames, float rate, uint32_t fps, uint32_t min_frames, uint32_t max_frames,
                                uint32_t* nframes_out, uint32_t* interval_out) {
  API_GUARD_BEGIN
  if (!nframes_out || !interval_out) {
    set_error("null argument");
    return AHA_ERR_INVALID;
  }
  return video_sample_frames(total_frames, rate, fps, min_frames, max_frames, nframes_out, interval_out);
  API_GUARD_END
}
int64_t aha_hip_video_timestamps(const uint32_t* frame_indices, size_t n, float fps, uint32_t t_merge_size, float* out, size_t cap) {
  API_GUARD_BEGIN
  return video_timestamps(frame_indices, n, fps, t_merge_size, out, cap);
  API_GUARD_END
}

int aha_hip_image_resize(const uint8_t* src_hwc, int32_t H, int32_t W, uint8_t* dst_hwc, int32_t new_h, int32_t new_w, void* stream) {
  API_GUARD_BEGIN
  if (!src_hwc || !dst_hwc || H <= 0 || W <= 0 || new_h <= 0 || new_w <= 0) {
    set_error("image_resize: bad arguments");
    return AHA_ERR_INVALID;
  }
  return image_resize(src_hwc, H, W, dst_hwc, new_h, new_w, (hipStream_t)stream);
  API_GUARD_END
}

int aha_hip_debug_resize_taps(int32_t n_in, int32_t n_out, int32_t* left, int32_t* count, float* weights, int64_t weights_cap) {
  API_GUARD_BEGIN
  if (n_in <= 0 || n_out <= 0 || !left || !count || !weights) {
    set_error("debug_resize_taps: bad arguments");
    return AHA_ERR_INVALID;
  }
  return debug_resize_taps(n_in, n_out, left, count, weights, weights_cap);
  API_GUARD_END
}

int64_t aha_hip_debug_resample_taps(int32_t orig, int32_t new_f, float* taps, int64_t cap, int32_t* width, int32_t* klen) {
  API_GUARD_BEGIN
  if (orig <= 0 || new_f <= 0 || !taps || !width || !klen) {
    set_error("debug_resample_taps: bad arguments");
    return AHA_ERR_INVALID;
  }
  return debug_resample_taps(orig, new_f, taps, cap, width, klen);
  API_GUARD_END
}

int aha_hip_get_rope_index_mm(const aha_model_desc* desc, const uint32_t* input_ids, size_t n_ids, const uint32_t* image_grid_thw,
                              int32_t n_images, const uint32_t* video_grid_thw, int32_t n_videos, int32_t* pos_out,
                              int64_t* rope_delta_out) {
  API_GUARD_BEGIN
  if (!desc || !input_ids || !pos_out || !rope_delta_out || (n_images > 0 && !image_grid_thw) || (n_videos > 0 && !video_grid_thw)) {
    set_error("aha_hip_get_rope_index: null argument");
    return AHA_ERR_INVALID;
  }
  if (n_images <= 0 && n_videos <= 0) {
    for (int a = 0; a < 3; ++a)
      for (size_t i = 0; i < n_ids; ++i) pos_out[a * n_ids + i] = (int32_t)i;
    *rope_delta_out = 0;
    return AHA_OK;
  }
  return rope_index_core(*desc, input_ids, n_ids, image_grid_thw, n_images, video_grid_thw, n_videos, pos_out, rope_delta_out);
  API_GUARD_END
}
int aha_hip_get_rope_index(const aha_model_desc* desc, const uint32_t* input_ids, size_t n_ids, const uint32_t* image_grid_thw,
                           int32_t n_images, int32_t* pos_out, int64_t* rope_delta_out) {
  return aha_hip_get_rope_index_mm(desc, input_ids, n_ids, image_grid_thw, n_images, nullptr, 0, pos_out, rope_delta_out);
}
int aha_hip_embed(aha_model* m, const uint32_t* input_ids, size_t n_ids, float* out) {
  API_GUARD_BEGIN
  if (!m) {
    set_error("null model");
    return AHA_ERR_INVALID;
  }
  return model_embed(m, input_ids, n_ids, out);
  API_GUARD_END
}
int aha_hip_config_parse(const char* model_dir, aha_model_desc* out) {
  API_GUARD_BEGIN
  if (!model_dir || !out) {
    set_error("aha_hip_config_parse: null argument");
    return AHA_ERR_INVALID;
  }
  return config_parse(model_dir, out);
  API_GUARD_END
}
int aha_hip_config_torch_dtype(const char* model_dir, char* out, size_t cap) {
  API_GUARD_BEGIN
  if (!model_dir || !out || cap == 0) {
    set_error("aha_hip_config_torch_dtype: null argument");
    return AHA_ERR_INVALID;
  }
  std::string s;
  const int rc = config_torch_dtype(model_dir, &s);
  if (rc) return rc;
  if (s.size() + 1 > cap) {
    set_error("aha_hip_config_torch_dtype: the buffer is too small");
    return AHA_ERR_INVALID;
  }
  memcpy(out, s.c_str(), s.size() + 1);
  return AHA_OK;
  API_GUARD_END
}
int aha_hip_weights_open(const char* model_dir, aha_weights** out) {
  API_GUARD_BEGIN
  if (!model_dir || !out) {
    set_error("aha_hip_weights_open: null argument");
    return AHA_ERR_INVALID;
  }
  return weights_open(model_dir, out);
  API_GUARD_END
}
size_t aha_hip_weights_count(const aha_weights* w) { return w ? w->views.size() : 0; }
int aha_hip_weights_get(const aha_weights* w, size_t index, aha_tensor_view* out) {
  if (!w || !out || index >= w->views.size()) {
    set_error("aha_hip_weights_get: bad handle or index");
    return AHA_ERR_INVALID;
  }
  *out = w->views[index];
  return AHA_OK;
}
void aha_hip_weights_close(aha_weights* w) { delete w; }
int aha_hip_model_load(aha_ctx* ctx, const char* model_dir, size_t kv_reserve_tokens, aha_model** out) {
  API_GUARD_BEGIN
  if (!ctx || !model_dir || !out) {
    set_error("aha_hip_model_load: null argument");
    return AHA_ERR_INVALID;
  }
  return model_load(ctx, model_dir, kv_reserve_tokens, out);
  API_GUARD_END
}

int aha_hip_set_allreduce(aha_model* m, aha_allreduce_fn fn, void* user) {
  if (!m) {
    set_error("null model");
    return AHA_ERR_INVALID;
  }
  m->allreduce_cb = fn;
  m->allreduce_user = user;
  return AHA_OK;
}
int aha_hip_set_seq_parallel(aha_model* m, aha_reduce_scatter_fn reduce_scatter, aha_all_gather_fn all_gather, void* user) {
  if (!m) {
    set_error("set_seq_parallel: null model");
    return AHA_ERR_INVALID;
  }
  if ((reduce_scatter == nullptr) != (all_gather == nullptr)) {
    set_error("set_seq_parallel: install both callbacks or neither");
    return AHA_ERR_INVALID;
  }
  m->reduce_scatter_cb = reduce_scatter;
  m->all_gather_cb = all_gather;
  m->sp_user = user;
  return AHA_OK;
}
int aha_hip_tp_unique_id(void* out128) {
  API_GUARD_BEGIN
  if (!out128) return AHA_ERR_INVALID;
  return tp_unique_id(out128);
  API_GUARD_END
}
int aha_hip_tp_init_rccl(aha_model* m, const void* unique_id128) {
  API_GUARD_BEGIN
  if (!m || !unique_id128) return AHA_ERR_INVALID;
  return tp_init_rccl(m, unique_id128);
  API_GUARD_END
}

int aha_hip_set_context_parallel(aha_model* m, int32_t rank, int32_t world, aha_all_gather_fn all_gather, void* user) {
  if (!m || world < 1 || world > 8 || rank < 0 || rank >= world) {
    set_error("set_context_parallel: rank in [0, world), world in 1..8");
    return AHA_ERR_INVALID;
  }
  if (m->tp_size > 1 && world > 1) {
    set_error("set_context_parallel: the model is tensor-parallel (sharded weights); context parallelism needs the full weights on every rank");
    return AHA_ERR_UNSUPPORTED;
  }
  if (m->desc.head_dim != 128 && world > 1) {
    set_error("set_context_parallel: head_dim 128 only");
    return AHA_ERR_UNSUPPORTED;
  }
  if (m->rccl_comm && (world != m->cp_size || rank != m->cp_rank)) {
    set_error("set_context_parallel: the RCCL communicator of this model was created for another (rank, world)");
    return AHA_ERR_STATE;
  }
  if (world != m->cp_size) m->pf_cap = 0;   // the prefill scratch carries the exchange's staging buffer: re-plan it on the next prefill
  m->cp_rank = rank;
  m->cp_size = world;
  m->cp_all_gather_cb = all_gather;
  m->cp_user = user;
  return AHA_OK;
}
int aha_hip_debug_cp_plan(int32_t n_tokens, int32_t world, int32_t rank, int32_t* out5) {
  if (!out5 || n_tokens <= 0) {
    set_error("debug_cp_plan: bad argument");
    return AHA_ERR_INVALID;
  }
  int o[5];
  if (debug_cp_plan(n_tokens, world, rank, o) != 0) return 1;   // not sharded (too few pages, world out of range)
  for (int i = 0; i < 5; ++i) out5[i] = o[i];
  return AHA_OK;
}
int aha_hip_cp_init_rccl(aha_model* m, const void* unique_id128) {
  API_GUARD_BEGIN
  if (!m || !unique_id128) return AHA_ERR_INVALID;
  return cp_init_rccl(m, unique_id128);
  API_GUARD_END
}

int aha_hip_debug_allreduce(aha_model* m, void* buf, size_t count) {
  API_GUARD_BEGIN
  if (!m || !buf) return AHA_ERR_INVALID;
  int rc;
  if (m->rccl_comm) {
    rc = rccl_allreduce(m, (float*)buf, count);
  } else if (m->allreduce_cb) {
    rc = m->allreduce_cb(buf, count, m->allreduce_user) ? AHA_ERR_STATE : AHA_OK;
  } else {
    set_error("no all-reduce installed");
    return AHA_ERR_STATE;
  }
  if (rc) return rc;
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  return AHA_OK;
  API_GUARD_END
}

int aha_hip_vision_encode(aha_model* m, const aha_mm_input* mm, void* out_dev, int64_t* n_tokens) {
  API_GUARD_BEGIN
  if (!m || !mm) {
    set_error("null argument");
    return AHA_ERR_INVALID;
  }
  return vision_encode(m, mm, out_dev, n_tokens);
  API_GUARD_END
}

int aha_hip_logmel(const float* samples, int64_t n_samples, float* out, void* stream) {
  API_GUARD_BEGIN
  if (!samples || !out || n_samples < 401) {
    set_error("logmel: need more than 400 samples");
    return AHA_ERR_INVALID;
  }
  return logmel_standalone(samples, n_samples, out, (hipStream_t)stream);
  API_GUARD_END
}

int64_t aha_hip_audio_resample(aha_ctx* ctx, const float* pcm, int64_t n_frames, int32_t channels, int32_t orig_sr,
                               int32_t target_sr, float* out, int64_t out_cap) {
  API_GUARD_BEGIN
  if (!ctx || n_frames < 0 || channels < 1 || orig_sr <= 0 || target_sr <= 0 || (n_frames > 0 && !pcm)) {
    set_error("audio_resample: frequencies must be positive, channels >= 1");  // audio_utils.rs:225-227
    return AHA_ERR_INVALID;
  }
  return audio_resample(ctx, pcm, n_frames, channels, orig_sr, target_sr, out, out_cap);
  API_GUARD_END
}

int aha_hip_debug_audio_embeds(aha_model* m, float* out, size_t n) {
  API_GUARD_BEGIN
  if (!m || !out) {
    set_error("null argument");
    return AHA_ERR_INVALID;
  }
  return audio_debug_embeds(m, out, n);
  API_GUARD_END
}

int aha_hip_argmax(const float* x, int64_t n, uint32_t* out_dev, void* stream) {
  API_GUARD_BEGIN
  if (!x || !out_dev || n <= 0) {
    set_error("argmax: bad arguments");
    return AHA_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  float* wv = nullptr;
  uint32_t* wi = nullptr;
  AHA_HIP_CHECK(hipMalloc((void**)&wv, 256 * 4));
  AHA_HIP_CHECK(hipMalloc((void**)&wi, 256 * 4));
  launch_argmax_f32(x, n, wv, wi, out_dev, st);
  hipError_t e = hipGetLastError();
  hipStreamSynchronize(st);
  hipFree(wv);
  hipFree(wi);
  AHA_HIP_CHECK(e);
  return AHA_OK;
  API_GUARD_END
}

int aha_hip_debug_image_embeds(aha_model* m, int which, float* out, size_t n) {
  API_GUARD_BEGIN
  if (!m || !out) {
    set_error("null argument");
    return AHA_ERR_INVALID;
  }
  return vision_debug_embeds(m, which, out, n);
  API_GUARD_END
}

}  // extern "C"
