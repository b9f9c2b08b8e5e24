// Host orchestration of the Qwen3 decoder stack on one MI355X.
//   create      <- Qwen3Model::new                      /root/reference/src/models/qwen3/model.rs:104-134
//   prefill     <- Qwen3Model::forward_hidden (S > 1)   /root/reference/src/models/qwen3/model.rs:146-189
//   decode step <- the same with S == 1, mask = None    (generate.rs:135-143 drives it once per token)
// The KV cache is paged (64-token pages, K / V blocks fragment-major, see common.h) and must be indistinguishable from
// the reference's Tensor::cat growth (modules.rs:558-566).
#include "model.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>

#include "common.h"
#include "audio.h"
#include "vision.h"

namespace aha {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
const char* last_error_cstr() { return g_last_error.c_str(); }

// ---------------------------------------------------------------------------------------------------------------
ProfScope::ProfScope(aha_model* m_, const char* cls, double bytes, double flops) : m(m_) {
  if (!m->profiling) return;
  auto it = m->prof_cls.find(cls);
  int c;
  if (it == m->prof_cls.end()) {
    c = (int)m->prof_names.size();
    m->prof_cls[cls] = c;
    m->prof_names.push_back(cls);
    m->prof_acc.emplace_back();
  } else {
    c = it->second;
  }
  ProfRec r;
  r.cls = c;
  r.bytes = bytes;
  r.flops = flops;
  hipEventCreate(&r.e0);
  hipEventCreate(&r.e1);
  hipEventRecord(r.e0, m->stream);
  idx = (int)m->prof.size();
  m->prof.push_back(r);
}
ProfScope::~ProfScope() {
  if (idx >= 0) hipEventRecord(m->prof[idx].e1, m->stream);
}
int prof_collect(aha_model* m) {
  if (m->prof.empty()) return AHA_OK;
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  for (auto& r : m->prof) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.e0, r.e1);
    auto& a = m->prof_acc[r.cls];
    a.ms += ms;
    a.bytes += r.bytes;
    a.flops += r.flops;
    a.n += 1;
    hipEventDestroy(r.e0);
    hipEventDestroy(r.e1);
  }
  m->prof.clear();
  return AHA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
int dev_alloc(aha_model* m, size_t bytes, void** out, bool zero) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
  if (e != hipSuccess) {
    set_error("hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? AHA_ERR_OOM : AHA_ERR_HIP;
  }
  if (zero) AHA_HIP_CHECK(hipMemsetAsync(p, 0, bytes, m->stream));
  m->owned.push_back(p);
  *out = p;
  return AHA_OK;
}

const aha_tensor_view* find_tensor(const aha_tensor_view* w, size_t nw, const std::string& name) {
  for (size_t i = 0; i < nw; ++i)
    if (w[i].name && name == w[i].name) return &w[i];
  return nullptr;
}

static inline uint16_t f32_to_bf16_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float f16_to_f32_host(uint16_t h) {
  const uint32_t s = (h >> 15) & 1u, e = (h >> 10) & 0x1fu, f = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (f == 0) u = s << 31;
    else {
      int sh = 0;
      uint32_t ff = f;
      while (!(ff & 0x400u)) { ff <<= 1; ++sh; }
      u = (s << 31) | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((ff & 0x3ffu) << 13);
    }
  } else if (e == 31) u = (s << 31) | 0x7f800000u | (f << 13);
  else u = (s << 31) | ((e + 112u) << 23) | (f << 13);
  float r;
  memcpy(&r, &u, 4);
  return r;
}

// host tensor (bf16 / f16 / f32) -> device bf16, viewed as (rows, cols) row-major, optionally zero-padded
int upload_tensor(aha_model* m, const aha_tensor_view* t, const std::vector<int64_t>& shape, void** out,
                  int64_t pad_rows_to, int64_t pad_cols_to) {
  int64_t n = 1;
  for (auto s : shape) n *= s;
  int64_t tn = 1;
  for (int i = 0; i < t->ndim; ++i) tn *= t->shape[i];
  if (tn != n) {
    std::string got;
    for (int i = 0; i < t->ndim; ++i) got += (i ? "," : "") + std::to_string(t->shape[i]);
    std::string want;
    for (size_t i = 0; i < shape.size(); ++i) want += (i ? "," : "") + std::to_string(shape[i]);
    set_error(std::string("tensor ") + t->name + " has shape (" + got + "), expected (" + want + ")");
    return AHA_ERR_SHAPE;
  }
  const int64_t cols = shape.back();
  const int64_t rows = n / cols;
  const int64_t prow = std::max(rows, pad_rows_to), pcol = std::max(cols, pad_cols_to);
  std::vector<uint16_t> tmp;
  const void* src = t->data;
  if (t->on_device && t->dtype != AHA_BF16) {
    set_error(std::string("tensor ") + t->name + ": device-resident weights must be bf16");
    return AHA_ERR_UNSUPPORTED;
  }
  if (t->dtype == AHA_F32) {
    tmp.resize(n);
    const float* f = (const float*)t->data;
    for (int64_t i = 0; i < n; ++i) tmp[i] = f32_to_bf16_host(f[i]);
    src = tmp.data();
  } else if (t->dtype == AHA_F16) {
    tmp.resize(n);
    const uint16_t* h = (const uint16_t*)t->data;
    for (int64_t i = 0; i < n; ++i) tmp[i] = f32_to_bf16_host(f16_to_f32_host(h[i]));
    src = tmp.data();
  } else if (t->dtype != AHA_BF16) {
    set_error(std::string("tensor ") + t->name + ": unsupported dtype");
    return AHA_ERR_UNSUPPORTED;
  }
  void* d = nullptr;
  const bool padded = prow != rows || pcol != cols;
  int rc = dev_alloc(m, (size_t)prow * pcol * 2, &d, padded);
  if (rc) return rc;
  if (!padded) AHA_HIP_CHECK(hipMemcpy(d, src, (size_t)n * 2, hipMemcpyDefault));
  else {
    AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
    AHA_HIP_CHECK(hipMemcpy2D(d, (size_t)pcol * 2, src, (size_t)cols * 2, (size_t)cols * 2, (size_t)rows, hipMemcpyDefault));
  }
  *out = d;
  return AHA_OK;
}

static int need(const aha_tensor_view* w, size_t nw, const std::string& name, const aha_tensor_view** out) {
  *out = find_tensor(w, nw, name);
  if (!*out) {
    set_error("missing weight tensor: " + name);
    return AHA_ERR_MISSING_WEIGHT;
  }
  return AHA_OK;
}

// copy `rows` rows of a host bf16-convertible tensor into dst rows [row0, row0+rows) of a device (.., cols) bf16 matrix
static int upload_rows_into(aha_model* m, const aha_tensor_view* t, int64_t rows, int64_t cols, void* dst, int64_t row0) {
  int64_t tn = 1;
  for (int i = 0; i < t->ndim; ++i) tn *= t->shape[i];
  if (tn != rows * cols) {
    set_error(std::string("tensor ") + t->name + " has " + std::to_string(tn) + " elements, expected " + std::to_string(rows * cols));
    return AHA_ERR_SHAPE;
  }
  std::vector<uint16_t> tmp;
  const void* src = t->data;
  if (t->on_device && t->dtype != AHA_BF16) {
    set_error(std::string("tensor ") + t->name + ": device-resident weights must be bf16");
    return AHA_ERR_UNSUPPORTED;
  }
  if (t->dtype == AHA_F32) {
    tmp.resize(tn);
    for (int64_t i = 0; i < tn; ++i) tmp[i] = f32_to_bf16_host(((const float*)t->data)[i]);
    src = tmp.data();
  } else if (t->dtype == AHA_F16) {
    tmp.resize(tn);
    for (int64_t i = 0; i < tn; ++i) tmp[i] = f32_to_bf16_host(f16_to_f32_host(((const uint16_t*)t->data)[i]));
    src = tmp.data();
  } else if (t->dtype != AHA_BF16) {
    set_error(std::string("tensor ") + t->name + ": unsupported dtype");
    return AHA_ERR_UNSUPPORTED;
  }
  AHA_HIP_CHECK(hipMemcpy((char*)dst + (size_t)row0 * cols * 2, src, (size_t)tn * 2, hipMemcpyDefault));
  return AHA_OK;
}

// bf16 view of a checkpoint tensor (host tensors are converted into tmp when they are f32 / f16)
static int bf16_source(const aha_tensor_view* t, int64_t expect, std::vector<uint16_t>& tmp, const void** src) {
  int64_t tn = 1;
  for (int i = 0; i < t->ndim; ++i) tn *= t->shape[i];
  if (tn != expect) {
    set_error(std::string("tensor ") + t->name + " has " + std::to_string(tn) + " elements, expected " + std::to_string(expect));
    return AHA_ERR_SHAPE;
  }
  *src = t->data;
  if (t->on_device && t->dtype != AHA_BF16) {
    set_error(std::string("tensor ") + t->name + ": device-resident weights must be bf16");
    return AHA_ERR_UNSUPPORTED;
  }
  if (t->dtype == AHA_F32) {
    tmp.resize(tn);
    for (int64_t i = 0; i < tn; ++i) tmp[i] = f32_to_bf16_host(((const float*)t->data)[i]);
    *src = tmp.data();
  } else if (t->dtype == AHA_F16) {
    tmp.resize(tn);
    for (int64_t i = 0; i < tn; ++i) tmp[i] = f32_to_bf16_host(f16_to_f32_host(((const uint16_t*)t->data)[i]));
    *src = tmp.data();
  } else if (t->dtype != AHA_BF16) {
    set_error(std::string("tensor ") + t->name + ": unsupported dtype");
    return AHA_ERR_UNSUPPORTED;
  }
  return AHA_OK;
}
// copy the (nr x nc) block at (r0, c0) of a (rows x cols) checkpoint matrix into dst rows [dr0, dr0+nr) of a (.. x nc) matrix
static int upload_block(aha_model* m, const aha_tensor_view* t, int64_t rows, int64_t cols, int64_t r0, int64_t nr, int64_t c0,
                        int64_t nc, void* dst, int64_t dr0) {
  std::vector<uint16_t> tmp;
  const void* src;
  int rc = bf16_source(t, rows * cols, tmp, &src);
  if (rc) return rc;
  AHA_HIP_CHECK(hipMemcpy2D((char*)dst + (size_t)dr0 * nc * 2, (size_t)nc * 2, (const char*)src + ((size_t)r0 * cols + c0) * 2,
                            (size_t)cols * 2, (size_t)nc * 2, (size_t)nr, hipMemcpyDefault));
  return AHA_OK;
}

// gate/up -> one (2I, H) matrix of alternating 16-row blocks: [gate 0..15 | up 0..15 | gate 16..31 | up 16..31 | ...]
static int upload_gate_up(aha_model* m, const aha_tensor_view* g, const aha_tensor_view* u, int64_t I_full, int64_t row0, int64_t I,
                          int64_t H, void** out) {
  if (I % 16) {
    set_error("intermediate_size must be a multiple of 16");
    return AHA_ERR_UNSUPPORTED;
  }
  void* d = nullptr;
  int rc = dev_alloc(m, (size_t)2 * I * H * 2, &d);
  if (rc) return rc;
  for (int which = 0; which < 2; ++which) {
    const aha_tensor_view* t = which ? u : g;
    int64_t tn = 1;
    for (int i = 0; i < t->ndim; ++i) tn *= t->shape[i];
    if (tn != I_full * H) {
      set_error(std::string("tensor ") + t->name + " has wrong size");
      return AHA_ERR_SHAPE;
    }
    std::vector<uint16_t> tmp;
    const void* src = t->data;
    if (t->on_device && t->dtype != AHA_BF16) {
      set_error(std::string("tensor ") + t->name + ": device-resident weights must be bf16");
      return AHA_ERR_UNSUPPORTED;
    }
    if (t->dtype == AHA_F32) {
      tmp.resize(tn);
      for (int64_t i = 0; i < tn; ++i) tmp[i] = f32_to_bf16_host(((const float*)t->data)[i]);
      src = tmp.data();
    } else if (t->dtype == AHA_F16) {
      tmp.resize(tn);
      for (int64_t i = 0; i < tn; ++i) tmp[i] = f32_to_bf16_host(f16_to_f32_host(((const uint16_t*)t->data)[i]));
      src = tmp.data();
    } else if (t->dtype != AHA_BF16) {
      set_error("unsupported dtype");
      return AHA_ERR_UNSUPPORTED;
    }
    const size_t blk = (size_t)16 * H * 2;
    AHA_HIP_CHECK(hipMemcpy2D((char*)d + which * blk, 2 * blk, (const char*)src + (size_t)row0 * H * 2, blk, blk, (size_t)(I / 16),
                              hipMemcpyDefault));
  }
  *out = d;
  return AHA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// paged KV cache
static int alloc_slab(aha_model* m) {
  void* p = nullptr;
  const size_t bytes = (size_t)m->layer_stride * m->desc.num_hidden_layers;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    set_error("KV slab hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? AHA_ERR_OOM : AHA_ERR_HIP;
  }
  // pages must never hold NaN bit patterns: masked P (= 0) still multiplies the V slots of the page tail
  AHA_HIP_CHECK(hipMemsetAsync(p, 0, bytes, m->stream));
  m->slabs.push_back(p);
  std::vector<uint64_t> pages(m->pages_per_slab);
  for (size_t i = 0; i < m->pages_per_slab; ++i) pages[i] = (uint64_t)(uintptr_t)p + i * m->page_bytes;
  if (m->scramble_pages) {
    uint64_t s = 0x9e3779b97f4a7c15ull * (m->slabs.size() + 1);
    for (size_t i = pages.size(); i > 1; --i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      std::swap(pages[i - 1], pages[(s >> 33) % i]);
    }
  }
  // free list is popped from the back
  for (size_t i = pages.size(); i > 0; --i) m->free_pages.push_back(pages[i - 1]);
  return AHA_OK;
}

int model_ensure_pages(aha_model* m, size_t tokens) {
  const size_t needp = (tokens + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;
  if (needp <= m->n_pages) return AHA_OK;
  if (needp > m->page_table_cap) {
    size_t cap = std::max<size_t>(1024, m->page_table_cap);
    while (cap < needp) cap *= 2;
    uint64_t* nd = nullptr;
    AHA_HIP_CHECK(hipMalloc((void**)&nd, cap * sizeof(uint64_t)));
    AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (m->d_page_ptrs) AHA_HIP_CHECK(hipFree(m->d_page_ptrs));
    m->d_page_ptrs = nd;
    m->page_table_cap = cap;
    if (m->n_pages)
      AHA_HIP_CHECK(hipMemcpy(m->d_page_ptrs, m->h_page_ptrs.data(), m->n_pages * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  const size_t first_new = m->n_pages;
  while (m->n_pages < needp) {
    if (m->free_pages.empty()) {
      int rc = alloc_slab(m);
      if (rc) return rc;
    }
    const uint64_t p = m->free_pages.back();
    m->free_pages.pop_back();
    if (m->h_page_ptrs.size() <= m->n_pages) m->h_page_ptrs.resize(m->n_pages + 1);
    m->h_page_ptrs[m->n_pages++] = p;
  }
  AHA_HIP_CHECK(hipMemcpyAsync(m->d_page_ptrs + first_new, m->h_page_ptrs.data() + first_new,
                               (m->n_pages - first_new) * sizeof(uint64_t), hipMemcpyHostToDevice, m->stream));
  // h_page_ptrs is pageable: the async copy above is staged synchronously by the runtime, so the vector may be reused
  return AHA_OK;
}

KvLayer model_kv_layer(aha_model* m, int layer) {
  KvLayer kv;
  kv.page_ptrs = m->d_page_ptrs;
  kv.layer_off = (uint64_t)layer * m->layer_stride;
  kv.kvh = m->desc.num_key_value_heads;
  kv.d = m->desc.head_dim;
  return kv;
}

// ---- KV hand-back (include/aha_hip.h aha_hip_kv_export / aha_hip_kv_import) ---------------------------------------------------
// One block per (page, layer, head, K|V): 16 KB = the head's fragment-major K (or V) block of that page, copied as bytes between
// the page (page_ptrs[page] + layer * layer_stride + block offset) and the packed buffer [layer][page][head][K | V].
constexpr int KV_BLOCK_BYTES = KV_PAGE_TOKENS * 128 * 2;
__global__ __launch_bounds__(256) void kv_pack_kernel(const uint64_t* __restrict__ page_ptrs, uint64_t layer_stride, int kvh_local,
                                                      char* __restrict__ buf, int buf_heads, int buf_head0, int local_head0, int n_pages,
                                                      int to_pages) {
  const int page = (int)blockIdx.x, layer = (int)blockIdx.y, h = (int)blockIdx.z >> 1, kv = (int)blockIdx.z & 1;
  char* pg = reinterpret_cast<char*>(page_ptrs[page] + (uint64_t)layer * layer_stride) +
             ((size_t)kv * kvh_local + (size_t)(local_head0 + h)) * KV_BLOCK_BYTES;
  char* bf = buf + ((((size_t)layer * n_pages + page) * buf_heads + (size_t)(buf_head0 + h)) * 2 + kv) * KV_BLOCK_BYTES;
  const u32x4_t* src = reinterpret_cast<const u32x4_t*>(to_pages ? bf : pg);
  u32x4_t* dst = reinterpret_cast<u32x4_t*>(to_pages ? pg : bf);
  u32x4_t v[KV_BLOCK_BYTES / 16 / 256];
#pragma unroll
  for (int i = 0; i < KV_BLOCK_BYTES / 16 / 256; ++i) v[i] = src[threadIdx.x + i * 256];
#pragma unroll
  for (int i = 0; i < KV_BLOCK_BYTES / 16 / 256; ++i) dst[threadIdx.x + i * 256] = v[i];
}

int model_kv_export(aha_model* m, void* out_dev, size_t out_bytes, size_t* bytes_needed, size_t* n_tokens, int64_t* rope_delta) {
  const aha_model_desc& c = m->desc;
  if (c.head_dim != 128) {
    set_error("kv_export: head_dim 128 only");
    return AHA_ERR_UNSUPPORTED;
  }
  const size_t pages = (m->cache_len + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;
  const size_t need_bytes = (size_t)c.num_hidden_layers * pages * c.num_key_value_heads * 2 * KV_BLOCK_BYTES;
  if (bytes_needed) *bytes_needed = need_bytes;
  if (n_tokens) *n_tokens = m->cache_len;
  if (rope_delta) *rope_delta = m->rope_delta;
  if (!out_dev || pages == 0) return AHA_OK;
  if (out_bytes < need_bytes) {
    set_error("kv_export: the buffer holds " + std::to_string(out_bytes) + " bytes, " + std::to_string(need_bytes) + " are needed");
    return AHA_ERR_INVALID;
  }
  AHA_HIP_CHECK(hipSetDevice(m->ctx->device));
  hipLaunchKernelGGL(kv_pack_kernel, dim3((unsigned)pages, (unsigned)c.num_hidden_layers, (unsigned)c.num_key_value_heads * 2), dim3(256), 0,
                     m->stream, m->d_page_ptrs, m->layer_stride, c.num_key_value_heads, (char*)out_dev, c.num_key_value_heads, 0, 0, (int)pages, 0);
  AHA_HIP_CHECK(hipGetLastError());
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));   // the caller hands the buffer to a collective on another stream
  return AHA_OK;
}

int model_kv_import(aha_model* m, const void* in_dev, size_t in_bytes, int src_heads, int src_head0, int dst_head0, int n_heads, size_t n_tokens,
                    int64_t rope_delta) {
  const aha_model_desc& c = m->desc;
  if (c.head_dim != 128) {
    set_error("kv_import: head_dim 128 only");
    return AHA_ERR_UNSUPPORTED;
  }
  if (m->tp_size > 1) {   // the hand-back ends in an UN-sharded model (decode stays single-GPU); a sharded destination would need a head map
    set_error("kv_import: the destination model is tensor-parallel (tp_size " + std::to_string(m->tp_size) + "); import into an un-sharded model");
    return AHA_ERR_UNSUPPORTED;
  }
  if (!in_dev || n_heads <= 0 || src_head0 < 0 || dst_head0 < 0 || src_head0 + n_heads > src_heads || dst_head0 + n_heads > c.num_key_value_heads) {
    set_error("kv_import: head ranges out of bounds (buffer holds " + std::to_string(src_heads) + " heads, the model " +
              std::to_string(c.num_key_value_heads) + ")");
    return AHA_ERR_INVALID;
  }
  if (n_tokens == 0) return AHA_OK;
  const size_t pages = (n_tokens + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS;
  // the kernel reads layers x pages x src_heads x (K | V) blocks of 16 KB from in_dev: the caller states what the buffer holds
  const size_t need_bytes = (size_t)c.num_hidden_layers * pages * (size_t)src_heads * 2 * KV_BLOCK_BYTES;
  if (in_bytes < need_bytes) {
    set_error("kv_import: the buffer holds " + std::to_string(in_bytes) + " bytes, " + std::to_string(need_bytes) + " are needed for " +
              std::to_string(n_tokens) + " tokens of " + std::to_string(src_heads) + " heads");
    return AHA_ERR_INVALID;
  }
  AHA_HIP_CHECK(hipSetDevice(m->ctx->device));
  int rc = model_ensure_pages(m, n_tokens);
  if (rc) return rc;
  hipLaunchKernelGGL(kv_pack_kernel, dim3((unsigned)pages, (unsigned)c.num_hidden_layers, (unsigned)n_heads * 2), dim3(256), 0, m->stream,
                     m->d_page_ptrs, m->layer_stride, c.num_key_value_heads, (char*)in_dev, src_heads, src_head0, dst_head0, (int)pages, 1);
  AHA_HIP_CHECK(hipGetLastError());
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));   // the caller may free / reuse the buffer
  m->cache_len = n_tokens;
  m->rope_delta = rope_delta;
  m->rope_delta_valid = true;
  m->have_logits = false;        // whatever logits an earlier forward left on the device belong to another cache
  m->logits_assembled = false;
  return AHA_OK;
}

static int assemble_logits(aha_model* m);

// D11 candidates: see kernels_sample.hip.  Works on the logits the last forward_initial / forward_step / decode_greedy left
// on the device; the caller slices `ctx` the way use_repeat_penalty does (sample.rs:47-53: the last repeat_last_n generated ids).
int model_sample_candidates(aha_model* m, const uint32_t* ctx, size_t n_ctx, float repeat_penalty, float temperature, int k,
                            float* vals_out, uint32_t* idx_out, float* max_out, float* sumexp_out) {
  const int V = m->desc.vocab_size;
  if (!m->have_logits) {
    set_error("sample_candidates: no forward call has produced logits yet");
    return AHA_ERR_STATE;
  }
  if (!sample_shape_ok(V, k)) {
    set_error("sample_candidates: k must be in [1, 64] (and vocab * k within the stage-2 capacity)");
    return AHA_ERR_UNSUPPORTED;
  }
  if (!(repeat_penalty > 0.f) || (n_ctx && !ctx) || !vals_out || !idx_out) {
    set_error("sample_candidates: bad argument");
    return AHA_ERR_INVALID;
  }
  const int nw = sample_stage1_waves(V);
  // device scratch: cand_val (+ intermediates) | part_m | part_s | out block {vals[64] f32, ms[2] f32, idx[64] u32}; cand_idx separately.
  // One pinned host block carries the context up and the 520-byte out block down (pageable copies cost ~30 us each).
  constexpr size_t OUT_WORDS = 64 + 2 + 64;
  if (!m->d_samp_f) {
    const size_t nc = ((size_t)nw + 16) * 64;  // stage-1 candidates + the 16 x 64 intermediates behind them
    const size_t nf = nc + 2 * (size_t)nw + OUT_WORDS, nu = nc;
    void *pw = nullptr, *pf = nullptr, *pu = nullptr;
    if (int rc = dev_alloc(m, (size_t)V * 4, &pw, false)) return rc;
    if (int rc = dev_alloc(m, nf * 4, &pf, false)) return rc;
    if (int rc = dev_alloc(m, nu * 4, &pu, false)) return rc;
    m->d_samp_work = (float*)pw;
    m->d_samp_f = (float*)pf;
    m->d_samp_u = (unsigned*)pu;
  }
  if (!m->h_samp || n_ctx > m->samp_ctx_cap) {
    const size_t cap = std::max<size_t>(1024, n_ctx * 2);
    void* pd = nullptr;
    if (int rc = dev_alloc(m, cap * 4, &pd, false)) return rc;  // a smaller predecessor stays in m->owned until destroy
    uint32_t* ph = nullptr;
    AHA_HIP_CHECK(hipHostMalloc((void**)&ph, (cap + OUT_WORDS) * 4));
    if (m->h_samp) hipHostFree(m->h_samp);
    m->h_samp = ph;
    m->d_samp_ctx = (uint32_t*)pd;
    m->samp_ctx_cap = cap;
  }
  if (int rc = assemble_logits(m)) return rc;
  const float* src = m->d_logits;
  if (repeat_penalty != 1.0f && n_ctx > 0) {  // sample.rs:47 `repeat_penalty == 1.0` => logits unchanged
    memcpy(m->h_samp, ctx, n_ctx * 4);
    AHA_HIP_CHECK(hipMemcpyAsync(m->d_samp_ctx, m->h_samp, n_ctx * 4, hipMemcpyHostToDevice, m->stream));
    AHA_HIP_CHECK(hipMemcpyAsync(m->d_samp_work, m->d_logits, (size_t)V * 4, hipMemcpyDeviceToDevice, m->stream));
    launch_repeat_penalty(m->d_logits, m->d_samp_work, m->d_samp_ctx, (int)n_ctx, repeat_penalty, V, m->stream);
    src = m->d_samp_work;
  }
  // `&logits / temperature` in LogitsProcessor::sample is an affine by 1/T computed in f64 and applied in f32
  const float inv_temp = temperature > 0.f ? (float)(1.0 / (double)temperature) : 1.0f;
  float* cand_val = m->d_samp_f;
  float* part_m = cand_val + ((size_t)nw + 16) * 64;
  float* part_s = part_m + nw;
  float* out_val = part_s + nw;
  float* out_ms = out_val + 64;
  unsigned* out_idx = reinterpret_cast<unsigned*>(out_ms + 2);
  unsigned* cand_idx = m->d_samp_u;
  launch_topk_candidates(src, V, k, inv_temp, cand_val, cand_idx, part_m, part_s, out_val, out_idx, out_ms, m->stream);
  AHA_HIP_CHECK(hipGetLastError());
  uint32_t* h_out = m->h_samp + m->samp_ctx_cap;
  AHA_HIP_CHECK(hipMemcpyAsync(h_out, out_val, OUT_WORDS * 4, hipMemcpyDeviceToHost, m->stream));
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  memcpy(vals_out, h_out, (size_t)k * 4);
  memcpy(idx_out, h_out + 66, (size_t)k * 4);
  if (max_out) memcpy(max_out, h_out + 64, 4);
  if (sumexp_out) memcpy(sumexp_out, h_out + 65, 4);
  return AHA_OK;
}

int model_last_logits(aha_model* m, float* logits_out) {
  if (!m->have_logits) {
    set_error("last_logits: no forward call has produced logits yet");
    return AHA_ERR_STATE;
  }
  if (int rc = assemble_logits(m)) return rc;
  AHA_HIP_CHECK(hipMemcpyAsync(m->h_logits, m->d_logits, (size_t)m->desc.vocab_size * 4, hipMemcpyDeviceToHost, m->stream));
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  memcpy(logits_out, m->h_logits, (size_t)m->desc.vocab_size * 4);
  return AHA_OK;
}

int model_clear_cache(aha_model* m) {
  // QKNormAttention::clear_kv_cache (modules.rs:581-583): the cache becomes empty; pages go back to the pool
  for (size_t i = m->n_pages; i > 0; --i) m->free_pages.push_back(m->h_page_ptrs[i - 1]);
  m->n_pages = 0;
  m->cache_len = 0;
  m->rope_delta = 0;
  m->rope_delta_valid = false;
  return AHA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
int model_create(aha_ctx* ctx, const aha_model_desc* desc, const aha_tensor_view* w, size_t nw, aha_model** out) {
  if (!ctx || !desc || !out || (!w && nw)) {
    set_error("model_create: null argument");
    return AHA_ERR_INVALID;
  }
  const aha_model_desc& cd = *desc;
  if (cd.head_dim != 128) {
    set_error("only head_dim == 128 is supported by the decoder kernels (Qwen3 family)");
    return AHA_ERR_UNSUPPORTED;
  }
  if (cd.hidden_size <= 0 || cd.intermediate_size <= 0 || cd.num_hidden_layers <= 0 || cd.num_attention_heads <= 0 ||
      cd.num_key_value_heads <= 0 || cd.vocab_size <= 0 || cd.n_stop_tokens < 0 || cd.n_stop_tokens > 8 ||
      (cd.arch == AHA_ARCH_QWEN3VL && (cd.vis_spatial_merge_size <= 0 || cd.vis_patch_size <= 0 || cd.vis_num_heads <= 0))) {
    set_error("model_create: sizes must be positive (and at most 8 stop tokens)");   // they divide below: a 0 would be SIGFPE
    return AHA_ERR_INVALID;
  }
  if (cd.num_attention_heads % cd.num_key_value_heads || cd.num_attention_heads / cd.num_key_value_heads > 16) {
    set_error("unsupported GQA group size");
    return AHA_ERR_UNSUPPORTED;
  }
  if (cd.hidden_size % 8 || cd.intermediate_size % 16) {
    set_error("hidden_size must be a multiple of 8 and intermediate_size of 16");
    return AHA_ERR_UNSUPPORTED;
  }
  AHA_HIP_CHECK(hipSetDevice(ctx->device));
  aha_model* m = new aha_model();
  m->ctx = ctx;
  m->desc = cd;
  m->stream = ctx->stream;
  if (const char* e = getenv("AHA_DECODE_FUSED")) m->decode_fused = atoi(e) != 0;
  int rc = AHA_OK;
  auto fail = [&](int code) {
    model_destroy(m);
    return code;
  };
  // tensor parallelism: from here on m->desc / the locals below hold THIS RANK's share
  const int T = cd.tp_size > 1 ? cd.tp_size : 1, R = T > 1 ? cd.tp_rank : 0;
  if (T > 1 && (R < 0 || R >= T || cd.num_key_value_heads % T || cd.num_attention_heads % T || cd.intermediate_size % (16 * T))) {
    set_error("tp_size must divide num_key_value_heads, num_attention_heads and intermediate_size/16; 0 <= tp_rank < tp_size");
    return fail(AHA_ERR_INVALID);
  }
  m->tp_rank = R;
  m->tp_size = T;
  const int nq_full = cd.num_attention_heads * cd.head_dim, nkv_full = cd.num_key_value_heads * cd.head_dim, I_full = cd.intermediate_size;
  m->desc.num_attention_heads /= T;
  m->desc.num_key_value_heads /= T;
  m->desc.intermediate_size /= T;
  const aha_model_desc& c = m->desc;  // local sizes from here on
  const int H = c.hidden_size, I = I_full / T, d = c.head_dim;
  const int nq = nq_full / T, nkv = nkv_full / T;

  // name prefixes: Qwen3 "model." optional (qwen3/model.rs:105-109); Qwen3-VL "model.language_model." (qwen3vl/model.rs:847-870)
  std::string pre;
  if (c.arch == AHA_ARCH_QWEN3VL) pre = "model.language_model.";
  else if (c.arch == AHA_ARCH_QWEN3ASR) pre = "thinker.model.";  // qwen3_asr/model.rs:318,378
  else pre = find_tensor(w, nw, "model.embed_tokens.weight") ? "model." : "";

  const aha_tensor_view* t = nullptr;
  if ((rc = need(w, nw, pre + "embed_tokens.weight", &t))) return fail(rc);
  if ((rc = upload_tensor(m, t, {c.vocab_size, H}, &m->embed))) return fail(rc);
  // lm_head under tensor parallelism: rank r streams vocab rows [r V/T, (r+1) V/T); the tied
  // case needs no copy (a row window of the embedding table, which every rank holds in full for the gather)
  m->lm_rows = c.vocab_size;
  m->lm_row0 = 0;
  if (T > 1 && c.vocab_size % T == 0) {
    m->lm_rows = c.vocab_size / T;
    m->lm_row0 = R * m->lm_rows;
  }
  if (c.tie_word_embeddings) m->lm_head = (char*)m->embed + (size_t)m->lm_row0 * H * 2;
  else {
    // HF stores lm_head at top level; the reference's Qwen3 (non-VL) branch would look it up under the prefix
    // (qwen3/model.rs:124) -- accept either.
    t = find_tensor(w, nw, c.arch == AHA_ARCH_QWEN3ASR ? "thinker.lm_head.weight" : "lm_head.weight");
    if (!t) t = find_tensor(w, nw, pre + "lm_head.weight");
    if (!t) {
      set_error("missing weight tensor: lm_head.weight");
      return fail(AHA_ERR_MISSING_WEIGHT);
    }
    if (m->lm_rows == c.vocab_size) {
      if ((rc = upload_tensor(m, t, {c.vocab_size, H}, &m->lm_head))) return fail(rc);
    } else {
      if ((rc = dev_alloc(m, (size_t)m->lm_rows * H * 2, &m->lm_head))) return fail(rc);
      if ((rc = upload_block(m, t, c.vocab_size, H, m->lm_row0, m->lm_rows, 0, H, m->lm_head, 0))) return fail(rc);
    }
  }
  if ((rc = need(w, nw, pre + "norm.weight", &t))) return fail(rc);
  if ((rc = upload_tensor(m, t, {H}, &m->final_norm))) return fail(rc);

  m->layers.resize(c.num_hidden_layers);
  for (int li = 0; li < c.num_hidden_layers; ++li) {
    const std::string p = pre + "layers." + std::to_string(li) + ".";
    LayerWeights& L = m->layers[li];
    const aha_tensor_view *tq, *tk, *tv, *tg, *tu;
    if ((rc = need(w, nw, p + "self_attn.q_proj.weight", &tq))) return fail(rc);
    if ((rc = need(w, nw, p + "self_attn.k_proj.weight", &tk))) return fail(rc);
    if ((rc = need(w, nw, p + "self_attn.v_proj.weight", &tv))) return fail(rc);
    if ((rc = dev_alloc(m, (size_t)(nq + 2 * nkv) * H * 2, &L.wqkv))) return fail(rc);
    // column-parallel q/k/v (this rank's heads), row-parallel o_proj (the matching input columns)
    if ((rc = upload_block(m, tq, nq_full, H, (int64_t)R * nq, nq, 0, H, L.wqkv, 0))) return fail(rc);
    if ((rc = upload_block(m, tk, nkv_full, H, (int64_t)R * nkv, nkv, 0, H, L.wqkv, nq))) return fail(rc);
    if ((rc = upload_block(m, tv, nkv_full, H, (int64_t)R * nkv, nkv, 0, H, L.wqkv, nq + nkv))) return fail(rc);
    if ((rc = need(w, nw, p + "self_attn.o_proj.weight", &t))) return fail(rc);
    if ((rc = dev_alloc(m, (size_t)H * nq * 2, &L.wo))) return fail(rc);
    if ((rc = upload_block(m, t, H, nq_full, 0, H, (int64_t)R * nq, nq, L.wo, 0))) return fail(rc);
    // column-parallel gate/up (this rank's intermediate columns), row-parallel down_proj
    if ((rc = need(w, nw, p + "mlp.gate_proj.weight", &tg))) return fail(rc);
    if ((rc = need(w, nw, p + "mlp.up_proj.weight", &tu))) return fail(rc);
    if ((rc = upload_gate_up(m, tg, tu, I_full, (int64_t)R * I, I, H, &L.wgu))) return fail(rc);
    if ((rc = need(w, nw, p + "mlp.down_proj.weight", &t))) return fail(rc);
    if ((rc = dev_alloc(m, (size_t)H * I * 2, &L.wdown))) return fail(rc);
    if ((rc = upload_block(m, t, H, I_full, 0, H, (int64_t)R * I, I, L.wdown, 0))) return fail(rc);
    if ((rc = need(w, nw, p + "input_layernorm.weight", &t))) return fail(rc);
    if ((rc = upload_tensor(m, t, {H}, &L.in_norm))) return fail(rc);
    if ((rc = need(w, nw, p + "post_attention_layernorm.weight", &t))) return fail(rc);
    if ((rc = upload_tensor(m, t, {H}, &L.post_norm))) return fail(rc);
    if ((rc = need(w, nw, p + "self_attn.q_norm.weight", &t))) return fail(rc);
    if ((rc = upload_tensor(m, t, {d}, &L.q_norm))) return fail(rc);
    if ((rc = need(w, nw, p + "self_attn.k_norm.weight", &t))) return fail(rc);
    if ((rc = upload_tensor(m, t, {d}, &L.k_norm))) return fail(rc);
  }

  // rope constants.  inv_freq_i = 1 / theta^(2i/d) in f32 with powf (rope.rs:7-13); the M-RoPE axis of slot i follows
  // apply_interleaved_mrope (rope.rs:454-476): H overwrites i = 1,4,.. < 3*sec[1], W overwrites i = 2,5,.. < 3*sec[2].
  {
    std::vector<float> inv(d / 2);
    std::vector<int32_t> axis(d / 2, 0);
    for (int i = 0; i < d / 2; ++i) {
      inv[i] = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)d);
      if (c.mrope_section[0] + c.mrope_section[1] + c.mrope_section[2] > 0) {
        if (i % 3 == 1 && i < 3 * c.mrope_section[1]) axis[i] = 1;
        if (i % 3 == 2 && i < 3 * c.mrope_section[2]) axis[i] = 2;
      }
    }
    void* p;
    if ((rc = dev_alloc(m, inv.size() * 4, &p))) return fail(rc);
    m->d_inv_freq = (float*)p;
    if ((rc = dev_alloc(m, axis.size() * 4, &p))) return fail(rc);
    m->d_axis_map = (int32_t*)p;
    hipMemcpy(m->d_inv_freq, inv.data(), inv.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(m->d_axis_map, axis.data(), axis.size() * 4, hipMemcpyHostToDevice);
    // `attn_weights * scaling` is a Candle affine op: the f64 scalar is cast to the tensor dtype first [unverified],
    // so the effective scale is bf16(1/sqrt(d)) (modules.rs:476,783; oracle/qwen3.py attn_scale)
    const float s = 1.0f / sqrtf((float)d);
    uint32_t u;
    memcpy(&u, &s, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&m->attn_scale, &u, 4);
  }

  // KV pages
  m->page_bytes = (uint64_t)2 * c.num_key_value_heads * KV_PAGE_TOKENS * d * 2;
  m->pages_per_slab = 64;
  m->layer_stride = m->pages_per_slab * m->page_bytes;
  if (c.kv_reserve_tokens > 0) {
    if ((rc = model_ensure_pages(m, c.kv_reserve_tokens))) return fail(rc);
    model_clear_cache(m);
  }

  // step state + decode scratch
  void* p;
  if ((rc = dev_alloc(m, sizeof(StepState), &p, true))) return fail(rc);
  m->d_state = (StepState*)p;
  if (hipHostMalloc((void**)&m->h_state, sizeof(StepState)) != hipSuccess ||
      hipHostMalloc((void**)&m->h_logits, (size_t)c.vocab_size * 4) != hipSuccess) {
    set_error("model_create: pinned host allocation failed");
    return fail(AHA_ERR_HIP);
  }
  if (hipHostMalloc((void**)&m->h_ring, (RING_CAP + 16) * 4, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer((void**)&m->h_ring_dev, m->h_ring, 0) != hipSuccess) {
    set_error("hipHostMalloc (decode token ring) failed");
    return fail(AHA_ERR_OOM);
  }
  memset(m->h_ring, 0, (RING_CAP + 16) * 4);
  m->h_done = m->h_ring + RING_CAP;
  m->h_done_dev = m->h_ring_dev + RING_CAP;
  m->token_log_cap = 1 << 16;
  if ((rc = dev_alloc(m, m->token_log_cap * 4, &p))) return fail(rc);
  m->d_token_log = (uint32_t*)p;
  if ((rc = dev_alloc(m, (size_t)H * 2, &m->d_x))) return fail(rc);
  if ((rc = dev_alloc(m, (size_t)(nq + 2 * nkv) * 2, &m->d_qkv))) return fail(rc);
  if ((rc = dev_alloc(m, (size_t)nq * 2, &m->d_q))) return fail(rc);
  if ((rc = dev_alloc(m, (size_t)nq * 2, &m->d_attn))) return fail(rc);
  if ((rc = dev_alloc(m, (size_t)I * 2, &m->d_act))) return fail(rc);
  if ((rc = dev_alloc(m, (size_t)H * 2, &m->d_hlast, true))) return fail(rc);
  if ((rc = dev_alloc(m, (size_t)H * 4, &p))) return fail(rc);
  m->d_partial = (float*)p;
  if ((rc = dev_alloc(m, (size_t)c.vocab_size * 4, &p))) return fail(rc);
  m->d_logits = (float*)p;
  const int nt = std::max(gemv_num_tiles(c.vocab_size, H), 512);
  if ((rc = dev_alloc(m, (size_t)nt * 4, &p))) return fail(rc);
  m->d_blk_max = (float*)p;
  if ((rc = dev_alloc(m, (size_t)nt * 4, &p))) return fail(rc);
  m->d_blk_idx = (uint32_t*)p;
  if ((rc = dev_alloc(m, (size_t)m->max_nsplit * 4 * c.num_attention_heads * d * 4, &p))) return fail(rc);
  m->d_part_o = (float*)p;
  if ((rc = dev_alloc(m, (size_t)m->max_nsplit * 4 * c.num_attention_heads * 2 * 4, &p))) return fail(rc);
  m->d_part_ml = (float*)p;
  if ((rc = dev_alloc(m, 128 * 4, &p, true))) return fail(rc);
  m->d_rope = (float*)p;

  // split-arrival counters of the fused decode attention (kernels.h DECODE_SYNC_BYTES): zeroed once, only ever grow; all
  // accesses are agent-scope atomics
  if ((rc = dev_alloc(m, DECODE_SYNC_BYTES, &p, true))) return fail(rc);
  m->d_bar = (unsigned*)p;
  // per-tile arrival counters of the persistent prefill GEMM (kernels_gemm_sk.hip): zeroed once, every launch leaves them zero
  if ((rc = dev_alloc(m, SK_MAX_COUNTERS * 4, &m->d_sk_ctrs, true))) return fail(rc);

  if (c.arch == AHA_ARCH_QWEN3VL) {
    if ((rc = vision_create(m, w, nw))) return fail(rc);
  }
  if (c.arch == AHA_ARCH_QWEN3ASR) {
    if ((rc = audio_create(m, w, nw))) return fail(rc);
  }
  if (hipStreamSynchronize(m->stream) != hipSuccess) {
    set_error(std::string("model_create: ") + hipGetErrorString(hipGetLastError()));
    return fail(AHA_ERR_HIP);
  }
  *out = m;
  return AHA_OK;
}

void model_destroy(aha_model* m) {
  if (!m) return;
  hipStreamSynchronize(m->stream);
  for (auto& r : m->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  vision_destroy(m);
  audio_destroy(m);
  tp_destroy(m);
  for (void* p : m->owned) hipFree(p);
  for (void* p : m->pf_owned) hipFree(p);
  for (void* p : m->slabs) hipFree(p);
  if (m->d_page_ptrs) hipFree(m->d_page_ptrs);
  if (m->h_state) hipHostFree(m->h_state);
  if (m->h_ring) hipHostFree(m->h_ring);
  if (m->h_logits) hipHostFree(m->h_logits);
  if (m->h_samp) hipHostFree(m->h_samp);
  delete m;
}

// ---------------------------------------------------------------------------------------------------------------
static int ensure_prefill_scratch(aha_model* m, size_t S) {
  if (S <= m->pf_cap) return AHA_OK;
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  for (void* p : m->pf_owned) hipFree(p);
  m->pf_owned.clear();
  m->pf_cap = 0;
  const aha_model_desc& c = m->desc;
  const size_t cap = (S + 255) / 256 * 256;
  const size_t H = c.hidden_size, I = c.intermediate_size, nq = (size_t)c.num_attention_heads * c.head_dim,
               nkv = (size_t)c.num_key_value_heads * c.head_dim;
  auto al = [&](size_t bytes, void** out) -> int {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      set_error(std::string("prefill scratch hipMalloc failed: ") + hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? AHA_ERR_OOM : AHA_ERR_HIP;
    }
    m->pf_owned.push_back(p);
    *out = p;
    return AHA_OK;
  };
  int rc;
  if ((rc = al(cap * 4, (void**)&m->p_ids))) return rc;
  if ((rc = al(cap * 3 * 4, (void**)&m->p_pos))) return rc;
  if ((rc = al(cap * 128 * 2, &m->p_rope))) return rc;
  if ((rc = al(cap * H * 2, &m->p_x))) return rc;
  if ((rc = al((cap + 64) * H * 2, &m->p_h))) return rc;   // + padding rows of the sequence-parallel all-gather (T * ceil(S/T) >= S)
  if ((rc = al(cap * (nq + 2 * nkv) * 2, &m->p_qkv))) return rc;
  if ((rc = al(cap * nq * 2, &m->p_q))) return rc;
  if ((rc = al(cap * nq * 2, &m->p_attn))) return rc;
  if ((rc = al(cap * I * 2, &m->p_act))) return rc;
  if (m->tp_size > 1 && (rc = al((cap + 64) * H * 4, (void**)&m->p_partial))) return rc;
  if (m->tp_size > 1 && (rc = al((cap + 64) * H * 2, &m->p_hstage))) return rc;
  if (m->cp_size > 1) {   // one layer's K / V of every rank's pages: cp_size x (max pages of a rank) x kv heads x (K | V) 16-KB blocks
    const size_t pages = cap / KV_PAGE_TOKENS + 1, pmax = pages / m->cp_size + 4;
    m->cp_stage_bytes = std::max((size_t)m->cp_size * pmax * nkv / 128 * 2 * KV_BLOCK_BYTES, (size_t)m->cp_size * H * 2);
    if ((rc = al(m->cp_stage_bytes, &m->p_cp_stage))) return rc;
  }   // chunked all-gather: [chunk][rank][rows of the chunk]
  m->gemm_ws_bytes = std::min((size_t)12 * cap * H * 4, (size_t)1 << 30);  // split-K slabs (up to 8 slices of an N = hidden GEMM, 6 of the qkv one)
  if ((rc = al(m->gemm_ws_bytes, &m->p_gemm_ws))) return rc;
  m->pf_cap = cap;
  return AHA_OK;
}

static int push_state(aha_model* m, uint32_t token, const int64_t pos[3], size_t kv_start, size_t kv_len) {
  StepState* s = m->h_state;
  s->token = token;
  for (int i = 0; i < 3; ++i) s->pos[i] = (int32_t)pos[i];
  s->kv_start = (int32_t)kv_start;
  s->kv_len = (int32_t)kv_len;
  s->next_token = 0;
  s->step = 0;
  AHA_HIP_CHECK(hipMemcpyAsync(m->d_state, s, sizeof(StepState), hipMemcpyHostToDevice, m->stream));
  return AHA_OK;
}

// final RMSNorm (qwen3/model.rs:186) fused into the lm_head matvec of the LAST position only (model.rs:187,142),
// f32 logits + argmax partials -> d_state->next_token
static void enqueue_lm_head(aha_model* m, const void* x_last) {
  const aha_model_desc& c = m->desc;
  GemvArgs g{};
  g.W = m->lm_head;
  g.x = x_last;
  g.norm_w = m->final_norm;
  g.eps = c.rms_norm_eps;
  g.N = m->lm_rows;
  g.K = c.hidden_size;
  g.y_f32 = m->d_logits + m->lm_row0;
  g.blk_max = m->d_blk_max;
  g.blk_idx = m->d_blk_idx;
  g.h_out = m->d_hlast;
  {
    ProfScope ps(m, "gemv", (double)m->lm_rows * c.hidden_size * 2 + c.hidden_size * 2 + m->lm_rows * 4.0, 2.0 * m->lm_rows * c.hidden_size);
    launch_gemv(g, GEMV_LOGITS, m->stream);
  }
  ProfScope ps(m, "argmax", 0, 0);
  const int ntiles = gemv_num_tiles(m->lm_rows, c.hidden_size);
  if (m->lm_rows == c.vocab_size) {
    launch_argmax_partials(m->d_blk_max, m->d_blk_idx, ntiles, &m->d_state->next_token, m->stream);
    return;
  }
  // vocab-parallel: each rank contributes its (max, global index) pair to a zeroed 2T-float vector; the all-reduce (sum)
  // is then an all-gather; the pick keeps the smallest index among equal maxima (candle argmax = first maximal index)
  launch_argmax_pair(m->d_blk_max, m->d_blk_idx, ntiles, m->lm_row0, m->d_partial, m->tp_rank, m->tp_size, m->stream);
  const int rc = model_allreduce(m, m->d_partial, (size_t)2 * m->tp_size);
  if (rc && !m->async_rc) m->async_rc = rc;
  launch_argmax_pick(m->d_partial, m->tp_size, &m->d_state->next_token, m->stream);
}

static void gemv_trace_dump(aha_model* m);

// vocab-parallel lm_head: every rank zeroes the slices it does not own and the all-reduce assembles the full vector
static int assemble_logits(aha_model* m) {
  const aha_model_desc& c = m->desc;
  if (m->lm_rows == c.vocab_size || m->logits_assembled) return AHA_OK;
  if (m->lm_row0 > 0) AHA_HIP_CHECK(hipMemsetAsync(m->d_logits, 0, (size_t)m->lm_row0 * 4, m->stream));
  const int tail0 = m->lm_row0 + m->lm_rows;
  if (tail0 < c.vocab_size) AHA_HIP_CHECK(hipMemsetAsync(m->d_logits + tail0, 0, (size_t)(c.vocab_size - tail0) * 4, m->stream));
  const int rc = model_allreduce(m, m->d_logits, (size_t)c.vocab_size);
  if (rc) return rc;
  m->logits_assembled = true;
  return AHA_OK;
}

static int fetch_outputs(aha_model* m, float* logits_out, uint32_t* argmax_out) {
  const aha_model_desc& c = m->desc;
  m->have_logits = true;
  m->logits_assembled = false;
  if (logits_out) {
    const int rc = assemble_logits(m);
    if (rc) return rc;
  }
  if (logits_out) AHA_HIP_CHECK(hipMemcpyAsync(m->h_logits, m->d_logits, (size_t)c.vocab_size * 4, hipMemcpyDeviceToHost, m->stream));
  AHA_HIP_CHECK(hipMemcpyAsync(&m->h_state->next_token, &m->d_state->next_token, 4, hipMemcpyDeviceToHost, m->stream));
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  if (m->d_gemv_trace) gemv_trace_dump(m);
  if (m->d_attn_trace) {
    const int L = m->desc.num_hidden_layers;
    std::vector<unsigned long long> t((size_t)L * 12);
    if (hipMemcpy(t.data(), m->d_attn_trace, t.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      for (int b = 0; b < 2; ++b) {
        double v[5] = {};
        int n = 0;
        for (int li = 1; li < L; ++li) {
          const unsigned long long* p = t.data() + (size_t)li * 12 + b * 6;
          if (p[0] == 0 || p[4] == 0) continue;
          for (int k = 0; k < 5; ++k) v[k] += (p[k + 1] > p[k] ? (double)(p[k + 1] - p[k]) : 0.0) * 0.01;
          ++n;
        }
        if (n) fprintf(stderr, "[attn trace] block %d: prologue %.2f | pages %.2f | merge+publish %.2f | arrive %.2f | final merge %.2f us\n", b, v[0] / n, v[1] / n, v[2] / n, v[3] / n, v[4] / n);
      }
    }
  }
  if (logits_out) memcpy(logits_out, m->h_logits, (size_t)c.vocab_size * 4);
  if (argmax_out) *argmax_out = m->h_state->next_token;
  return AHA_OK;
}


// ---- tensor-parallel seam ---------------------------------------------------------------------------------------------
// Row-parallel projections (o_proj, down_proj): with tp_size > 1 each rank holds a K slice, produces un-rounded f32
// partial sums, the partials are all-reduced, and only then does the reference's rounding chain run
// (Linear output -> bf16, + residual -> bf16).  Summing in f32 keeps the result equal to the single-GPU one up to f32
// summation order.
int model_allreduce(aha_model* m, float* buf, size_t count) {
  if (m->tp_size <= 1) return AHA_OK;
  ProfScope ps(m, "allreduce", (double)count * 4, 0);
  if (m->rccl_comm) return rccl_allreduce(m, buf, count);
  if (!m->allreduce_cb) {
    set_error("tp_size > 1 but no all-reduce was installed (aha_hip_set_allreduce / aha_hip_tp_init_rccl)");
    return AHA_ERR_STATE;
  }
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  if (m->allreduce_cb(buf, count, m->allreduce_user) != 0) {
    set_error("all-reduce callback failed");
    return AHA_ERR_STATE;
  }
  return AHA_OK;
}
// Sequence-parallel prefill (include/aha_hip.h aha_hip_set_seq_parallel): in-place reduce-scatter of f32 partial sums over
// row slices, in-place all-gather of bf16 rows.  RCCL communicator if the library owns one, else the host callbacks.
static bool seq_parallel_on(const aha_model* m) {
  static const bool env_on = [] { const char* e = getenv("AHA_TP_SP"); return e ? atoi(e) != 0 : true; }();
  return env_on && m->tp_size > 1 && (m->rccl_comm || (m->reduce_scatter_cb && m->all_gather_cb));
}
static int model_reduce_scatter(aha_model* m, float* buf, size_t count_per_rank) {
  ProfScope ps(m, "reduce_scatter", (double)count_per_rank * m->tp_size * 4, 0);
  if (m->rccl_comm) return rccl_reduce_scatter(m, buf, count_per_rank);
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  if (m->reduce_scatter_cb(buf, count_per_rank, m->sp_user) != 0) {
    set_error("reduce-scatter callback failed");
    return AHA_ERR_STATE;
  }
  return AHA_OK;
}
static int model_all_gather(aha_model* m, void* buf, size_t bytes_per_rank) {
  ProfScope ps(m, "all_gather", (double)bytes_per_rank * m->tp_size, 0);
  if (m->rccl_comm) return rccl_all_gather(m, buf, bytes_per_rank);
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  if (m->all_gather_cb(buf, bytes_per_rank, m->sp_user) != 0) {
    set_error("all-gather callback failed");
    return AHA_ERR_STATE;
  }
  return AHA_OK;
}
static void gemv_row_parallel(aha_model* m, GemvArgs g) {
  if (m->tp_size <= 1) {
    launch_gemv(g, GEMV_RESIDUAL, m->stream);
    return;
  }
  g.y_f32 = m->d_partial;
  launch_gemv(g, GEMV_PARTIAL_F32, m->stream);
  const int rc = model_allreduce(m, m->d_partial, (size_t)g.N);
  if (rc && !m->async_rc) m->async_rc = rc;
  launch_residual_add_f32(g.y, m->d_partial, g.N, m->stream);
}
// rows_per_rank > 0 selects the sequence-parallel form: the partial sums are reduce-scattered over row slices and only this
// rank's rows [tp_rank * rows_per_rank, ...) of the residual stream are updated (the caller all-gathers the NORMALISED rows).
// Column chunks of a sequence-parallel row-parallel projection ("chunk ... to overlap with GEMMs").  The
// projection is cut along N (output columns) -- NOT along the rows, so that rank r keeps owning the contiguous rows
// [r * rows_per_rank, ...) -- into `nch` GEMMs whose f32 partial sums land in separate (rows_pad x N / nch) blocks; the reduce-scatter
// of block j (count_per_rank = rows_per_rank * N / nch: rank r receives its rows of that column block) runs on the communication
// stream while the matrix cores compute block j + 1.  Same sums in the same order as the unchunked form: every output element is one
// K-sum and one sum over ranks either way (tests/test_tp_gpu.py compares both with the all-reduce path bit for bit).  At cfg 5
// (S = 40 980, T = 8) a projection's reduce-scatter moves 7/8 x 671 MB per rank against ~1.2 ms of GEMM per 1024-column block.
static int tp_overlap_chunks(const aha_model* m, const GemmArgs& g) {
  const char* e = getenv("AHA_TP_OVERLAP_CHUNKS");
  const char* er = getenv("AHA_TP_OVERLAP_MIN_ROWS");
  int nch = e ? atoi(e) : 4;
  const int min_rows = er ? atoi(er) : 2048;
  if (nch > 8) nch = 8;
  if (nch <= 1 || g.M < min_rows) return 1;
  while (nch > 1 && (g.N % nch != 0 || (g.N / nch) % 256 != 0)) --nch;
  return nch;
}
static int ensure_comm_stream(aha_model* m) {
  if (m->comm_stream) return AHA_OK;
  // RCCL's kernels run on the communication stream BESIDE the next column block's GEMM.  A grid of one-tile blocks fills every CU with
  // a 512-register wave per SIMD + 128 KiB of LDS, so nothing else can be placed until a block retires; the persistent GEMM kernel takes
  // a worker count instead (kernels_gemm_sk.hip).  Leave 16 CUs (two per XCD) to the collective unless the caller chose a number
  // (aha_hip_set_gemm_reserved_cus / AHA_GEMM_RESERVE_CUS); process-wide while this model's communicator lives.
  // The setting is restored when this model's communicator is torn down (tp_rccl.hip): un-sharded models and op-level GEMMs of the same
  // process are planned on all CUs again from then on (round-4 advisor).
  // (reference-counted: several models of one process may each hold a share, kernels_gemm_sk.hip acquire_gemm_cu_reservation)
  m->reserved_cus_set = acquire_gemm_cu_reservation(16);
  int lo = 0, hi = 0;
  AHA_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // (numerically lowest = highest priority)
  AHA_HIP_CHECK(hipStreamCreateWithPriority(&m->comm_stream, hipStreamNonBlocking, hi));
  for (auto& ev : m->ev_gemm) AHA_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (auto& ev : m->ev_ag) AHA_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  AHA_HIP_CHECK(hipEventCreateWithFlags(&m->ev_comm, hipEventDisableTiming));
  return AHA_OK;
}

// rows_per_rank > 0 selects the sequence-parallel form: the partial sums are reduce-scattered over row slices and only this
// rank's rows [tp_rank * rows_per_rank, ...) of the residual stream are updated (the caller all-gathers the NORMALISED rows).
static int gemm_row_parallel(aha_model* m, GemmArgs g, int rows_per_rank = 0) {
  if (m->tp_size <= 1) {
    launch_gemm(g, m->stream);
    return AHA_OK;
  }
  void* xres = g.C;
  const int nch = rows_per_rank > 0 ? tp_overlap_chunks(m, g) : 1;
  if (nch > 1) {
    const int N = g.N, Nc = N / nch;
    const size_t rows_pad = (size_t)rows_per_rank * m->tp_size;
    const bool on_comm_stream = m->rccl_comm != nullptr;   // host-callback seam (tests): the callback synchronises, no overlap to set up
    int rc;
    if (on_comm_stream && (rc = ensure_comm_stream(m))) return rc;
    // an error inside the loop must not leave the compute stream free to overwrite p_partial while the communication stream still
    // reads it: join the streams before returning
    auto fail = [&](int code) {
      if (on_comm_stream && hipEventRecord(m->ev_comm, m->comm_stream) == hipSuccess) (void)hipStreamWaitEvent(m->stream, m->ev_comm, 0);
      return code;
    };
    for (int j = 0; j < nch; ++j) {
      GemmArgs gj = g;
      gj.W = (const bf16_t*)g.W + (int64_t)j * Nc * g.ldw;
      gj.C = m->p_partial + (size_t)j * rows_pad * Nc;
      gj.N = Nc;
      gj.ldc = Nc;
      gj.residual = nullptr;
      gj.norm_w = nullptr;
      gj.act = ACT_PARTIAL_F32;
      launch_gemm(gj, m->stream);
      float* bj = m->p_partial + (size_t)j * rows_pad * Nc;
      if (on_comm_stream) {
        // (no ProfScope here: it would time m->stream, and this collective runs on the communication stream; the wait for it is
        // what the compute stream sees -- the "rs_wait" scope below)
        if (hipEventRecord(m->ev_gemm[j], m->stream) != hipSuccess || hipStreamWaitEvent(m->comm_stream, m->ev_gemm[j], 0) != hipSuccess) {
          set_error("gemm_row_parallel: event hand-off to the communication stream failed");
          return fail(AHA_ERR_HIP);
        }
        if ((rc = rccl_reduce_scatter(m, bj, (size_t)rows_per_rank * Nc, m->comm_stream))) return fail(rc);
      } else if ((rc = model_reduce_scatter(m, bj, (size_t)rows_per_rank * Nc))) {
        return rc;
      }
    }
    if (on_comm_stream) {
      ProfScope ps(m, "rs_wait", (double)rows_pad * N * 4, 0);   // what the compute stream waits for the overlapped reduce-scatters
      AHA_HIP_CHECK(hipEventRecord(m->ev_comm, m->comm_stream));
      AHA_HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_comm, 0));
    }
    const int64_t r0 = (int64_t)m->tp_rank * rows_per_rank, r1 = std::min<int64_t>(g.M, r0 + rows_per_rank);
    for (int j = 0; j < nch && r1 > r0; ++j)
      launch_residual_add_f32_cols((char*)xres + (r0 * N + (int64_t)j * Nc) * 2, N, m->p_partial + (size_t)j * rows_pad * Nc + r0 * Nc, Nc, r1 - r0,
                                   Nc, m->stream);
    return AHA_OK;
  }
  g.C = m->p_partial;
  g.residual = nullptr;
  g.act = ACT_PARTIAL_F32;
  launch_gemm(g, m->stream);
  if (rows_per_rank > 0) {
    int rc = model_reduce_scatter(m, m->p_partial, (size_t)rows_per_rank * g.N);
    if (rc) return rc;
    const int64_t r0 = (int64_t)m->tp_rank * rows_per_rank, r1 = std::min<int64_t>(g.M, r0 + rows_per_rank);
    if (r1 > r0)
      launch_residual_add_f32((char*)xres + r0 * g.N * 2, m->p_partial + r0 * g.N, (r1 - r0) * g.N, m->stream);
    return AHA_OK;
  }
  int rc = model_allreduce(m, m->p_partial, (size_t)g.M * g.N);
  if (rc) return rc;
  launch_residual_add_f32(xres, m->p_partial, (int64_t)g.M * g.N, m->stream);
  return AHA_OK;
}
// RMSNorm of the prefill rows into p_h.  Sequence-parallel: each rank normalises its own row slice and the bf16 rows are
// all-gathered (the slices are the only valid rows of the residual stream on their rank).
static int prefill_norm(aha_model* m, const void* w, int S, int rows_per_rank) {
  const aha_model_desc& c = m->desc;
  const int H = c.hidden_size;
  if (rows_per_rank <= 0) {
    ProfScope ps(m, "elem", (double)S * H * 4, 0);
    launch_rmsnorm_rows(m->p_x, w, m->p_h, S, H, H, H, c.rms_norm_eps, m->stream);
    return AHA_OK;
  }
  const int64_t r0 = (int64_t)m->tp_rank * rows_per_rank, r1 = std::min<int64_t>(S, r0 + rows_per_rank);
  if (r1 > r0) {
    ProfScope ps(m, "elem", (double)(r1 - r0) * H * 4, 0);
    launch_rmsnorm_rows((const char*)m->p_x + r0 * H * 2, w, (char*)m->p_h + r0 * H * 2, r1 - r0, H, H, H, c.rms_norm_eps, m->stream);
  }
  return model_all_gather(m, m->p_h, (size_t)rows_per_rank * H * 2);
}

// Sequence-parallel column-parallel projection with the all-gather of its input in row chunks, overlapped with the GEMM (round 4; the
// round-3 verdict: "the all-gather of the normalised rows is still one serial call in front of each column-parallel GEMM").
// Every rank's row slice [r * spr, + spr) is cut at the same offsets into chunks of w rows (a multiple of 256: whole row tiles); chunk c
// of all ranks is gathered into its own block of the staging buffer, laid out [rank][rows of the chunk] as ncclAllGather delivers it, on
// the communication stream, while the matrix cores run chunk c - 1: ONE grouped GEMM launch per chunk (launch_gemm_grouped: T row
// segments of the staging block -> the token-order rows r * spr + off .. of the output).  Rows are independent in a GEMM, so every
// output element is the same K-ordered sum as in the unchunked form (tests/test_tp_gpu.py: bit-identical logits and KV).
// cfg 5 on 8 GPUs: the gather moves 7/8 x 336 MB per rank in front of 0.25 ms (qkv) / 0.95 ms (gate+up) of GEMM per rank.
static int tp_ag_chunk_rows(const aha_model* m, int S, int spr) {
  const char* e = getenv("AHA_TP_AG_CHUNKS");
  const char* er = getenv("AHA_TP_AG_MIN_ROWS");
  const int nch = std::min(e ? atoi(e) : 4, 8), min_rows = er ? atoi(er) : 4096;
  if (nch <= 1 || S < min_rows) return 0;
  const int w = ((spr + nch - 1) / nch + 255) / 256 * 256;
  return w < spr ? w : 0;   // one chunk: nothing to overlap
}
static int norm_gather_gemm(aha_model* m, const void* norm_w, const GemmArgs& g, int S, int spr) {
  const aha_model_desc& c = m->desc;
  const int H = c.hidden_size, T = m->tp_size;
  const int w = spr > 0 ? tp_ag_chunk_rows(m, S, spr) : 0;
  if (w <= 0) {
    int rc = prefill_norm(m, norm_w, S, spr);
    if (rc) return rc;
    ProfScope ps(m, "gemm", ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.ldc) * 2, 2.0 * g.M * g.N * g.K);
    launch_gemm(g, m->stream);
    return AHA_OK;
  }
  const int64_t r0 = (int64_t)m->tp_rank * spr, r1 = std::min<int64_t>(S, r0 + spr);
  if (r1 > r0) {
    ProfScope ps(m, "elem", (double)(r1 - r0) * H * 4, 0);
    launch_rmsnorm_rows((const char*)m->p_x + r0 * H * 2, norm_w, (char*)m->p_h + r0 * H * 2, r1 - r0, H, H, H, c.rms_norm_eps, m->stream);
  }
  const bool on_comm = m->rccl_comm != nullptr;   // host-callback seam (tests): the callback synchronises, nothing to overlap
  int rc;
  if (on_comm) {
    if ((rc = ensure_comm_stream(m))) return rc;
    AHA_HIP_CHECK(hipEventRecord(m->ev_comm, m->stream));            // the normalised rows are ready
    AHA_HIP_CHECK(hipStreamWaitEvent(m->comm_stream, m->ev_comm, 0));
  }
  const int nch = (spr + w - 1) / w;
  auto join = [&](int code) {   // on an error: the compute stream must not run ahead of collectives still reading / writing the staging
    if (on_comm && hipEventRecord(m->ev_comm, m->comm_stream) == hipSuccess) (void)hipStreamWaitEvent(m->stream, m->ev_comm, 0);
    return code;
  };
  auto gather = [&](int ci) -> int {
    const int off = ci * w, wc = std::min(w, spr - off);
    char* stage = (char*)m->p_hstage + (size_t)T * off * H * 2;
    const size_t bytes = (size_t)wc * H * 2;
    // this rank's rows of the chunk into its slot (rows past the end of the sequence -- the last rank's padding -- carry whatever the
    // buffer holds: their outputs are clipped by the GEMM)
    hipStream_t cs = on_comm ? m->comm_stream : m->stream;
    AHA_HIP_CHECK(hipMemcpyAsync(stage + (size_t)m->tp_rank * bytes, (const char*)m->p_h + (r0 + off) * H * 2, bytes, hipMemcpyDeviceToDevice, cs));
    if (on_comm) {
      if ((rc = rccl_all_gather(m, stage, bytes, m->comm_stream))) return rc;
      AHA_HIP_CHECK(hipEventRecord(m->ev_ag[ci], m->comm_stream));
      return AHA_OK;
    }
    return model_all_gather(m, stage, bytes);
  };
  if (on_comm)
    for (int ci = 0; ci < nch; ++ci)
      if ((rc = gather(ci))) return join(rc);
  for (int ci = 0; ci < nch; ++ci) {
    const int off = ci * w, wc = std::min(w, spr - off);
    if (on_comm) {
      ProfScope ps(m, "ag_wait", (double)T * wc * H * 2, 0);   // what the compute stream waits for chunk ci of the gather
      AHA_HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_ag[ci], 0));
    } else if ((rc = gather(ci))) {
      return rc;
    }
    GemmArgs gc = g;
    gc.A = (const char*)m->p_hstage + (size_t)T * off * H * 2;
    gc.M = wc;
    gc.groups = T; gc.a_gstride = wc; gc.c_gstride = spr; gc.c_row0 = off; gc.m_total = S;
    ProfScope ps(m, "gemm", ((double)T * wc * g.K + (double)g.N * g.K + (double)T * wc * g.ldc) * 2, 2.0 * T * wc * g.N * g.K);
    launch_gemm_grouped(gc, m->stream);
  }
  return AHA_OK;
}

// ---- context-parallel prefill (round 4) ----------------------------------------------------------------------------------------------
// MI355X has 288 GB per GPU and the 8B checkpoint is 16 GB: for the long-context prefill of BASELINE cfg 5 every rank can hold the FULL
// weights, own a share of the prompt's ROWS, and run every GEMM / norm / rope on its own rows with no collective at all -- a GEMM's rows
// are independent.  The one op that couples rows is causal attention, which needs the K / V of every earlier position: per layer, the
// ranks all-gather the layer's K / V pages (8B: 147 456 B per position and layer -> 40 980 tokens = 168 MB per layer, 7/8 of it inbound
// per rank, against 2 x 294 MB of bf16 all-gather + 2 x 587 MB of f32 reduce-scatter per layer and rank in the tensor-parallel form).
// Ownership: the prompt's KV pages (64 tokens) are cut into 2 x world contiguous chunks; rank r owns chunks r and 2 x world - 1 - r (the
// "zigzag" that balances causal attention: an early chunk with few visible keys rides with a late one with many), so every page is
// written by exactly one rank, chunk boundaries are page boundaries, and rank 0 owns the LAST rows -- it ends the prefill holding the
// last hidden row, and, like every rank, the complete KV cache: decode continues on it without a hand-back.
// Results: every output row is computed by the same kernels on the same values as in the single-GPU prefill; with the GEMM plan pinned
// (plans depend on M) the logits and the cache are bit-identical to it (tests/test_cp_gpu.py).
struct CpPlan { int world, pmax; int a0[8], an[8], b0[8], bn[8]; };   // rank r's two page ranges [a0, a0 + an), [b0, b0 + bn)
struct RowSeg { int r0, len, l0; };                                     // prompt rows [r0, r0 + len) = rows [l0, l0 + len) of the rank's activation buffers
static bool cp_make_plan(int S, int world, int rank, CpPlan* p, std::vector<RowSeg>* segs) {
  const int P = (S + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS, nc = 2 * world;
  if (world < 2 || world > 8 || P < 2 * nc) return false;   // at least two pages per chunk
  auto edge = [&](int j) { return (int)((int64_t)j * P / nc); };
  p->world = world;
  p->pmax = 0;
  for (int r = 0; r < world; ++r) {
    p->a0[r] = edge(r); p->an[r] = edge(r + 1) - edge(r);
    p->b0[r] = edge(nc - 1 - r); p->bn[r] = edge(nc - r) - edge(nc - 1 - r);
    p->pmax = std::max(p->pmax, p->an[r] + p->bn[r]);
  }
  segs->clear();
  for (int k = 0; k < 2; ++k) {
    const int pg0 = k ? p->b0[rank] : p->a0[rank], n = k ? p->bn[rank] : p->an[rank];
    const int r0 = pg0 * KV_PAGE_TOKENS, r1 = std::min(S, (pg0 + n) * KV_PAGE_TOKENS);
    segs->push_back(RowSeg{r0, r1 - r0, 0});
  }
  return true;
}
int debug_cp_plan(int S, int world, int rank, int* out5) {   // host only (tests): {r0_a, len_a, r0_b, len_b, pmax}; -1: this prompt is not sharded
  CpPlan p{};
  std::vector<RowSeg> segs;
  if (rank < 0 || rank >= world || !cp_make_plan(S, world, rank, &p, &segs)) return -1;
  out5[0] = segs[0].r0; out5[1] = segs[0].len; out5[2] = segs[1].r0; out5[3] = segs[1].len; out5[4] = p.pmax;
  return 0;
}
// staging slot s = rank * pmax + j  <->  the j-th page rank owns; one block copies 4 KB of the page's layer slice (kv heads x (K | V) x 16 KB)
__global__ __launch_bounds__(256) void kv_cp_copy_kernel(const uint64_t* __restrict__ page_ptrs, uint64_t layer_off, int slice_bytes,
                                                         char* __restrict__ stage, CpPlan p, int only_rank, int skip_rank, int to_pages) {
  const int rank = (int)blockIdx.x / p.pmax, j = (int)blockIdx.x % p.pmax;
  if ((only_rank >= 0 && rank != only_rank) || rank == skip_rank) return;
  int page;
  if (j < p.an[rank]) page = p.a0[rank] + j;
  else if (j < p.an[rank] + p.bn[rank]) page = p.b0[rank] + (j - p.an[rank]);
  else return;
  char* pg = reinterpret_cast<char*>(page_ptrs[page] + layer_off) + (size_t)blockIdx.y * 4096;
  char* sg = stage + (size_t)blockIdx.x * slice_bytes + (size_t)blockIdx.y * 4096;
  const u32x4_t v = *reinterpret_cast<const u32x4_t*>((to_pages ? sg : pg) + threadIdx.x * 16);
  *reinterpret_cast<u32x4_t*>((to_pages ? pg : sg) + threadIdx.x * 16) = v;
}
static int cp_all_gather(aha_model* m, void* buf, size_t bytes_per_rank) {
  if (m->rccl_comm) return rccl_all_gather(m, buf, bytes_per_rank, m->stream);
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  if (!m->cp_all_gather_cb || m->cp_all_gather_cb(buf, bytes_per_rank, m->cp_user) != 0) {
    set_error("context-parallel all-gather callback failed");
    return AHA_ERR_STATE;
  }
  return AHA_OK;
}
// after the layer's rope + KV append on the owned rows: every rank's pages of this layer to every rank
static int cp_gather_kv(aha_model* m, int li, const CpPlan& p) {
  const KvLayer kv = model_kv_layer(m, li);
  const int slice = m->desc.num_key_value_heads * 2 * KV_BLOCK_BYTES;
  const size_t per_rank = (size_t)p.pmax * slice;
  if ((size_t)p.world * per_rank > m->cp_stage_bytes) {
    set_error("context-parallel staging buffer too small");
    return AHA_ERR_STATE;
  }
  ProfScope ps(m, "cp_kv_gather", (double)p.world * per_rank, 0);
  const dim3 grid((unsigned)(p.world * p.pmax), (unsigned)(slice / 4096));
  hipLaunchKernelGGL(kv_cp_copy_kernel, grid, dim3(256), 0, m->stream, kv.page_ptrs, (uint64_t)kv.layer_off, slice, (char*)m->p_cp_stage, p, m->cp_rank, -1, 0);
  int rc = cp_all_gather(m, m->p_cp_stage, per_rank);
  if (rc) return rc;
  hipLaunchKernelGGL(kv_cp_copy_kernel, grid, dim3(256), 0, m->stream, kv.page_ptrs, (uint64_t)kv.layer_off, slice, (char*)m->p_cp_stage, p, -1, m->cp_rank, 1);
  return AHA_OK;
}

// ---- decode: one token through all layers; every length-dependent value is read from d_state on the device ----
// First kernel of a decode step: the token's embedding row (Qwen3Model::embedding_token_id, qwen3/model.rs:191-193) and
// the step's rope table -- cos/sin of pos[axis]*inv_freq, cast to the model dtype as apply_rotary_pos_emb does
// (rope.rs:96-132, 454-476), computed ONCE per step here instead of once per layer and block in the attention kernel.
__global__ void embed_state_kernel(const bf16_t* __restrict__ table, const StepState* __restrict__ st, bf16_t* __restrict__ out, int H,
                                   const float* __restrict__ inv_freq, const int32_t* __restrict__ axis_map, float* __restrict__ rope) {
  const u32x4_t* src = reinterpret_cast<const u32x4_t*>(table + (size_t)st->token * H);
  u32x4_t* dst = reinterpret_cast<u32x4_t*>(out);
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < H / 8; i += blockDim.x * gridDim.x) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const int i = threadIdx.x;
    const float ang = (float)st->pos[axis_map[i]] * inv_freq[i];
    rope[i] = rbf(cosf(ang));
    rope[64 + i] = rbf(sinf(ang));
  }
}
__global__ void advance_state_kernel(StepState* st, uint32_t* token_log, uint32_t* host_ring, uint32_t* host_done) {
  const uint32_t t = st->next_token;
  const int32_t step = st->step;
  token_log[step & 0xffff] = t;
  st->step = step + 1;
  st->token = t;
  st->pos[0] += 1;
  st->pos[1] += 1;
  st->pos[2] += 1;
  st->kv_start += 1;
  st->kv_len += 1;
  // publish to the host: token first, then the count (release at system scope: the host reads the count, then the slot)
  __hip_atomic_store(host_ring + ((uint32_t)step % RING_CAP), t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(host_done, (uint32_t)step + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// AHA_GEMV_TRACE=1: in-kernel timeline of the decode matvecs (launch-per-op path), dumped by fetch_outputs
static unsigned long long* gemv_trace_slot(aha_model* m, int launch_idx) {
  static const char* e = getenv("AHA_GEMV_TRACE");
  if (!e || !atoi(e)) return nullptr;
  const size_t n = (size_t)(4 * m->desc.num_hidden_layers + 1) * 18;
  if (!m->d_gemv_trace) {
    void* p = nullptr;
    if (dev_alloc(m, n * 8, &p, true) != AHA_OK) return nullptr;
    m->d_gemv_trace = (unsigned long long*)p;
  }
  return m->d_gemv_trace + (size_t)launch_idx * 18;
}

static void gemv_trace_dump(aha_model* m) {
  const int L = m->desc.num_hidden_layers, nl = 4 * L + 1;
  std::vector<unsigned long long> t((size_t)nl * 18);
  if (hipMemcpy(t.data(), m->d_gemv_trace, t.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  static const char* names[4] = {"qkv", "o_proj", "gate_up", "down"};
  for (int k = 0; k < 4; ++k) {
    double d[3][5] = {}, gap = 0, span = 0;
    int n = 0;
    for (int li = 1; li < L; ++li) {
      const unsigned long long* p = t.data() + (size_t)(li * 4 + k) * 18;
      const unsigned long long* prev = p - 18;
      if (p[0] == 0 || prev[4] == 0) continue;
      for (int b = 0; b < 3; ++b)
        for (int j = 0; j < 4; ++j) d[b][j] += (double)(p[b * 6 + j + 1] - p[b * 6 + j]) * 0.01;
      unsigned long long pend = std::max(prev[4], std::max(prev[6 + 4], prev[12 + 4]));
      unsigned long long start = std::min(p[0], std::min(p[6], p[12]));
      unsigned long long end = std::max(p[4], std::max(p[6 + 4], p[12 + 4]));
      gap += (double)(start - pend) * 0.01;   // previous launch's last consume -> this launch's first instruction
      span += (double)(end - start) * 0.01;
      ++n;
    }
    if (!n) continue;
    fprintf(stderr, "[gemv trace] %-8s span %.2f us, gap since previous launch's last consume %.2f us |", names[k], span / n, gap / n);
    for (int b = 0; b < 3; ++b)
      fprintf(stderr, " blk%d: issue %.2f prologue %.2f first-consume %.2f rest %.2f |", b, d[b][0] / n, d[b][1] / n, d[b][2] / n, d[b][3] / n);
    fprintf(stderr, "\n");
  }
}

static void enqueue_decode_step(aha_model* m, size_t kv_len_after) {
  const aha_model_desc& c = m->desc;
  const int H = c.hidden_size, I = c.intermediate_size, d = c.head_dim, nh = c.num_attention_heads, kvh = c.num_key_value_heads;
  const int nq = nh * d, nkv = kvh * d;
  hipStream_t st = m->stream;
  {
    ProfScope ps(m, "elem", H * 4.0, 0);
    hipLaunchKernelGGL(embed_state_kernel, dim3(1), dim3(256), 0, st, (const bf16_t*)m->embed, m->d_state, (bf16_t*)m->d_x, H,
                       m->d_inv_freq, m->d_axis_map, m->d_rope);
  }
  const int npages = (int)((kv_len_after + KV_PAGE_TOKENS - 1) / KV_PAGE_TOKENS);
  // KV splits of the fused decode attention: one block (4 waves = 4 KV units) per `div` pages
  static const char* e_div = getenv("AHA_ATTN_PAGES_PER_BLOCK");
  const int div = e_div ? std::max(1, atoi(e_div)) : 4;
  int nsplit = (npages + div - 1) / div;
  nsplit = std::max(1, std::min(std::min(nsplit, m->max_nsplit), 1024 / (nh / kvh)));  // g * nsplit <= 1024: LDS tables of the split merge
  for (int li = 0; li < c.num_hidden_layers; ++li) {
    const LayerWeights& L = m->layers[li];
    {  // h = RMSNorm(x); qkv = h Wqkv^T                      (qwen3/model.rs:79, modules.rs:538-552)
      GemvArgs g{};
      g.W = L.wqkv; g.x = m->d_x; g.norm_w = L.in_norm; g.eps = c.rms_norm_eps; g.y = m->d_qkv; g.N = nq + 2 * nkv; g.K = H;
      g.trace = gemv_trace_slot(m, li * 4 + 0);
      ProfScope ps(m, "gemv", (double)g.N * g.K * 2 + g.K * 4.0 + g.N * 2.0, 2.0 * g.N * g.K);
      launch_gemv(g, GEMV_STORE, st);
    }
    if (m->decode_fused) {
      // q/k norm + rope + KV append + attention over the paged cache (modules.rs:544-574, 757-813), then
      // x = x + attn Wo^T with the KV-split partials merged in the matvec prologue (modules.rs:577, qwen3/model.rs:81)
      AttnDecodeFusedArgs a{};
      a.qkv = m->d_qkv; a.q_norm_w = L.q_norm; a.k_norm_w = L.k_norm; a.rope = m->d_rope;
      a.kv = model_kv_layer(m, li); a.kv_start_v = (int)kv_len_after - 1; a.kv_len_v = (int)kv_len_after;
      a.part_o = m->d_part_o; a.part_ml = m->d_part_ml; a.nh = nh; a.kvh = kvh; a.nsplit = nsplit; a.eps = c.rms_norm_eps;
      a.scale = m->attn_scale; a.o = m->d_attn; a.head_ctr = m->d_bar + DECODE_HEAD_CTR_WORD;
      if (nsplit > 1) m->head_ctr_base += (unsigned)nsplit;  // a single split never touches the counter
      a.ctr_target = m->head_ctr_base;
      static const char* e_at = getenv("AHA_ATTN_TRACE");
      if (e_at && atoi(e_at)) {
        if (!m->d_attn_trace) {
          void* tp = nullptr;
          if (dev_alloc(m, (size_t)c.num_hidden_layers * 12 * 8, &tp, true) == AHA_OK) m->d_attn_trace = (unsigned long long*)tp;
        }
        if (m->d_attn_trace) a.trace = m->d_attn_trace + (size_t)li * 12;
      }
      GemvArgs g{};
      g.W = L.wo; g.x = m->d_attn; g.residual = m->d_x; g.y = m->d_x; g.N = H; g.K = nq;
      g.trace = gemv_trace_slot(m, li * 4 + 1);
      const double attn_bytes = (double)kv_len_after * 2 * nkv * 2 + (nq + 2 * nkv) * 2.0 + nsplit * nq * 4.0;
      const double gemv_bytes = (double)g.N * g.K * 2 + g.K * 2.0 + g.N * 4.0;
      {
        ProfScope ps(m, "attn_decode", attn_bytes, 4.0 * kv_len_after * nq);
        launch_attn_decode_fused(a, st);
      }
      ProfScope ps(m, "gemv", gemv_bytes, 2.0 * g.N * g.K);
      gemv_row_parallel(m, g);
    } else {  // three-launch variant (A/B knob AHA_DECODE_FUSED=0)
      {  // q/k norm + rope + append                            (modules.rs:544-566)
        RopeArgs r{};
        r.qkv = m->d_qkv; r.ld = nq + 2 * nkv; r.q_norm_w = L.q_norm; r.k_norm_w = L.k_norm;
        r.pos = m->d_state->pos; r.pos_ld = 1; r.inv_freq = m->d_inv_freq; r.axis_map = m->d_axis_map;
        r.q_out = m->d_q; r.kv = model_kv_layer(m, li); r.kv_start = &m->d_state->kv_start;
        r.S = 1; r.nh = nh; r.kvh = kvh; r.d = d; r.eps = c.rms_norm_eps;
        ProfScope ps(m, "elem", (nq + 2 * nkv) * 4.0, 0);
        launch_qknorm_rope(r, st);
      }
      {  // attention over the paged cache                       (modules.rs:567-574, 757-813)
        AttnDecodeArgs a{};
        a.q = m->d_q; a.kv = model_kv_layer(m, li); a.kv_len = &m->d_state->kv_len; a.part_o = m->d_part_o; a.part_ml = m->d_part_ml;
        a.o = m->d_attn; a.nh = nh; a.kvh = kvh; a.d = d; a.nsplit = nsplit; a.scale = m->attn_scale;
        ProfScope ps(m, "attn_decode", (double)kv_len_after * 2 * nkv * 2 + nq * 4.0, 4.0 * kv_len_after * nq);
        launch_attn_decode(a, st);
      }
      {  // x = x + attn Wo^T                                    (modules.rs:577, qwen3/model.rs:81)
        GemvArgs g{};
        g.W = L.wo; g.x = m->d_attn; g.residual = m->d_x; g.y = m->d_x; g.N = H; g.K = nq;
        ProfScope ps(m, "gemv", (double)g.N * g.K * 2 + g.K * 2.0 + g.N * 4.0, 2.0 * g.N * g.K);
        gemv_row_parallel(m, g);
      }
    }
    {  // act = silu(h Wg^T) * (h Wu^T), h = RMSNorm(x)        (qwen3/model.rs:83, modules.rs:81-84)
      GemvArgs g{};
      g.W = L.wgu; g.W2 = nullptr; g.x = m->d_x; g.norm_w = L.post_norm; g.eps = c.rms_norm_eps; g.y = m->d_act; g.N = I; g.K = H;
      g.trace = gemv_trace_slot(m, li * 4 + 2);
      ProfScope ps(m, "gemv", (double)2 * I * H * 2 + H * 4.0 + I * 2.0, 4.0 * I * H);
      launch_gemv(g, GEMV_SILU_MUL, st);
    }
    {  // x = x + act Wd^T                                     (modules.rs:85, qwen3/model.rs:86)
      GemvArgs g{};
      g.W = L.wdown; g.x = m->d_act; g.residual = m->d_x; g.y = m->d_x; g.N = H; g.K = I;
      g.trace = gemv_trace_slot(m, li * 4 + 3);
      ProfScope ps(m, "gemv", (double)g.N * g.K * 2 + g.K * 2.0 + g.N * 4.0, 2.0 * g.N * g.K);
      gemv_row_parallel(m, g);
    }
  }
  enqueue_lm_head(m, m->d_x);
}

int model_forward_step(aha_model* m, uint32_t token, size_t offset, float* logits_out, uint32_t* argmax_out) {
  const aha_model_desc& c = m->desc;
  if (token >= (uint32_t)c.vocab_size) {
    set_error("token id out of range");
    return AHA_ERR_INVALID;
  }
  AHA_HIP_CHECK(hipSetDevice(m->ctx->device));
  int rc = model_ensure_pages(m, m->cache_len + 1);
  if (rc) return rc;
  // rope position: seqlen_offset (+ rope_delta for Qwen3-VL, qwen3vl/model.rs:1235-1264); cache slot: current length
  int64_t p = (int64_t)offset + m->rope_delta;
  if (c.arch == AHA_ARCH_QWEN3VL && !m->rope_delta_valid) {
    // first forward after clear_cache: get_rope_index on the ids alone -> position 0, delta 0 (model.rs:1229-1236)
    p = 0;
    m->rope_delta = 0;
    m->rope_delta_valid = true;
  }
  const int64_t pos[3] = {p, p, p};
  if ((rc = push_state(m, token, pos, m->cache_len, m->cache_len + 1))) return rc;
  enqueue_decode_step(m, m->cache_len + 1);
  m->cache_len += 1;
  AHA_HIP_CHECK(hipGetLastError());
  if (m->async_rc) { const int e = m->async_rc; m->async_rc = 0; return e; }
  return fetch_outputs(m, logits_out, argmax_out);
}

// Diagnostic (include/aha_hip.h aha_hip_debug_graph_step): what a hipGraph replay of ONE decode step would cost against enqueueing
// its launches, at the current cache length.  Both variants run `replays` times with the SAME kernel arguments (lengths travel as
// kernel arguments here, so a captured step can only be replayed for the length it was captured at): the attention's split-arrival
// target is stale after the first repetition in BOTH, the step's results are not meaningful, and the cache is cleared afterwards --
// the comparison isolates the launch path (stream launches vs one graph launch), nothing else.
int model_debug_graph_step(aha_model* m, int replays, double* us_launches, double* us_graph) {
  if (replays <= 0 || m->cache_len == 0 || m->tp_size > 1 || m->profiling) {
    set_error("debug_graph_step: needs a non-empty cache, tp_size 1, profiling off, replays > 0");
    return AHA_ERR_STATE;
  }
  AHA_HIP_CHECK(hipSetDevice(m->ctx->device));
  int rc = model_ensure_pages(m, m->cache_len + 1);
  if (rc) return rc;
  const int64_t pos[3] = {(int64_t)m->cache_len, (int64_t)m->cache_len, (int64_t)m->cache_len};
  if ((rc = push_state(m, 0, pos, m->cache_len, m->cache_len + 1))) return rc;
  hipEvent_t e0, e1;
  AHA_HIP_CHECK(hipEventCreate(&e0));
  AHA_HIP_CHECK(hipEventCreate(&e1));
  const unsigned base = m->head_ctr_base;
  auto step_with_fixed_args = [&] {
    m->head_ctr_base = base;   // the same arguments every repetition (see above)
    enqueue_decode_step(m, m->cache_len + 1);
  };
  for (int i = 0; i < 3; ++i) step_with_fixed_args();
  AHA_HIP_CHECK(hipEventRecord(e0, m->stream));
  for (int i = 0; i < replays; ++i) step_with_fixed_args();
  AHA_HIP_CHECK(hipEventRecord(e1, m->stream));
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  float ms = 0.f;
  AHA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  if (us_launches) *us_launches = 1e3 * ms / replays;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  AHA_HIP_CHECK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
  step_with_fixed_args();
  AHA_HIP_CHECK(hipStreamEndCapture(m->stream, &graph));
  AHA_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) AHA_HIP_CHECK(hipGraphLaunch(exec, m->stream));
  AHA_HIP_CHECK(hipEventRecord(e0, m->stream));
  for (int i = 0; i < replays; ++i) AHA_HIP_CHECK(hipGraphLaunch(exec, m->stream));
  AHA_HIP_CHECK(hipEventRecord(e1, m->stream));
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  AHA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  if (us_graph) *us_graph = 1e3 * ms / replays;
  hipGraphExecDestroy(exec);
  hipGraphDestroy(graph);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  // leave a consistent model behind: arrival counters back to zero, cache cleared
  AHA_HIP_CHECK(hipMemsetAsync(m->d_bar, 0, DECODE_SYNC_BYTES, m->stream));
  m->head_ctr_base = 0;
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  return model_clear_cache(m);
}

int model_decode_greedy(aha_model* m, uint32_t first_token, size_t offset, size_t max_new, uint32_t* out) {
  const aha_model_desc& c = m->desc;
  if (max_new == 0) return 0;
  if (first_token >= (uint32_t)c.vocab_size) {
    set_error("token id out of range");
    return AHA_ERR_INVALID;
  }
  AHA_HIP_CHECK(hipSetDevice(m->ctx->device));
  int rc = model_ensure_pages(m, m->cache_len + max_new);
  if (rc) return rc;
  const int64_t p = (int64_t)offset + m->rope_delta;
  const int64_t pos[3] = {p, p, p};
  if ((rc = push_state(m, first_token, pos, m->cache_len, m->cache_len + 1))) return rc;
  // Bounded run-ahead: the host keeps at most `ahead` steps queued beyond the last token it has SEEN (the last kernel of a step
  // publishes the token to pinned host memory), so a stop token costs at most `ahead` further steps of GPU time -- and nothing
  // waits on the stream while tokens keep coming: the queue never drains, the host never blocks the device.  (Round 2 enqueued
  // chunks of 32 steps and looked for a stop token after each: up to 31 dead steps, ~100 ms at 8B speed.)
  static const size_t ahead = [] { const char* e = getenv("AHA_DECODE_RUNAHEAD"); return (size_t)std::max(1, std::min(64, e ? atoi(e) : 4)); }();
  m->h_state->step = 0;
  AHA_HIP_CHECK(hipMemcpyAsync(&m->d_state->step, &m->h_state->step, 4, hipMemcpyHostToDevice, m->stream));
  __atomic_store_n(m->h_done, 0u, __ATOMIC_RELEASE);
  size_t produced = 0, enq = 0, seen = 0;
  bool stop = false;
  const size_t base_len = m->cache_len;
  // Tensor parallelism: every step holds collectives, so all ranks must enqueue the SAME number of steps.  The run-ahead rule above
  // depends on when this host thread happens to read h_done (one rank may have seen the stop token and queue nothing while another
  // queues `ahead` more steps and then waits for ever in their all-reduces).  Under TP the schedule is therefore a function of the
  // token sequence alone: whole groups of `ahead` steps, the next group only after every token of the last one has been read.
  const bool lockstep = m->tp_size > 1;
  while (seen < max_new && !stop) {
    size_t want = 0;
    if (!lockstep) want = enq < max_new && enq - seen < ahead ? std::min(max_new - enq, ahead - (enq - seen)) : 0;
    else if (seen == enq) want = std::min(max_new - enq, ahead);
    for (size_t i = 0; i < want; ++i) {
      enqueue_decode_step(m, base_len + enq + 1);
      hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, m->stream, m->d_state, m->d_token_log, m->h_ring_dev, m->h_done_dev);
      ++enq;
    }
    AHA_HIP_CHECK(hipGetLastError());
    if (m->async_rc) { const int e = m->async_rc; m->async_rc = 0; hipStreamSynchronize(m->stream); return e; }
    size_t done = __atomic_load_n(m->h_done, __ATOMIC_ACQUIRE);
    for (unsigned spins = 0; done == seen; ++spins) {   // the next token: ~one step of GPU time away at most
      if ((spins & 1023u) == 1023u && hipStreamQuery(m->stream) == hipSuccess && __atomic_load_n(m->h_done, __ATOMIC_ACQUIRE) == seen) {
        set_error("decode loop: the stream drained without publishing a token");
        return AHA_ERR_STATE;
      }
      done = __atomic_load_n(m->h_done, __ATOMIC_ACQUIRE);
    }
    for (; seen < done && !stop; ++seen) {
      const uint32_t t = __atomic_load_n(m->h_ring + (seen % RING_CAP), __ATOMIC_RELAXED);
      out[produced++] = t;
      for (int e = 0; e < c.n_stop_tokens; ++e)
        if (t == c.stop_tokens[e]) stop = true;
    }
  }
  // steps queued past a stop token still run (at most `ahead` - 1 of them): their KV slots lie beyond the cache length kept below
  // and are overwritten by the next append; wait for them so nothing of this call is in flight when it returns
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  m->steps_executed = (int64_t)enq;
  m->cache_len = base_len + produced;  // inputs consumed: first_token and the first produced-1 generated tokens
  m->have_logits = produced > 0 || m->have_logits;
  if (produced > 0) m->logits_assembled = false;
  return (int)produced;
}

// ---- prefill ---------------------------------------------------------------------------------------------------
static int forward_initial_impl(aha_model* m, const uint32_t* ids, size_t n, size_t offset, const aha_mm_input* mm,
                                float* logits_out, uint32_t* argmax_out, bool hidden_only);

int model_forward_initial(aha_model* m, const uint32_t* ids, size_t n, size_t offset, const aha_mm_input* mm,
                          float* logits_out, uint32_t* argmax_out) {
  return forward_initial_impl(m, ids, n, offset, mm, logits_out, argmax_out, false);
}

// Qwen3Embedding::embed_one (qwen3_embedding/mod.rs:50-64): forward_hidden(ids, offset 0) -> last position after the final
// RMSNorm, bf16 -> f32, l2_normalize over the last dim (modules.rs:1287-1294: x / sqrt(sum(x^2) + 1e-6)), cache cleared.
// The stack runs on the GPU without the lm_head; the H-element normalisation is f32 host arithmetic like the reference's.
int model_embed(aha_model* m, const uint32_t* ids, size_t n, float* out) {
  if (!out) {
    set_error("embed: out is null");
    return AHA_ERR_INVALID;
  }
  if (m->desc.arch != AHA_ARCH_QWEN3) {
    set_error("embed: only the Qwen3 text stack has an embedding head in the reference (qwen3_embedding/mod.rs)");
    return AHA_ERR_UNSUPPORTED;
  }
  int rc = model_clear_cache(m);
  if (rc) return rc;
  if ((rc = forward_initial_impl(m, ids, n, 0, nullptr, nullptr, nullptr, true))) return rc;
  const int H = m->desc.hidden_size;
  std::vector<uint16_t> h((size_t)H);
  AHA_HIP_CHECK(hipMemcpyAsync(h.data(), m->d_hlast, (size_t)H * 2, hipMemcpyDeviceToHost, m->stream));
  AHA_HIP_CHECK(hipStreamSynchronize(m->stream));
  float ss = 0.f;
  for (int i = 0; i < H; ++i) {
    uint32_t u = (uint32_t)h[i] << 16;
    memcpy(&out[i], &u, 4);
    ss += out[i] * out[i];
  }
  const float nrm = sqrtf(ss + 1e-6f);
  for (int i = 0; i < H; ++i) out[i] /= nrm;
  return model_clear_cache(m);
}

static int forward_initial_impl(aha_model* m, const uint32_t* ids, size_t n, size_t offset, const aha_mm_input* mm,
                                float* logits_out, uint32_t* argmax_out, bool hidden_only) {
  const aha_model_desc& c = m->desc;
  if (!ids || n == 0) {
    set_error("forward_initial: empty input_ids");
    return AHA_ERR_INVALID;
  }
  for (size_t i = 0; i < n; ++i)
    if (ids[i] >= (uint32_t)c.vocab_size) {
      set_error("token id out of range at position " + std::to_string(i));
      return AHA_ERR_INVALID;
    }
  const bool has_audio = mm && ((mm->audio_features && mm->n_frames > 0) || (mm->audio_samples && mm->n_samples > 0));
  if (has_audio && (c.arch != AHA_ARCH_QWEN3ASR || !m->audio)) {
    set_error("audio input given but this model has no audio tower");
    return AHA_ERR_UNSUPPORTED;
  }
  if (n == 1 && !mm) return model_forward_step(m, ids[0], offset, logits_out, argmax_out);
  AHA_HIP_CHECK(hipSetDevice(m->ctx->device));
  // AHA_PREFILL_TRACE=1: host wall clock of this call's stages on stderr (where the prefill's host side spends its time: the launches
  // themselves are asynchronous)
  static const bool ptrace = [] { const char* e = getenv("AHA_PREFILL_TRACE"); return e && atoi(e) != 0; }();
  const auto pt0 = std::chrono::steady_clock::now();
  auto pstamp = [&](const char* what) {
    if (ptrace) fprintf(stderr, "[prefill trace] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - pt0).count());
  };
  const int S = (int)n;
  const int H = c.hidden_size, I = c.intermediate_size, d = c.head_dim, nh = c.num_attention_heads, kvh = c.num_key_value_heads;
  const int nq = nh * d, nkv = kvh * d;
  hipStream_t st = m->stream;
  int rc;
  if ((rc = ensure_prefill_scratch(m, n))) return rc;
  GemmWorkspaceScope ws_scope(m->p_gemm_ws, m->gemm_ws_bytes, m->d_sk_ctrs);  // split-K slabs / persistent-kernel chunks of this thread's GEMM launches
  if ((rc = model_ensure_pages(m, m->cache_len + n))) return rc;

  // Context-parallel prefill (cp_make_plan above): this rank owns two row chunks of the prompt.  Its activation buffers are COMPACT --
  // buffer row l0 + i holds prompt row r0 + i of a segment -- so every norm and GEMM below runs ONCE over all of the rank's rows; only
  // rope / KV append and attention, which need the rows' cache positions, run per segment.  A fresh cache only (offset 0), prompts of at
  // least AHA_CP_MIN_ROWS tokens, no audio rows; otherwise every rank simply computes the whole prompt (same results).
  const int kv_off = (int)m->cache_len;
  std::vector<RowSeg> segs;
  CpPlan cpp{};
  const char* e_cp = getenv("AHA_CP_MIN_ROWS");   // (read per call, like the AHA_TP_* thresholds: every rank's host sets it before the call)
  const int cp_min_rows = e_cp ? atoi(e_cp) : 2048;
  const bool cp = m->cp_size > 1 && m->tp_size <= 1 && kv_off == 0 && d == 128 && S >= cp_min_rows && !has_audio &&
                  (m->rccl_comm || m->cp_all_gather_cb) && cp_make_plan(S, m->cp_size, m->cp_rank, &cpp, &segs);
  if (!cp) segs.assign(1, RowSeg{0, S, 0});
  int Mloc = 0;
  for (RowSeg& sg : segs) { sg.l0 = Mloc; Mloc += sg.len; }
  struct MapGuard {   // the vision tower's scatter / DeepStack rows go through m->cp_row_map while this prefill runs
    aha_model* m;
    ~MapGuard() { m->cp_row_map.clear(); }
  } map_guard{m};
  if (cp) {
    m->cp_row_map.assign((size_t)S, Mloc);   // rows of other ranks -> the scratch row behind this rank's rows
    for (const RowSeg& sg : segs)
      for (int i = 0; i < sg.len; ++i) m->cp_row_map[(size_t)sg.r0 + i] = sg.l0 + i;
  }

  // positions: 1-D arange(offset, offset+S) on all three rows (rope.rs:599-604), or get_rope_index for Qwen3-VL
  std::vector<int32_t> pos(3 * (size_t)S);
  const bool has_image = mm && (mm->n_images > 0 || mm->n_videos > 0 || mm->image_embeds);   // images and / or videos
  if (has_image && (c.arch != AHA_ARCH_QWEN3VL || !m->vision)) {
    set_error("image input given but this model has no vision tower (arch / model.visual.* weights)");
    return AHA_ERR_UNSUPPORTED;
  }
  if (c.arch == AHA_ARCH_QWEN3VL) {
    if ((rc = vl_rope_index(m, ids, n, offset, has_image ? mm : nullptr, pos.data()))) return rc;
  } else {
    for (int a = 0; a < 3; ++a)
      for (int i = 0; i < S; ++i) pos[(size_t)a * S + i] = (int32_t)(offset + i);
  }
  const int64_t p0[3] = {pos[0], pos[S], pos[2 * (size_t)S]};
  std::vector<uint32_t> ids_loc;
  std::vector<int32_t> pos_loc;
  const uint32_t* ids_up = ids;
  const int32_t* pos_up = pos.data();
  if (cp) {   // this rank's rows of the ids and of the three position rows
    ids_loc.resize((size_t)Mloc);
    pos_loc.resize(3 * (size_t)Mloc);
    for (const RowSeg& sg : segs) {
      std::copy(ids + sg.r0, ids + sg.r0 + sg.len, ids_loc.begin() + sg.l0);
      for (int a = 0; a < 3; ++a)
        std::copy(pos.begin() + (size_t)a * S + sg.r0, pos.begin() + (size_t)a * S + sg.r0 + sg.len, pos_loc.begin() + (size_t)a * Mloc + sg.l0);
    }
    ids_up = ids_loc.data();
    pos_up = pos_loc.data();
  }
  AHA_HIP_CHECK(hipMemcpyAsync(m->p_ids, ids_up, (size_t)Mloc * 4, hipMemcpyHostToDevice, st));
  AHA_HIP_CHECK(hipMemcpyAsync(m->p_pos, pos_up, 3 * (size_t)Mloc * 4, hipMemcpyHostToDevice, st));
  if ((rc = push_state(m, ids[n - 1], p0, m->cache_len, m->cache_len + n))) return rc;
  pstamp("ids / positions enqueued");
  AHA_HIP_CHECK(hipStreamSynchronize(st));  // pos / ids are pageable host memory
  pstamp("... and landed");

  {
    ProfScope ps(m, "elem", (double)Mloc * H * 4, 0);
    launch_embed_gather(m->embed, m->p_ids, m->p_x, Mloc, H, st);
  }
  if (has_image) {
    // ViT -> masked_scatter of image embeds into the <|image_pad|> rows (qwen3vl/model.rs:1166-1190)
    if ((rc = vision_forward_and_scatter(m, ids, n, mm, m->p_x))) return rc;
    pstamp("vision tower enqueued");
  }
  if (has_audio) {
    // audio tower -> masked_scatter into the <|audio_pad|> rows (qwen3_asr/model.rs:343-358)
    if ((rc = audio_forward_and_scatter(m, ids, n, mm, m->p_x))) return rc;
  }
  if (d == 128) launch_rope_table(m->p_pos, Mloc, m->d_inv_freq, m->d_axis_map, Mloc, m->p_rope, st);   // cos / sin once for all layers
  // sequence-parallel tensor parallelism: rank r owns rows [r * spr, (r+1) * spr) of the residual stream between the GEMMs
  const int spr = seq_parallel_on(m) ? (S + m->tp_size - 1) / m->tp_size : 0;
  auto rows = [](const void* base, int64_t r0, int64_t row_elems) { return (void*)((char*)base + r0 * row_elems * 2); };   // bf16 rows
  // Single GPU: the RMSNorm that follows o_proj / down_proj + residual rides on the GEMM call (GemmArgs::norm_w: inside the
  // split-K reduce pass where the plan has one, a separate launch otherwise -- the same values).  Not across a DeepStack add,
  // which changes the rows between down_proj and the next layer's norm.
  const bool norm_in_gemm = m->tp_size <= 1;
  bool in_norm_done = false;
  for (int li = 0; li < c.num_hidden_layers; ++li) {
    const LayerWeights& L = m->layers[li];
    {
      GemmArgs g{};
      g.A = m->p_h; g.W = L.wqkv; g.C = m->p_qkv; g.M = Mloc; g.N = nq + 2 * nkv; g.K = H; g.lda = H; g.ldw = H; g.ldc = g.N; g.act = ACT_NONE;
      if (spr > 0) {   // sequence-parallel: norm of this rank's rows, all-gather in chunks, GEMM per chunk (norm_gather_gemm)
        if ((rc = norm_gather_gemm(m, L.in_norm, g, S, spr))) return rc;
      } else {
        if (!in_norm_done && (rc = prefill_norm(m, L.in_norm, Mloc, 0))) return rc;
        ProfScope ps(m, "gemm", ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N) * 2, 2.0 * g.M * g.N * g.K);
        launch_gemm(g, st);
      }
    }
    in_norm_done = false;
    // q-norm + RoPE of the q heads inside the attention kernel's Q load where that kernel takes it (round 6: head_dim 128 on the 16-row
    // form, i.e. every prompt below ~3 k tokens): the rope kernel then handles K and V only and the attention reads the raw q heads of p_qkv
    bool q_fused = d == 128 && m->p_rope != nullptr;
    {
      const bool one = segs.size() == 2 && segs[1].l0 == segs[0].l0 + segs[0].len && segs[1].r0 > segs[0].r0;
      for (size_t si = 0; si < segs.size() && q_fused; ++si) {
        AttnPrefillArgs pa{};
        pa.S = segs[si].len; pa.nh = nh; pa.kvh = kvh; pa.d = d; pa.causal = 1; pa.scale = m->attn_scale; pa.rows_hint = (int)S;
        if (one) pa.S2 = segs[1].len;
        q_fused = attn_prefill_takes_qfuse(pa);
        if (one) break;
      }
    }
    for (const RowSeg& sg : segs) {
      RopeArgs r{};
      r.qkv = rows(m->p_qkv, sg.l0, nq + 2 * nkv); r.ld = nq + 2 * nkv; r.q_norm_w = L.q_norm; r.k_norm_w = L.k_norm;
      r.pos = m->p_pos + sg.l0; r.pos_ld = Mloc; r.inv_freq = m->d_inv_freq; r.axis_map = m->d_axis_map;
      r.q_out = rows(m->p_q, sg.l0, nq); r.kv = model_kv_layer(m, li); r.kv_start = &m->d_state->kv_start;
      r.S = sg.len; r.nh = nh; r.kvh = kvh; r.d = d; r.eps = c.rms_norm_eps;
      r.kv_start_host = kv_off + sg.r0;   // == d_state->kv_start (push_state above) for the whole prompt
      r.rope_tab = rows(m->p_rope, sg.l0, 128);
      r.skip_q = q_fused ? 1 : 0;
      ProfScope ps(m, "elem", (double)sg.len * ((q_fused ? 0 : nq) + 2 * nkv) * 4, 0);
      launch_qknorm_rope(r, st);
    }
    if (cp && (rc = cp_gather_kv(m, li, cpp))) return rc;
    // A context-parallel rank's two chunks (early + late, adjacent local rows) go out as ONE launch: each alone is a few hundred
    // blocks -- block rounds and ramp, not tile work, set its time (profiles/r05_shard_rank_time.txt) -- together the early chunk's
    // short blocks fill the late chunk's last round (kernels_attn.hip, AttnPrefillArgs::S2).
    const bool one_launch = segs.size() == 2 && segs[1].l0 == segs[0].l0 + segs[0].len && segs[1].r0 > segs[0].r0;
    for (size_t si = 0; si < segs.size(); ++si) {
      const RowSeg& sg = segs[si];
      AttnPrefillArgs a{};
      a.q = rows(m->p_q, sg.l0, nq); a.kv = model_kv_layer(m, li); a.o = rows(m->p_attn, sg.l0, nq); a.S = sg.len; a.nh = nh; a.kvh = kvh; a.d = d;
      a.kv_offset = kv_off + sg.r0; a.kv_total = kv_off + sg.r0 + sg.len; a.causal = 1; a.scale = m->attn_scale;
      a.rows_hint = (int)S;   // the kernel form by the whole prompt: a context-parallel rank's rows stay bit-identical to the un-sharded prefill
      if (q_fused) {
        a.q = rows(m->p_qkv, sg.l0, nq + 2 * nkv); a.q_ld = nq + 2 * nkv;
        a.q_norm_w = L.q_norm; a.q_rope_tab = rows(m->p_rope, sg.l0, 128); a.q_eps = c.rms_norm_eps;
      }
      double Lk = a.kv_total, flops = 4.0 * sg.len * (a.kv_offset + 0.5 * sg.len) * nq, rows_io = sg.len;
      if (one_launch) {
        const RowSeg& s2 = segs[1];
        a.S2 = s2.len; a.kv_offset2 = kv_off + s2.r0; a.kv_total2 = kv_off + s2.r0 + s2.len;
        Lk = a.kv_total2; flops += 4.0 * s2.len * (a.kv_offset2 + 0.5 * s2.len) * nq; rows_io += s2.len;
      }
      ProfScope ps(m, "attn_prefill", rows_io * nq * 4 + Lk * nkv * 4, flops);
      launch_attn_prefill(a, st);
      if (one_launch) break;
    }
    {
      GemmArgs g{};
      g.A = m->p_attn; g.W = L.wo; g.C = m->p_x; g.residual = m->p_x; g.M = Mloc; g.N = H; g.K = nq; g.lda = nq; g.ldw = nq; g.ldc = H; g.act = ACT_NONE;
      if (norm_in_gemm) { g.norm_w = L.post_norm; g.norm_out = m->p_h; g.norm_eps = c.rms_norm_eps; }
      ProfScope ps(m, "gemm", ((double)g.M * g.K + (double)g.N * g.K + 2.0 * g.M * g.N) * 2, 2.0 * g.M * g.N * g.K);
      if ((rc = gemm_row_parallel(m, g, spr))) return rc;
    }
    {
      GemmArgs g{};
      g.A = m->p_h; g.W = L.wgu; g.C = m->p_act; g.M = Mloc; g.N = 2 * I; g.K = H; g.lda = H; g.ldw = H; g.ldc = I; g.act = ACT_SILU_MUL_PAIRS;
      if (spr > 0) {
        if ((rc = norm_gather_gemm(m, L.post_norm, g, S, spr))) return rc;
      } else {
        if (!norm_in_gemm && (rc = prefill_norm(m, L.post_norm, Mloc, 0))) return rc;
        ProfScope ps(m, "gemm", ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * I) * 2, 2.0 * g.M * g.N * g.K);
        launch_gemm(g, st);
      }
    }
    {
      GemmArgs g{};
      g.A = m->p_act; g.W = L.wdown; g.C = m->p_x; g.residual = m->p_x; g.M = Mloc; g.N = H; g.K = I; g.lda = I; g.ldw = I; g.ldc = H; g.act = ACT_NONE;
      if (norm_in_gemm && li + 1 < c.num_hidden_layers && !(has_image && vision_has_deepstack(m, li))) {
        g.norm_w = m->layers[li + 1].in_norm; g.norm_out = m->p_h; g.norm_eps = c.rms_norm_eps;
        in_norm_done = true;
      }
      ProfScope ps(m, "gemm", ((double)g.M * g.K + (double)g.N * g.K + 2.0 * g.M * g.N) * 2, 2.0 * g.M * g.N * g.K);
      if ((rc = gemm_row_parallel(m, g, spr))) return rc;
    }
    if (has_image) {
      // DeepStack: add visual feature k to the visual rows after decoder layer k (qwen3vl/model.rs:806-822)
      // (context-parallel: the rows of other ranks all land on the scratch row behind this rank's rows)
      if ((rc = vision_deepstack_add(m, li, m->p_x))) return rc;
    }
  }
  const void* x_last = (const char*)m->p_x + (size_t)(Mloc - 1) * H * 2;   // the prompt's last row (context-parallel: on rank 0, which owns the last chunk)
  if (cp) {
    // rank 0 owns the last chunk, hence the last row: every rank contributes its (stale, except rank 0's) copy of that row to one small
    // all-gather and takes slot 0 -- a broadcast; the final norm + lm_head then run replicated (full weights everywhere)
    AHA_HIP_CHECK(hipMemcpyAsync((char*)m->p_cp_stage + (size_t)m->cp_rank * H * 2, x_last, (size_t)H * 2, hipMemcpyDeviceToDevice, st));
    if ((rc = cp_all_gather(m, m->p_cp_stage, (size_t)H * 2))) return rc;
    AHA_HIP_CHECK(hipMemcpyAsync(m->d_x, m->p_cp_stage, (size_t)H * 2, hipMemcpyDeviceToDevice, st));
    x_last = m->d_x;
  }
  if (spr > 0) {
    // the last position lives on the rank that owns row S-1: every rank contributes that row as f32 (zeros elsewhere) to one
    // H-float all-reduce -- a broadcast, exact (bf16 -> f32 -> + 0 -> bf16)
    const bool mine = (S - 1) / spr == m->tp_rank;
    launch_row_to_f32(mine ? x_last : nullptr, m->d_partial, H, st);
    if ((rc = model_allreduce(m, m->d_partial, (size_t)H))) return rc;
    launch_f32_to_row(m->d_partial, m->d_x, H, st);
    x_last = m->d_x;
  }
  if (hidden_only) {  // forward_hidden: final norm of the last position only (qwen3/model.rs:186-188)
    launch_rmsnorm_rows(x_last, m->final_norm, m->d_hlast, 1, H, H, H, c.rms_norm_eps, st);
    m->cache_len += n;
    AHA_HIP_CHECK(hipGetLastError());
    return AHA_OK;
  }
  enqueue_lm_head(m, x_last);
  m->cache_len += n;
  AHA_HIP_CHECK(hipGetLastError());
  if (m->async_rc) { const int e = m->async_rc; m->async_rc = 0; return e; }   // e.g. a failed vocab-parallel all-reduce
  pstamp("decoder stack enqueued");
  rc = fetch_outputs(m, logits_out, argmax_out);
  pstamp("outputs fetched (GPU done)");
  return rc;
}

}  // namespace aha
