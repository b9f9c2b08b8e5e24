// Checkpoint directory -> aha_model_desc + tensor views.  Host code only.
//   config.json / generation_config.json : the fields serde deserialises into Qwen3Config, Qwen3VLConfig, Qwen3ASRConfig
//     (/root/reference/src/models/qwen3/config.rs:4-27, qwen3vl/config.rs:51-133, qwen3_asr/config.rs:6-22)
//   *.safetensors : every file of the directory (find_type_files, /root/reference/src/utils/mod.rs:121-137), mmapped like
//     VarBuilder::from_mmaped_safetensors (qwen3/generate.rs:30-31).  Format: u64 LE header length, JSON header
//     {name: {dtype, shape, data_offsets:[begin,end]}, "__metadata__": {...}}, then the byte buffer.
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>

#include "json.h"
#include "model.h"

aha_weights::~aha_weights() {
  for (auto& m : maps) munmap(m.base, m.len);
}

namespace aha {

namespace {

bool read_file(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out->append(buf, n);
  fclose(f);
  return true;
}

int parse_json_file(const std::string& path, JsonValue* v, bool required) {
  std::string text;
  if (!read_file(path, &text)) {
    if (!required) return 1;
    set_error("cannot read " + path);
    return AHA_ERR_INVALID;
  }
  std::string err;
  if (!JsonParser(text.data(), text.size()).parse(v, &err)) {
    set_error(path + ": " + err);
    return AHA_ERR_INVALID;
  }
  return AHA_OK;
}

// required integer / float fields: a missing key is an error, as it is for serde (no #[serde(default)] on these)
int need_i32(const JsonValue& o, const char* key, const std::string& where, int32_t* out) {
  const JsonValue* v = o.get(key);
  if (!v || !v->is_num()) {
    set_error(where + ": missing numeric field \"" + key + "\"");
    return AHA_ERR_INVALID;
  }
  // serde deserialises these into usize / u32: a fraction, an exponent form that is not integral, a negative value or one
  // beyond i32 is a parse error there, not a silent truncation (strtoll of "1e3" is 1)
  // (serde_json classifies any literal with a fraction or an exponent as a float and refuses to put it into an integer)
  const double x = v->num;
  if (v->str.find_first_of(".eE") != std::string::npos || !(x >= 0.0) || x > 2147483647.0 || x != (double)(int64_t)x) {
    set_error(where + ": field \"" + key + "\" is not a non-negative 32-bit integer");
    return AHA_ERR_INVALID;
  }
  *out = (int32_t)(int64_t)x;
  return AHA_OK;
}
int need_f32(const JsonValue& o, const char* key, const std::string& where, float* out) {
  const JsonValue* v = o.get(key);
  if (!v || !v->is_num()) {
    set_error(where + ": missing numeric field \"" + key + "\"");
    return AHA_ERR_INVALID;
  }
  *out = (float)v->num;
  return AHA_OK;
}
const JsonValue* need_obj(const JsonValue& o, const char* key, const std::string& where) {
  const JsonValue* v = o.get(key);
  if (!v || v->kind != JsonValue::OBJ) {
    set_error(where + ": missing object \"" + key + "\"");
    return nullptr;
  }
  return v;
}
bool get_bool(const JsonValue& o, const char* key, bool dflt) {
  const JsonValue* v = o.get(key);
  return (v && v->kind == JsonValue::BOOL) ? v->b : dflt;
}

#define NEED(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

int parse_text_tower(const JsonValue& t, const std::string& where, aha_model_desc* d) {
  NEED(need_i32(t, "hidden_size", where, &d->hidden_size));
  NEED(need_i32(t, "intermediate_size", where, &d->intermediate_size));
  NEED(need_i32(t, "num_hidden_layers", where, &d->num_hidden_layers));
  NEED(need_i32(t, "num_attention_heads", where, &d->num_attention_heads));
  NEED(need_i32(t, "num_key_value_heads", where, &d->num_key_value_heads));
  NEED(need_i32(t, "head_dim", where, &d->head_dim));
  NEED(need_i32(t, "vocab_size", where, &d->vocab_size));
  NEED(need_f32(t, "rms_norm_eps", where, &d->rms_norm_eps));
  NEED(need_f32(t, "rope_theta", where, &d->rope_theta));
  return AHA_OK;
}

}  // namespace

// The string get_dtype's cfg_dtype argument receives in the reference's XxxGenerateModel::init: Qwen3 `cfg.torch_dtype`
// (qwen3/config.rs:23, generate.rs:28), Qwen3-VL `text_config.dtype` (qwen3vl/config.rs:100), Qwen3-ASR the constant "bfloat16"
// (qwen3_asr/config.rs:186).  A missing field is an error (serde: no default on it).
int config_torch_dtype(const char* dir, std::string* out) {
  const std::string where = std::string(dir) + "/config.json";
  JsonValue cfg;
  NEED(parse_json_file(where, &cfg, true));
  if (cfg.kind != JsonValue::OBJ) {
    set_error(where + ": top level is not an object");
    return AHA_ERR_INVALID;
  }
  if (cfg.get("thinker_config")) {
    *out = "bfloat16";
    return AHA_OK;
  }
  const JsonValue* v = nullptr;
  std::string field = "torch_dtype";
  if (cfg.get("vision_config")) {
    const JsonValue* t = cfg.get("text_config");
    v = t ? t->get("dtype") : nullptr;
    field = "text_config.dtype";
  } else {
    v = cfg.get("torch_dtype");
  }
  if (!v || v->kind != JsonValue::STR) {
    set_error(where + ": missing string field \"" + field + "\"");
    return AHA_ERR_INVALID;
  }
  *out = v->str;
  return AHA_OK;
}

int config_parse(const char* dir, aha_model_desc* d) {
  memset(d, 0, sizeof(*d));
  const std::string base = std::string(dir) + "/";
  JsonValue cfg;
  NEED(parse_json_file(base + "config.json", &cfg, true));
  if (cfg.kind != JsonValue::OBJ) {
    set_error(base + "config.json: top level is not an object");
    return AHA_ERR_INVALID;
  }
  const std::string where = base + "config.json";
  if (cfg.get("thinker_config")) {  // Qwen3ASRConfig { thinker_config: { audio_config, text_config, audio_token_id, .. } }
    d->arch = AHA_ARCH_QWEN3ASR;
    const JsonValue* th = need_obj(cfg, "thinker_config", where);
    if (!th) return AHA_ERR_INVALID;
    const JsonValue* a = need_obj(*th, "audio_config", where + ".thinker_config");
    const JsonValue* t = need_obj(*th, "text_config", where + ".thinker_config");
    if (!a || !t) return AHA_ERR_INVALID;
    NEED(parse_text_tower(*t, where + ".thinker_config.text_config", d));
    d->tie_word_embeddings = get_bool(*t, "tie_word_embeddings", false);  // qwen3_asr/model.rs:319
    const std::string aw = where + ".thinker_config.audio_config";
    NEED(need_i32(*a, "d_model", aw, &d->aud_d_model));
    NEED(need_i32(*a, "encoder_layers", aw, &d->aud_encoder_layers));
    NEED(need_i32(*a, "encoder_attention_heads", aw, &d->aud_attention_heads));
    NEED(need_i32(*a, "encoder_ffn_dim", aw, &d->aud_ffn_dim));
    NEED(need_i32(*a, "num_mel_bins", aw, &d->aud_num_mel_bins));
    NEED(need_i32(*a, "downsample_hidden_size", aw, &d->aud_downsample_hidden_size));
    NEED(need_i32(*a, "output_dim", aw, &d->aud_output_dim));
    NEED(need_i32(*a, "n_window", aw, &d->aud_n_window));
    NEED(need_i32(*th, "audio_token_id", where + ".thinker_config", &d->audio_token_id));
    // text rope: the three M-RoPE position rows are identical for this model (rope.rs:486-498 is moot, SURVEY.md
    // appendix A.6), which equals plain 1-D RoPE: mrope_section stays {0,0,0}
  } else if (cfg.get("vision_config")) {  // Qwen3VLConfig
    d->arch = AHA_ARCH_QWEN3VL;
    const JsonValue* t = need_obj(cfg, "text_config", where);
    const JsonValue* v = need_obj(cfg, "vision_config", where);
    if (!t || !v) return AHA_ERR_INVALID;
    NEED(parse_text_tower(*t, where + ".text_config", d));
    d->tie_word_embeddings = get_bool(cfg, "tie_word_embeddings", false);  // top-level flag decides (qwen3vl/model.rs:853)
    const JsonValue* rs = need_obj(*t, "rope_scaling", where + ".text_config");
    if (!rs) return AHA_ERR_INVALID;
    const JsonValue* ms = rs->get("mrope_section");
    if (!ms || ms->kind != JsonValue::ARR || ms->arr.size() != 3) {
      set_error(where + ".text_config.rope_scaling: mrope_section must be a list of 3 integers");
      return AHA_ERR_INVALID;
    }
    for (int i = 0; i < 3; ++i) d->mrope_section[i] = (int32_t)ms->arr[i].as_i64();
    const std::string vw = where + ".vision_config";
    NEED(need_i32(*v, "depth", vw, &d->vis_depth));
    NEED(need_i32(*v, "hidden_size", vw, &d->vis_hidden_size));
    NEED(need_i32(*v, "num_heads", vw, &d->vis_num_heads));
    NEED(need_i32(*v, "intermediate_size", vw, &d->vis_intermediate_size));
    NEED(need_i32(*v, "in_channels", vw, &d->vis_in_channels));
    NEED(need_i32(*v, "patch_size", vw, &d->vis_patch_size));
    NEED(need_i32(*v, "temporal_patch_size", vw, &d->vis_temporal_patch_size));
    NEED(need_i32(*v, "spatial_merge_size", vw, &d->vis_spatial_merge_size));
    NEED(need_i32(*v, "out_hidden_size", vw, &d->vis_out_hidden_size));
    NEED(need_i32(*v, "num_position_embeddings", vw, &d->vis_num_position_embeddings));
    const JsonValue* ds = v->get("deepstack_visual_indexes");
    if (!ds || ds->kind != JsonValue::ARR || ds->arr.size() > 8) {
      set_error(vw + ": deepstack_visual_indexes must be a list of at most 8 integers");
      return AHA_ERR_INVALID;
    }
    d->vis_num_deepstack = (int32_t)ds->arr.size();
    for (size_t i = 0; i < ds->arr.size(); ++i) d->vis_deepstack_indexes[i] = (int32_t)ds->arr[i].as_i64();
    NEED(need_i32(cfg, "image_token_id", where, &d->image_token_id));
    NEED(need_i32(cfg, "video_token_id", where, &d->video_token_id));
    NEED(need_i32(cfg, "vision_start_token_id", where, &d->vision_start_token_id));
    NEED(need_i32(cfg, "vision_end_token_id", where, &d->vision_end_token_id));
  } else {  // Qwen3Config
    d->arch = AHA_ARCH_QWEN3;
    NEED(parse_text_tower(cfg, where, d));
    const JsonValue* tie = cfg.get("tie_word_embeddings");
    if (!tie || tie->kind != JsonValue::BOOL) {
      set_error(where + ": missing boolean field \"tie_word_embeddings\"");
      return AHA_ERR_INVALID;
    }
    d->tie_word_embeddings = tie->b;
  }
  // generation_config.json: eos_token_id is Vec<u32> in the reference (qwen3/config.rs:34); HF also writes a scalar
  JsonValue gen;
  int rc = parse_json_file(base + "generation_config.json", &gen, true);
  if (rc) return rc;
  const JsonValue* eos = gen.get("eos_token_id");
  if (!eos) {
    set_error(base + "generation_config.json: missing \"eos_token_id\"");
    return AHA_ERR_INVALID;
  }
  if (eos->kind == JsonValue::NUM) {
    d->n_stop_tokens = 1;
    d->stop_tokens[0] = (uint32_t)eos->as_u64();
  } else if (eos->kind == JsonValue::ARR && eos->arr.size() <= 8) {
    d->n_stop_tokens = (int32_t)eos->arr.size();
    for (size_t i = 0; i < eos->arr.size(); ++i) d->stop_tokens[i] = (uint32_t)eos->arr[i].as_u64();
  } else {
    set_error(base + "generation_config.json: eos_token_id must be an integer or a list of at most 8 integers");
    return AHA_ERR_INVALID;
  }
  return AHA_OK;
}

namespace {

int st_dtype(const std::string& s, size_t* elem) {
  if (s == "BF16") { *elem = 2; return AHA_BF16; }
  if (s == "F16") { *elem = 2; return AHA_F16; }
  if (s == "F32") { *elem = 4; return AHA_F32; }
  if (s == "U32") { *elem = 4; return AHA_U32; }
  if (s == "U8") { *elem = 1; return AHA_U8; }
  if (s == "I64" || s == "F64" || s == "U64") { *elem = 8; return -1; }
  if (s == "I32") { *elem = 4; return -1; }
  if (s == "I16" || s == "U16") { *elem = 2; return -1; }
  if (s == "I8" || s == "BOOL" || s == "F8_E4M3" || s == "F8_E5M2") { *elem = 1; return -1; }
  *elem = 0;
  return -1;
}

int open_one(const std::string& path, aha_weights* w) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) {
    set_error("cannot open " + path);
    return AHA_ERR_INVALID;
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 8) {
    close(fd);
    set_error(path + ": not a safetensors file (shorter than its 8-byte header length)");
    return AHA_ERR_INVALID;
  }
  const size_t len = (size_t)st.st_size;
  void* base = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (base == MAP_FAILED) {
    set_error("mmap of " + path + " failed");
    return AHA_ERR_OOM;
  }
  w->maps.push_back({base, len});
  const unsigned char* b = (const unsigned char*)base;
  uint64_t hlen = 0;
  for (int i = 7; i >= 0; --i) hlen = (hlen << 8) | b[i];
  if (hlen > len - 8 || hlen > (1ull << 30)) {
    set_error(path + ": header length " + std::to_string(hlen) + " exceeds the file");
    return AHA_ERR_INVALID;
  }
  JsonValue hdr;
  std::string err;
  if (!JsonParser((const char*)b + 8, (size_t)hlen).parse(&hdr, &err) || hdr.kind != JsonValue::OBJ) {
    set_error(path + ": bad header: " + (err.empty() ? "not an object" : err));
    return AHA_ERR_INVALID;
  }
  const unsigned char* data = b + 8 + hlen;
  const size_t data_len = len - 8 - (size_t)hlen;
  for (const auto& kv : hdr.obj) {
    if (kv.first == "__metadata__") continue;
    const JsonValue& t = kv.second;
    const JsonValue* dt = t.get("dtype");
    const JsonValue* sh = t.get("shape");
    const JsonValue* off = t.get("data_offsets");
    if (!dt || dt->kind != JsonValue::STR || !sh || sh->kind != JsonValue::ARR || !off || off->kind != JsonValue::ARR ||
        off->arr.size() != 2) {
      set_error(path + ": tensor \"" + kv.first + "\": malformed entry");
      return AHA_ERR_INVALID;
    }
    size_t elem = 0;
    const int dtype = st_dtype(dt->str, &elem);
    if (elem == 0) {
      set_error(path + ": tensor \"" + kv.first + "\": unknown dtype " + dt->str);
      return AHA_ERR_UNSUPPORTED;
    }
    if (sh->arr.size() > 5) {
      set_error(path + ": tensor \"" + kv.first + "\": more than 5 dimensions");
      return AHA_ERR_UNSUPPORTED;
    }
    const uint64_t b0 = off->arr[0].as_u64(), b1 = off->arr[1].as_u64();
    uint64_t numel = 1;
    aha_tensor_view v;
    memset(&v, 0, sizeof(v));
    v.ndim = (int32_t)sh->arr.size();
    for (size_t i = 0; i < sh->arr.size(); ++i) {
      v.shape[i] = sh->arr[i].as_i64();
      numel *= (uint64_t)v.shape[i];
    }
    if (b1 < b0 || b1 > data_len || b1 - b0 != numel * elem) {
      set_error(path + ": tensor \"" + kv.first + "\": data_offsets [" + std::to_string(b0) + ", " + std::to_string(b1) +
                ") do not match shape x dtype (" + std::to_string(numel * elem) + " bytes) or exceed the file");
      return AHA_ERR_INVALID;
    }
    for (const auto& n : w->names)
      if (n == kv.first) {
        set_error(path + ": tensor \"" + kv.first + "\" appears in more than one file");
        return AHA_ERR_INVALID;
      }
    v.dtype = dtype;  // -1: a dtype the model never reads (e.g. I64 position ids); listed, rejected only if looked up
    v.data = data + b0;
    v.on_device = 0;
    w->names.push_back(kv.first);
    w->views.push_back(v);
  }
  return AHA_OK;
}

}  // namespace

int weights_open(const char* dir, aha_weights** out) {
  DIR* d = opendir(dir);
  if (!d) {
    set_error(std::string("cannot open directory ") + dir);
    return AHA_ERR_INVALID;
  }
  std::vector<std::string> files;
  while (struct dirent* e = readdir(d)) {
    const std::string n = e->d_name;
    if (n.size() > 12 && n.compare(n.size() - 12, 12, ".safetensors") == 0) files.push_back(std::string(dir) + "/" + n);
  }
  closedir(d);
  if (files.empty()) {
    set_error(std::string("no *.safetensors file in ") + dir);
    return AHA_ERR_INVALID;
  }
  std::sort(files.begin(), files.end());
  std::unique_ptr<aha_weights> w(new aha_weights());
  for (const auto& f : files) {
    int rc = open_one(f, w.get());
    if (rc) return rc;
  }
  for (size_t i = 0; i < w->views.size(); ++i) w->views[i].name = w->names[i].c_str();
  *out = w.release();
  return AHA_OK;
}

int model_load(aha_ctx* ctx, const char* dir, size_t kv_reserve_tokens, aha_model** out) {
  aha_model_desc d;
  int rc = config_parse(dir, &d);
  if (rc) return rc;
  d.kv_reserve_tokens = (int32_t)kv_reserve_tokens;
  aha_weights* w = nullptr;
  if ((rc = weights_open(dir, &w))) return rc;
  std::vector<aha_tensor_view> usable;
  for (const auto& v : w->views)
    if (v.dtype >= 0) usable.push_back(v);
  rc = model_create(ctx, &d, usable.data(), usable.size(), out);
  delete w;
  return rc;
}

}  // namespace aha
